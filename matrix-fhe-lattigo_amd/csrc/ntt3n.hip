// ntt3n.hip -- 3N-cyclotomic transform over Z_q[X]/(X^N - X^(N/2) + 1), N = 2^a 3^b (a,b >= 1).
//
// Replaces NumberTheoreticTransformer3N (ring/ntt_3n.go:21-156): Forward(p)[k] = p(omega^E[k]) with E the ascending
// totatives of 3N (:82-109, :235-243); Backward = the exact inverse (:118-151).  The reference evaluates by Horner
// (O(N^2)) and interpolates by Gaussian elimination (O(N^3)); both maps are unique, so an O(N log N) factorisation
// that lands in [0,q) is bit-identical.  Factorisation (math of references/integer_dft.py:150-183, 266-432):
//     X^N - X^(N/2) + 1 = (X^(N/2) - z)(X^(N/2) - z^5),   z = omega^(N/2)             "split" layer
//     X^(3m) - w^e      = prod_{k<3} (X^m - w^(e/3 + kN))                              b radix-3 layers
//     X^(2m) - w^e      = (X^m - w^(e/2))(X^m - w^(e/2 + 3N/2))                        a-1 radix-2 layers
// After the split and radix-3 layers a limb is nb = 2*3^b independent twisted power-of-two transforms of length
// n2 = 2^(a-1); they run on the SAME radix-2 kernels as the negacyclic NTT (ntt_kernels.hip.hpp) by presenting each
// (limb, block) as a "virtual limb" of a sub-ring with its own twiddle table.  Slot j of block c holds the value at
// omega^e, e = e0_c + 2*3^(b+1)*bitrev(j); its rank among the totatives is nb*bitrev(j) + rank(e0_c), which is the
// final permutation (ntt3n_perm_*).
#include <hip/hip_runtime.h>
#include <vector>
#include <cstring>
#include "engine_internal.hpp"
#include "hostmath.hpp"

struct rh_ring3n_state {
  int a = 0, b = 0, nb = 0, n2 = 0, log_n2 = 0;
  rh_ring* sub = nullptr;             // radix-2 part: L*nb virtual limbs of length n2
  Limb3N* d_l3 = nullptr;             // [L]
  tw2* d_r3_fwd = nullptr;            // radix-3 twiddles: per limb, per layer blocks: (z1, z2) pairs, forward
  tw2* d_r3_inv = nullptr;            //                                             (z1^-1, z2^-1)
  std::vector<int> r3_off;            // offset (in pairs of tw2) of layer l inside a limb's radix-3 table
  int r3_stride = 0;                  // tw2 entries per limb
  int* d_rank = nullptr;              // [nb] rank(e0_c)
  int* d_block_of_rank = nullptr;     // [nb] inverse map
  u64* d_tmp = nullptr; size_t tmp_words = 0;
};

// ---------------------------------------------------------------------------------------------------------------
// kernels.  Rows are (poly, limb) pairs, grid.x = rows, grid.y = chunks over the butterflies of the layer.
// All layer kernels keep values < 4q (inputs < 4q accepted).
// ---------------------------------------------------------------------------------------------------------------
RH_DEV u64 add4(u64 x, u64 y, u64 q4) { return csub(x + y, q4); }          // x,y < 4q -> < 4q
RH_DEV u64 sub4(u64 x, u64 y, u64 q4) { return csub(x + q4 - y, q4); }     // x,y < 4q -> < 4q

__global__ void __launch_bounds__(256)
ntt3n_split_fwd(const u64* in, u64* out, int N, const Limb3N* __restrict__ l3, const LimbConsts* __restrict__ consts, int L) {
  const u32 row = blockIdx.x, limb = row % (u32)L;
  const LimbConsts c = consts[limb]; const Limb3N k = l3[limb];
  const u64 q4 = 4 * c.q;
  const size_t base = (size_t)row * N;
  const int h = N >> 1;
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < h; i += gridDim.y * blockDim.x) {
    u64 b0 = csub(in[base + i], q4), b1 = csub(in[base + i + h], q4);     // tolerate inputs < 8q
    u64 t = shoup_mul(b1, k.zeta.w, k.zeta.wp, c.nq);
    out[base + i] = add4(b0, t, q4);                                      // f mod (X^h - z)
    out[base + i + h] = sub4(add4(b0, b1, q4), t, q4);                    // f mod (X^h - z^5), z^5 = 1 - z
  }
}

// radix-3 layer: `cnt` blocks of 3*step; block blk uses (z1, z2) = tw[blk*2], tw[blk*2+1]
__global__ void __launch_bounds__(256)
ntt3n_radix3_fwd(u64* data, int N, int step, int cnt, const tw2* __restrict__ tw, int tw_stride,
                 const Limb3N* __restrict__ l3, const LimbConsts* __restrict__ consts, int L) {
  const u32 row = blockIdx.x, limb = row % (u32)L;
  const LimbConsts c = consts[limb]; const tw2 w3 = l3[limb].w3;
  const u64 q4 = 4 * c.q;
  const size_t base = (size_t)row * N;
  const tw2* t = tw + (size_t)limb * tw_stride;
  const int nbf = cnt * step;
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < nbf; i += gridDim.y * blockDim.x) {
    const int blk = i / step, e = i - blk * step;
    const size_t j = base + (size_t)blk * 3 * step + e;
    const tw2 z1 = t[2 * blk], z2 = t[2 * blk + 1];
    u64 b0 = data[j], b1 = data[j + step], b2 = data[j + 2 * step];
    u64 t1 = shoup_mul(b1, z1.w, z1.wp, c.nq), t2 = shoup_mul(b2, z2.w, z2.wp, c.nq);
    u64 t3 = shoup_mul(t1 + q4 - t2, w3.w, w3.wp, c.nq);
    data[j] = add4(add4(b0, t1, q4), t2, q4);
    data[j + step] = add4(sub4(b0, t2, q4), t3, q4);
    data[j + 2 * step] = sub4(sub4(b0, t1, q4), t3, q4);
  }
}
// inverse radix-3 (un-normalised, factor 3): s0 = B0+B1+B2, t = w3*(B1-B2), b1 = (B0-B1-t)/z1, b2 = (B0-B2+t)/z2
__global__ void __launch_bounds__(256)
ntt3n_radix3_inv(u64* data, int N, int step, int cnt, const tw2* __restrict__ tw, int tw_stride,
                 const Limb3N* __restrict__ l3, const LimbConsts* __restrict__ consts, int L) {
  const u32 row = blockIdx.x, limb = row % (u32)L;
  const LimbConsts c = consts[limb]; const tw2 w3 = l3[limb].w3;
  const u64 q4 = 4 * c.q;
  const size_t base = (size_t)row * N;
  const tw2* t = tw + (size_t)limb * tw_stride;
  const int nbf = cnt * step;
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < nbf; i += gridDim.y * blockDim.x) {
    const int blk = i / step, e = i - blk * step;
    const size_t j = base + (size_t)blk * 3 * step + e;
    const tw2 z1 = t[2 * blk], z2 = t[2 * blk + 1];
    u64 B0 = data[j], B1 = data[j + step], B2 = data[j + 2 * step];
    u64 tt = shoup_mul(B1 + q4 - B2, w3.w, w3.wp, c.nq);
    u64 s1 = sub4(sub4(B0, B1, q4), tt, q4), s2 = add4(sub4(B0, B2, q4), tt, q4);
    data[j] = add4(add4(B0, B1, q4), B2, q4);
    data[j + step] = shoup_mul(s1, z1.w, z1.wp, c.nq);
    data[j + 2 * step] = shoup_mul(s2, z2.w, z2.wp, c.nq);
  }
}
// inverse split + scaling by (N/2)^-1 + canonical reduction:
//   lo = b0 + z b1, hi = b0 + z^5 b1  =>  b1 = (hi-lo)/(z^5-z), b0 = lo - z b1
__global__ void __launch_bounds__(256)
ntt3n_split_inv(const u64* in, u64* out, int N, const Limb3N* __restrict__ l3, const LimbConsts* __restrict__ consts, int L) {
  const u32 row = blockIdx.x, limb = row % (u32)L;
  const LimbConsts c = consts[limb]; const Limb3N k = l3[limb];
  const u64 q4 = 4 * c.q;
  const size_t base = (size_t)row * N;
  const int h = N >> 1;
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < h; i += gridDim.y * blockDim.x) {
    u64 lo = in[base + i], hi = in[base + i + h];
    u64 d = hi + q4 - lo;
    u64 b1 = shoup_mul(d, k.inv_b1.w, k.inv_b1.wp, c.nq);
    u64 zb1 = shoup_mul(d, k.inv_b0z.w, k.inv_b0z.wp, c.nq);
    u64 los = shoup_mul(lo, k.inv_s.w, k.inv_s.wp, c.nq);
    out[base + i] = canon8(los + q4 - zb1, c.q);
    out[base + i + h] = canon4(b1, c.q);
  }
}

RH_DEV u32 brev(u32 x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }
// forward permutation: out[nb*bitrev(j) + rank[c]] = in[c*n2 + j]; thread per OUTPUT element (contiguous writes)
__global__ void __launch_bounds__(256)
ntt3n_perm_fwd(const u64* in, u64* out, int N, int nb, int log_n2, const int* __restrict__ block_of_rank,
               const LimbConsts* __restrict__ consts, int L, int canon) {
  const size_t base = (size_t)blockIdx.x * N;
  const u64 q = consts[blockIdx.x % (u32)L].q;
  for (int o = blockIdx.y * blockDim.x + threadIdx.x; o < N; o += gridDim.y * blockDim.x) {
    const int jb = o / nb, rk = o - jb * nb;
    const int c = block_of_rank[rk];
    u64 v = in[base + ((size_t)c << log_n2) + brev((u32)jb, log_n2)];
    out[base + o] = canon ? canon4(v, q) : v;     // canon: only when there is no radix-2 part (a == 1), values < 4q
  }
}
// inverse permutation: out[c*n2 + j] = in[nb*bitrev(j) + rank[c]]; thread per INPUT element (contiguous reads)
__global__ void __launch_bounds__(256)
ntt3n_perm_inv(const u64* in, u64* out, int N, int nb, int log_n2, const int* __restrict__ block_of_rank) {
  const size_t base = (size_t)blockIdx.x * N;
  for (int o = blockIdx.y * blockDim.x + threadIdx.x; o < N; o += gridDim.y * blockDim.x) {
    const int jb = o / nb, rk = o - jb * nb;
    const int c = block_of_rank[rk];
    out[base + ((size_t)c << log_n2) + brev((u32)jb, log_n2)] = in[base + o];
  }
}

// the b = 1 butterflies on the six coefficients {i + k*N/6}: split + radix-3 layer (forward) and their inverses
RH_DEV void pre_b1(u64 (&x)[6], const LimbConsts& c, const Limb3N& k, const tw2* __restrict__ t, u64 q4) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const u64 tt = shoup_mul(x[j + 3], k.zeta.w, k.zeta.wp, c.nq);
    const u64 lo = add4(x[j], tt, q4), hi = sub4(add4(x[j], x[j + 3], q4), tt, q4);
    x[j] = lo; x[j + 3] = hi;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const tw2 z1 = t[2 * h], z2 = t[2 * h + 1];
    const u64 b0 = x[3 * h], b1 = x[3 * h + 1], b2 = x[3 * h + 2];
    const u64 t1 = shoup_mul(b1, z1.w, z1.wp, c.nq), t2 = shoup_mul(b2, z2.w, z2.wp, c.nq);
    const u64 t3 = shoup_mul(t1 + q4 - t2, k.w3.w, k.w3.wp, c.nq);
    x[3 * h] = add4(add4(b0, t1, q4), t2, q4);
    x[3 * h + 1] = add4(sub4(b0, t2, q4), t3, q4);
    x[3 * h + 2] = sub4(sub4(b0, t1, q4), t3, q4);
  }
}
RH_DEV void post_b1(u64 (&x)[6], const LimbConsts& c, const Limb3N& k, const tw2* __restrict__ t, u64 q4) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const tw2 z1 = t[2 * h], z2 = t[2 * h + 1];
    const u64 B0 = x[3 * h], B1 = x[3 * h + 1], B2 = x[3 * h + 2];
    const u64 tt = shoup_mul(B1 + q4 - B2, k.w3.w, k.w3.wp, c.nq);
    const u64 s1 = sub4(sub4(B0, B1, q4), tt, q4), s2 = add4(sub4(B0, B2, q4), tt, q4);
    x[3 * h] = add4(add4(B0, B1, q4), B2, q4);
    x[3 * h + 1] = shoup_mul(s1, z1.w, z1.wp, c.nq);
    x[3 * h + 2] = shoup_mul(s2, z2.w, z2.wp, c.nq);
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const u64 lo = x[j], hi = x[j + 3];
    const u64 d = hi + q4 - lo;
    const u64 b1 = shoup_mul(d, k.inv_b1.w, k.inv_b1.wp, c.nq);
    const u64 zb1 = shoup_mul(d, k.inv_b0z.w, k.inv_b0z.wp, c.nq);
    const u64 los = shoup_mul(lo, k.inv_s.w, k.inv_s.wp, c.nq);
    x[j] = canon8(los + q4 - zb1, c.q); x[j + 3] = canon4(b1, c.q);
  }
}

// ---- b = 1 fast path: split + the single radix-3 layer fused, 6 coefficients {i + k*N/6} per thread (one pass) -------
__global__ void __launch_bounds__(256)
ntt3n_pre_b1_fwd(const u64* in, u64* out, int N, const tw2* __restrict__ r3, int r3_stride,
                 const Limb3N* __restrict__ l3, const LimbConsts* __restrict__ consts, int L) {
  const u32 row = blockIdx.x, limb = row % (u32)L;
  const LimbConsts c = consts[limb]; const Limb3N k = l3[limb];
  const tw2* t = r3 + (size_t)limb * r3_stride;           // layer 1: blocks 0,1 -> (z1,z2) pairs
  const u64 q4 = 4 * c.q;
  const size_t base = (size_t)row * N;
  const int s = N / 6;
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < s; i += gridDim.y * blockDim.x) {
    u64 x[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) x[j] = csub(in[base + i + (size_t)j * s], q4);
    pre_b1(x, c, k, t, q4);
#pragma unroll
    for (int j = 0; j < 6; ++j) out[base + i + (size_t)j * s] = x[j];
  }
}
__global__ void __launch_bounds__(256)
ntt3n_post_b1_inv(const u64* in, u64* out, int N, const tw2* __restrict__ r3, int r3_stride,
                  const Limb3N* __restrict__ l3, const LimbConsts* __restrict__ consts, int L) {
  const u32 row = blockIdx.x, limb = row % (u32)L;
  const LimbConsts c = consts[limb]; const Limb3N k = l3[limb];
  const tw2* t = r3 + (size_t)limb * r3_stride;
  const u64 q4 = 4 * c.q;
  const size_t base = (size_t)row * N;
  const int s = N / 6;
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < s; i += gridDim.y * blockDim.x) {
    u64 x[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) x[j] = in[base + i + (size_t)j * s];
    post_b1(x, c, k, t, q4);
#pragma unroll
    for (int j = 0; j < 6; ++j) out[base + i + (size_t)j * s] = x[j];
  }
}

// ---- b = 1, n2 >= 8192: the split + radix-3 layer fused with the COLUMN stages of the radix-2 sub-transforms (one pass
// instead of two).  A thread owns R = 2^S1 local positions p_k = col + 4096 k of all six blocks: 6R coefficients in
// registers.  After the pre-butterflies, block j's R values are exactly what ntt_fwd_cols would load for that block, so
// the first S1 stages run in place on them with block j's twiddles (virtual limb limb*6 + j of the sub-ring) and the
// sub-ring only runs its tile kernel afterwards.  Same arithmetic as ntt3n_pre_b1_fwd followed by fwd_cols_body.
template <int S1>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, S1 == 3 ? 3 : (S1 == 2 ? 5 : 8))))   // 6 * 2^S1 coefficients live
ntt3n_pre_cols_fwd(const u64* in, u64* out, int N, const tw2* __restrict__ r3, int r3_stride, const Limb3N* __restrict__ l3,
                   const LimbConsts* __restrict__ consts, int L, const tw2* __restrict__ sub_tw, int log_n2) {
  constexpr int R = 1 << S1;
  const u32 limb = blockIdx.x % (u32)L, rr = blockIdx.x / (u32)L;
  const u32 col = (rr & 15) * 256 + threadIdx.x;
  const size_t base = ((size_t)(rr >> 4) * L + limb) * N;
  const LimbConsts c = consts[limb]; const Limb3N k3 = l3[limb];
  const tw2* t = r3 + (size_t)limb * r3_stride;
  const u64 q4 = 4 * c.q;
  const size_t s = (size_t)1 << log_n2;                    // = N / 6
  u64 x[6][R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    u64 v[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) v[j] = csub(in[base + col + ((size_t)k << 12) + j * s], q4);
    pre_b1(v, c, k3, t, q4);
#pragma unroll
    for (int j = 0; j < 6; ++j) x[j][k] = v[j];
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const tw2* tw = sub_tw + ((size_t)(limb * 6 + j) << log_n2);
#pragma unroll
    for (int st = 0; st < S1; ++st) {
      const int h = R >> (st + 1);
#pragma unroll
      for (int g = 0; g < (1 << st); ++g) {
        const tw2 w = tw[(1 << st) + g];
#pragma unroll
        for (int e = 0; e < h; ++e) {                      // ShoupPolicy::fwd (ntt_kernels.hip.hpp)
          u64& U = x[j][g * 2 * h + e]; u64& V = x[j][g * 2 * h + e + h];
          const u64 u = csub(U, q4);
          const u64 X = shoup_mul_acc(V, w.w, w.wp, c.nq, u);
          V = ((u << 1) + q4) - X;
          U = X;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) out[base + j * s + col + ((size_t)k << 12)] = x[j][k];
  }
}

// inverse mirror: the sub-transforms' column stages (unscaled, values < 4q) fused with the radix-3 layer + split
template <int S1>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, S1 == 3 ? 4 : (S1 == 2 ? 5 : 8))))   // 6 * 2^S1 coefficients live
ntt3n_cols_post_inv(const u64* in, u64* out, int N, const tw2* __restrict__ r3, int r3_stride, const Limb3N* __restrict__ l3,
                    const LimbConsts* __restrict__ consts, int L, const tw2* __restrict__ sub_tw, int log_n2) {
  constexpr int R = 1 << S1;
  const u32 limb = blockIdx.x % (u32)L, rr = blockIdx.x / (u32)L;
  const u32 col = (rr & 15) * 256 + threadIdx.x;
  const size_t base = ((size_t)(rr >> 4) * L + limb) * N;
  const LimbConsts c = consts[limb]; const Limb3N k3 = l3[limb];
  const tw2* t = r3 + (size_t)limb * r3_stride;
  const u64 q4 = 4 * c.q;
  const size_t s = (size_t)1 << log_n2;
  u64 x[6][R];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
#pragma unroll
    for (int k = 0; k < R; ++k) x[j][k] = in[base + j * s + col + ((size_t)k << 12)];
    const tw2* tw = sub_tw + ((size_t)(limb * 6 + j) << log_n2);
#pragma unroll
    for (int it = 0; it < S1; ++it) {                      // inv_cols_body(scale = 0): stages with 2^st blocks, the last one plain
      const int st = S1 - 1 - it;
      const int h = R >> (st + 1);
#pragma unroll
      for (int g = 0; g < (1 << st); ++g) {
        const tw2 w = tw[(1 << st) + g];
#pragma unroll
        for (int e = 0; e < h; ++e) {                      // ShoupPolicy::inv
          u64& U = x[j][g * 2 * h + e]; u64& V = x[j][g * 2 * h + e + h];
          const u64 d = U + q4 - V;
          U = csub(U + V, q4);
          V = shoup_mul(d, w.w, w.wp, c.nq);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    u64 v[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) v[j] = x[j][k];
    post_b1(v, c, k3, t, q4);
#pragma unroll
    for (int j = 0; j < 6; ++j) out[base + col + ((size_t)k << 12) + j * s] = v[j];
  }
}

// ---- tiled permutation (log_n2 >= 10): a tile is {j_low: 32} x {j_high: 32} x {nb blocks} for one j_mid.
// Reads are 256 B runs (32 consecutive j), writes are nb*256 B runs (32 consecutive bitrev(j) x nb ranks).
#define PT 5
// A = low bits of j (contiguous run on the block-order side), B = high bits (contiguous, bit-reversed, on the rank-order
// side): a tile is {jlow: 2^A} x {jhigh: 2^B} x {nb blocks} for one jmid.  Block-order runs are 2^A words, rank-order runs
// nb * 2^B words.  Forward uses (5, 5); the inverse WRITES the block-order side, where longer runs pay: (6, 4).
template <bool FWD, int A, int B>
__global__ void __launch_bounds__(256)
ntt3n_perm_tiled(const u64* in, u64* out, int N, int nb, int log_n2, const int* __restrict__ rank_of_block) {
  extern __shared__ u64 tile[];                           // [2^A j_low][2^B * nb + 1]
  const int m = log_n2, midbits = m - A - B;
  const u32 jmid = blockIdx.y;                            // 0 .. 2^midbits - 1
  const size_t base = (size_t)blockIdx.x * N;
  const int nhi = 1 << B, nlo = 1 << A;
  const int rowlen = nhi * nb + 1;
  const u32 jbmid = midbits ? (__brev(jmid) >> (32 - midbits)) : 0u;
  // global "block order" side: element (c, jhigh, jlow) at c*n2 + (jhigh << (m-B)) | (jmid << A) | jlow
  // global "rank order" side:  element at nb * ((brevA(jlow) << (m-A)) | (jbmid << B) | brevB(jhigh)) + rank[c]
  const int nseg = nb * nhi;                              // (c, jhigh) pairs
  if (FWD) {
    for (int e = threadIdx.x; e < nseg * nlo; e += 256) {
      const int seg = e >> A, jl = e & (nlo - 1), c = seg >> B, jh = seg & (nhi - 1);
      const u64 v = in[base + ((size_t)c << m) + ((size_t)jh << (m - B)) + ((size_t)jmid << A) + jl];
      tile[jl * rowlen + (int)(__brev((u32)jh) >> (32 - B)) * nb + rank_of_block[c]] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nlo * nseg; e += 256) {
      const int jl = e / nseg, w = e - jl * nseg;
      const size_t jbhi = (size_t)(__brev((u32)jl) >> (32 - A));
      out[base + (size_t)nb * ((jbhi << (m - A)) + ((size_t)jbmid << B)) + w] = tile[jl * rowlen + w];
    }
  } else {
    for (int e = threadIdx.x; e < nlo * nseg; e += 256) {
      const int jl = e / nseg, w = e - jl * nseg;
      const size_t jbhi = (size_t)(__brev((u32)jl) >> (32 - A));
      tile[jl * rowlen + w] = in[base + (size_t)nb * ((jbhi << (m - A)) + ((size_t)jbmid << B)) + w];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nseg * nlo; e += 256) {
      const int seg = e >> A, jl = e & (nlo - 1), c = seg >> B, jh = seg & (nhi - 1);
      out[base + ((size_t)c << m) + ((size_t)jh << (m - B)) + ((size_t)jmid << A) + jl] =
          tile[jl * rowlen + (int)(__brev((u32)jh) >> (32 - B)) * nb + rank_of_block[c]];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host: twiddle tree and tables
// ---------------------------------------------------------------------------------------------------------------
template <class T>
static int up(T** d, const std::vector<T>& h) {
  if (hipMalloc((void**)d, (h.size() ? h.size() : 1) * sizeof(T)) != hipSuccess) return rh_fail(RH_ERR_NOMEM, "hipMalloc failed");
  if (!h.empty() && hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return rh_fail(RH_ERR_DEVICE, "hipMemcpy failed");
  return 0;
}
static tw2 pair_of(u64 w, u64 q) { return tw2{w, rh::shoup_quotient(w, q)}; }

int rh_ring3n_setup(rh_ring* r, std::vector<LimbConsts>& hc) {
  const int N = r->N, L = r->L;
  rh_ring3n_state* s = new rh_ring3n_state();
  r->s3n = s;
  int m = N; while (m % 2 == 0) { m /= 2; s->a++; } while (m % 3 == 0) { m /= 3; s->b++; }
  s->log_n2 = s->a - 1; s->n2 = 1 << s->log_n2;
  s->nb = 2; for (int i = 0; i < s->b; ++i) s->nb *= 3;
  const int a = s->a, b = s->b, nb = s->nb, n2 = s->n2;
  const long long threeN = 3LL * N;

  // block exponents after the split and the radix-3 layers (tree of references/integer_dft.py:157-175)
  std::vector<std::vector<long long>> lvl(b + 2);
  lvl[1] = {threeN / 6, 5 * threeN / 6};
  s->r3_off.assign(b + 1, 0);
  int off = 0;
  for (int l = 1; l <= b; ++l) {
    s->r3_off[l] = off; off += 2 * (int)lvl[l].size();
    for (long long e : lvl[l]) { lvl[l + 1].push_back(e / 3); lvl[l + 1].push_back(e / 3 + N); lvl[l + 1].push_back(e / 3 + 2 * (long long)N); }
  }
  s->r3_stride = off;
  const std::vector<long long>& E = lvl[b + 1];      // nb block exponents, each divisible by n2
  std::vector<int> rank(nb), block_of_rank(nb);
  for (int c = 0; c < nb; ++c) {
    long long e0 = E[c] / n2;
    rank[c] = (int)(2 * (e0 / 6) + ((e0 % 6) == 5 ? 1 : 0));
    if (rank[c] < 0 || rank[c] >= nb) return rh_fail(RH_ERR_ARG, "3N setup: internal rank error");
    block_of_rank[rank[c]] = c;
  }

  std::vector<Limb3N> l3(L);
  std::vector<tw2> r3f((size_t)L * s->r3_stride), r3i((size_t)L * s->r3_stride);
  // sub-ring tables: virtual limb v = limb*nb + c, natural order index m+i (m = 2^s) holds w^(E(s,i)/2)
  const int Lv = L * nb;
  std::vector<tw2> fs, is, lastw(Lv);
  if (s->log_n2 >= 1) { fs.assign((size_t)Lv * n2, tw2{0, 0}); is.assign((size_t)Lv * n2, tw2{0, 0}); }
  std::vector<LimbConsts> hcv(Lv);
  std::vector<u64> pw((size_t)threeN);
  for (int i = 0; i < L; ++i) {
    const u64 q = r->moduli[i], om = r->omega3n[i] % q;
    if ((q - 1) % (u64)threeN != 0) return rh_fail(RH_ERR_MODULUS, "failed to find primitive 3N-th root: (q-1) not divisible by 3N for modulus %llu", (unsigned long long)q);
    if (rh::powmod(om, (u64)threeN, q) != 1 || rh::powmod(om, (u64)threeN / 2, q) == 1 || rh::powmod(om, (u64)threeN / 3, q) == 1)
      return rh_fail(RH_ERR_MODULUS, "omega for modulus %llu is not a primitive 3N-th root", (unsigned long long)q);
    pw[0] = 1; for (long long e = 1; e < threeN; ++e) pw[e] = rh::mulmod(pw[e - 1], om, q);
    auto W = [&](long long e) { e %= threeN; if (e < 0) e += threeN; return pw[e]; };
    const u64 z = W(N / 2), z5 = W(5LL * N / 2);
    const u64 dinv = rh::invmod_prime((z5 + q - z) % q, q), sinv = rh::invmod_prime((u64)(N / 2) % q, q);
    l3[i].zeta = pair_of(z, q); l3[i].w3 = pair_of(W(N), q);
    l3[i].inv_b1 = pair_of(rh::mulmod(dinv, sinv, q), q);
    l3[i].inv_b0z = pair_of(rh::mulmod(z, rh::mulmod(dinv, sinv, q), q), q);
    l3[i].inv_s = pair_of(sinv, q);
    hc[i].ninv_mont = rh::mform(rh::invmod_prime((u64)N % q, q), q);
    for (int l = 1; l <= b; ++l)
      for (size_t k = 0; k < lvl[l].size(); ++k) {
        const long long e = lvl[l][k] / 3;
        const size_t o = (size_t)i * s->r3_stride + s->r3_off[l] + 2 * k;
        r3f[o] = pair_of(W(e), q); r3f[o + 1] = pair_of(W(2 * e), q);
        r3i[o] = pair_of(W(-e), q); r3i[o + 1] = pair_of(W(-2 * e), q);
      }
    for (int c = 0; c < nb; ++c) {
      const int v = i * nb + c;
      hcv[v] = hc[i];
      if (s->log_n2 < 1) continue;
      tw2* tf = fs.data() + (size_t)v * n2; tw2* ti = is.data() + (size_t)v * n2;
      std::vector<long long> cur{E[c]}, nxt;
      tf[0] = pair_of(1 % q, q); ti[0] = tf[0];
      for (int st = 0; st < s->log_n2; ++st) {
        nxt.clear();
        for (size_t k = 0; k < cur.size(); ++k) {
          const long long e = cur[k] / 2;
          tf[((size_t)1 << st) + k] = pair_of(W(e), q);
          ti[((size_t)1 << st) + k] = pair_of(W(-e), q);
          nxt.push_back(e); nxt.push_back(e + threeN / 2);
        }
        cur.swap(nxt);
      }
      lastw[v] = ti[1];
    }
  }
  int rc = 0;
  if (!rc) rc = up(&s->d_l3, l3);
  if (!rc) rc = up(&s->d_r3_fwd, r3f);
  if (!rc) rc = up(&s->d_r3_inv, r3i);
  if (!rc) rc = up(&s->d_rank, rank);
  if (!rc) rc = up(&s->d_block_of_rank, block_of_rank);
  if (!rc && s->log_n2 >= 1) {
    rh_ring* sub = new rh_ring();
    s->sub = sub;
    sub->device = r->device; sub->kind = RH_RING_STANDARD; sub->N = n2; sub->logN = s->log_n2; sub->L = Lv;
    sub->inv_scale = false;
    rc = rh_std_upload_tables(sub, fs, is, nullptr, lastw);
    if (!rc) rc = rh_upload_consts(sub, hcv);
  }
  (void)a;
  return rc;
}

void rh_ring3n_teardown(rh_ring* r) {
  rh_ring3n_state* s = r->s3n;
  if (!s) return;
  void* ptrs[] = {s->d_l3, s->d_r3_fwd, s->d_r3_inv, s->d_rank, s->d_block_of_rank, s->d_tmp};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (s->sub) {
    rh_ring* b = s->sub;
    void* q[] = {b->d_consts, b->d_tw_fwd, b->d_tw_inv, b->d_twk_fwd, b->d_twk_inv, b->d_lastw};
    for (void* p : q) if (p) (void)hipFree(p);
    delete b;
  }
  delete s; r->s3n = nullptr;
}

static int ensure_tmp(rh_ring3n_state* s, size_t words) {
  if (s->tmp_words >= words) return 0;
  if (s->d_tmp) (void)hipFree(s->d_tmp);
  s->d_tmp = nullptr; s->tmp_words = 0;
  if (hipMalloc((void**)&s->d_tmp, words * 8) != hipSuccess) return rh_fail(RH_ERR_NOMEM, "hipMalloc(3N scratch, %zu words) failed", words);
  s->tmp_words = words;
  return 0;
}

// BLOCK ORDER (tuning key ntt3n_block_order, device-batched calls only): the NTT domain is kept in the layout the radix-2
// sub-transforms produce -- slot j of block c at word c*n2 + j -- instead of the Go transformer's ascending-totative order
// (ring/ntt_3n.go:82-109).  Every NTT-domain operation of the ring is coefficient-wise, so the layout is invisible to Add / Mul...
// and to NTT -> pointwise -> INTT chains (matrix_ckks.Evaluator.Mul, config 4), and it removes the permutation pass: a 2-pass
// (N = 3*2^k) transform needs a transposition SOMEWHERE (natural-order output of a column-first decomposition is a comb of
// stride 2^S1 * nb words per tile), and the only place where it costs nothing is the host boundary.  rh_ring_ntt3n_reorder converts.
// Supported for b = 1 rings with n2 >= 4096 (N >= 24576); the host-limb interface always speaks the reference order.
static void launch_pre_cols_fwd(rh_ring* r, int S1sub, dim3 g, hipStream_t st, const u64* in, u64* out, int N, const tw2* r3, int r3_stride,
                                const Limb3N* l3, const LimbConsts* c, int Lrows, const tw2* stw, int log_n2) {
  if (r->asm_tile) { rh_3n_launch_layer(false, S1sub, g.x, st, in, out, N3Layer{r3, r3_stride, l3, c, Lrows, stw, N}, r->nt_streams); return; }
  if (S1sub == 1) ntt3n_pre_cols_fwd<1><<<g, 256, 0, st>>>(in, out, N, r3, r3_stride, l3, c, Lrows, stw, log_n2);
  else if (S1sub == 2) ntt3n_pre_cols_fwd<2><<<g, 256, 0, st>>>(in, out, N, r3, r3_stride, l3, c, Lrows, stw, log_n2);
  else ntt3n_pre_cols_fwd<3><<<g, 256, 0, st>>>(in, out, N, r3, r3_stride, l3, c, Lrows, stw, log_n2);
}

static void launch_cols_post_inv(rh_ring* r, int S1sub, dim3 g, hipStream_t st, const u64* in, u64* out, int N, const tw2* r3, int r3_stride,
                                 const Limb3N* l3, const LimbConsts* c, int Lrows, const tw2* stw, int log_n2) {
  if (r->asm_tile) { rh_3n_launch_layer(true, S1sub, g.x, st, in, out, N3Layer{r3, r3_stride, l3, c, Lrows, stw, N}, r->nt_streams); return; }   // hand-scheduled body (tuning asm_tile = 0: the compiled kernel)
  if (S1sub == 1) ntt3n_cols_post_inv<1><<<g, 256, 0, st>>>(in, out, N, r3, r3_stride, l3, c, Lrows, stw, log_n2);
  else if (S1sub == 2) ntt3n_cols_post_inv<2><<<g, 256, 0, st>>>(in, out, N, r3, r3_stride, l3, c, Lrows, stw, log_n2);
  else ntt3n_cols_post_inv<3><<<g, 256, 0, st>>>(in, out, N, r3, r3_stride, l3, c, Lrows, stw, log_n2);
}

bool rh_ring3n_block_order_ok(const rh_ring* r) { const rh_ring3n_state* s = r->s3n; return s && s->b == 1 && s->sub && s->log_n2 >= LT; }

static int ntt3n_block_order_launch(rh_ring* r, const u64* in, u64* out, int npoly, int Lrows, int limb0, bool inverse) {
  rh_ring3n_state* s = r->s3n;
  const int N = r->N, nb = s->nb;
  const unsigned rows = (unsigned)npoly * (unsigned)Lrows;
  if (rows == 0) return RH_OK;
  if (s->b != 1 || !s->sub || s->log_n2 < LT) return rh_fail(RH_ERR_UNSUPPORTED, "3N block order needs N = 3 * 2^k with k >= 13");
  (void)hipGetLastError();
  hipStream_t st = rh_stream(r);
  RhCallScope scope(st);
  const LimbConsts* c = r->d_consts + limb0;
  const Limb3N* l3 = s->d_l3 + limb0;
  const tw2* r3f = s->d_r3_fwd + (size_t)limb0 * s->r3_stride;
  const tw2* r3i = s->d_r3_inv + (size_t)limb0 * s->r3_stride;
  auto chunks = [](int work) { int g = (work + 255) / 256; return g < 1 ? 1 : (g > 64 ? 64 : g); };
  const int S1sub = s->log_n2 - 12;
  const bool fuse = r->fuse3n && S1sub >= 1 && S1sub <= 3;
  const dim3 g(rows * 16);
  // every layer kernel reads and writes the same index set per thread ({i + k N/6}, resp. {c n2 + col + 4096 k}): in place is safe
  if (!inverse) {
    if (fuse) {
      const tw2* stw = s->sub->d_tw_fwd + ((size_t)limb0 * nb << s->log_n2);
      launch_pre_cols_fwd(r, S1sub, g, st, in, out, N, r3f + s->r3_off[1], s->r3_stride, l3, c, Lrows, stw, s->log_n2);
      if (int rc = rh_std_ntt_launch(s->sub, out, out, npoly, Lrows * nb, limb0 * nb, false, false, 2)) return rc;      // tile stages
    } else {
      ntt3n_pre_b1_fwd<<<dim3(rows, chunks(N / 6)), 256, 0, st>>>(in, out, N, r3f + s->r3_off[1], s->r3_stride, l3, c, Lrows);
      if (int rc = rh_std_ntt_launch(s->sub, out, out, npoly, Lrows * nb, limb0 * nb, false, false, 0)) return rc;
    }
  } else {
    if (int rc = rh_std_ntt_launch(s->sub, in, out, npoly, Lrows * nb, limb0 * nb, true, false, fuse ? 2 : 0)) return rc;
    if (fuse) {
      const tw2* stw = s->sub->d_tw_inv + ((size_t)limb0 * nb << s->log_n2);
      launch_cols_post_inv(r, S1sub, g, st, out, out, N, r3i + s->r3_off[1], s->r3_stride, l3, c, Lrows, stw, s->log_n2);
    } else {
      ntt3n_post_b1_inv<<<dim3(rows, chunks(N / 6)), 256, 0, st>>>(out, out, N, r3i + s->r3_off[1], s->r3_stride, l3, c, Lrows);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rh_fail(RH_ERR_DEVICE, "3N transform (block order) launch failed: %s", hipGetErrorString(e));
  return RH_OK;
}

// tiled permutation with tile shape (A, B) encoded as 10*A + B (block-order runs of 2^A words, rank-order runs of nb * 2^B words);
// the tile is 2^(A+B) * nb words of LDS, so smaller shapes trade run length for workgroups per CU
template <bool FWD>
static bool launch_perm_tiled(int shape, unsigned rows, int log_n2, int nb, hipStream_t st, const u64* in, u64* out, int N, const int* rank) {
  const int A = shape / 10, B = shape % 10;
  if (A < 3 || B < 3 || A + B > log_n2) return false;
  const dim3 g(rows, 1u << (log_n2 - A - B));
  const size_t lds = ((size_t)1 << A) * (((size_t)nb << B) + 1) * 8;
#define RH_PERM(a, b) case 10 * a + b: ntt3n_perm_tiled<FWD, a, b><<<g, 256, lds, st>>>(in, out, N, nb, log_n2, rank); return true
  switch (shape) {
    RH_PERM(5, 5); RH_PERM(6, 4); RH_PERM(7, 3); RH_PERM(4, 4); RH_PERM(5, 4); RH_PERM(4, 5); RH_PERM(5, 3); RH_PERM(6, 3); RH_PERM(3, 5); RH_PERM(4, 3); RH_PERM(3, 4);
  }
#undef RH_PERM
  return false;
}

// reference order <-> block order of NTT-domain data (never in place): the permutation pass on its own
int rh_ring3n_reorder_launch(rh_ring* r, const u64* in, u64* out, int npoly, int Lrows, bool to_reference) {
  rh_ring3n_state* s = r->s3n;
  const int N = r->N, nb = s->nb;
  const unsigned rows = (unsigned)npoly * (unsigned)Lrows;
  if (rows == 0) return RH_OK;
  if (in == out) return rh_fail(RH_ERR_ARG, "3N reorder cannot run in place");
  (void)hipGetLastError();
  hipStream_t st = rh_stream(r);
  auto chunks = [](int work) { int g = (work + 255) / 256; return g < 1 ? 1 : (g > 64 ? 64 : g); };
  const bool tiled = s->log_n2 >= 2 * PT && nb <= 6;
  if (to_reference) {
    if (!(tiled && launch_perm_tiled<true>(r->perm_fwd_shape, rows, s->log_n2, nb, st, in, out, N, s->d_rank)))
      ntt3n_perm_fwd<<<dim3(rows, chunks(N)), 256, 0, st>>>(in, out, N, nb, s->log_n2, s->d_block_of_rank, r->d_consts, Lrows, 0);
  } else {
    if (!(tiled && launch_perm_tiled<false>(r->perm_inv_shape, rows, s->log_n2, nb, st, in, out, N, s->d_rank)))
      ntt3n_perm_inv<<<dim3(rows, chunks(N)), 256, 0, st>>>(in, out, N, nb, s->log_n2, s->d_block_of_rank);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rh_fail(RH_ERR_DEVICE, "3N reorder launch failed: %s", hipGetErrorString(e));
  return RH_OK;
}

int rh_ring3n_ntt_launch(rh_ring* r, const u64* in, u64* out, int npoly, int Lrows, int limb0, bool inverse, bool block_order) {
  if (block_order) return ntt3n_block_order_launch(r, in, out, npoly, Lrows, limb0, inverse);
  rh_ring3n_state* s = r->s3n;
  const int N = r->N, nb = s->nb;
  const unsigned rows = (unsigned)npoly * (unsigned)Lrows;
  if (rows == 0) return RH_OK;
  u64* tmp = rh_ws_override((size_t)rows * N);            // the host-limb path brings its own workspace (one row)
  if (!tmp) { if (int rc = ensure_tmp(s, (size_t)rows * N)) return rc; tmp = s->d_tmp; }
  (void)hipGetLastError();
  hipStream_t st = rh_stream(r);
  RhCallScope scope(st);                                  // the sub-ring's launches go to the same stream
  const LimbConsts* c = r->d_consts + limb0;
  const Limb3N* l3 = s->d_l3 + limb0;
  const tw2* r3f = s->d_r3_fwd + (size_t)limb0 * s->r3_stride;
  const tw2* r3i = s->d_r3_inv + (size_t)limb0 * s->r3_stride;
  auto chunks = [](int work) { int g = (work + 255) / 256; return g < 1 ? 1 : (g > 64 ? 64 : g); };
  const bool tiled = s->log_n2 >= 2 * PT && nb <= 6;     // LDS: 2^A * (2^B * nb + 1) * 8 bytes (48 KiB for (5, 5) at nb = 6)
  if (!inverse) {
    const int S1sub = s->sub ? s->log_n2 - 12 : 0;
    const bool fuse = s->b == 1 && s->sub && r->fuse3n && S1sub >= 1 && S1sub <= 3;    // 6 * 2^S1 coefficients per thread
    if (fuse) {
      const tw2* stw = s->sub->d_tw_fwd + ((size_t)limb0 * nb << s->log_n2);
      const dim3 g(rows * 16);
      launch_pre_cols_fwd(r, S1sub, g, st, in, tmp, N, r3f + s->r3_off[1], s->r3_stride, l3, c, Lrows, stw, s->log_n2);
      if (int rc = rh_std_ntt_launch(s->sub, tmp, tmp, npoly, Lrows * nb, limb0 * nb, false, false, 2)) return rc;   // tile stages only
    } else if (s->b == 1) {
      ntt3n_pre_b1_fwd<<<dim3(rows, chunks(N / 6)), 256, 0, st>>>(in, tmp, N, r3f + s->r3_off[1], s->r3_stride, l3, c, Lrows);
    } else {
      ntt3n_split_fwd<<<dim3(rows, chunks(N / 2)), 256, 0, st>>>(in, tmp, N, l3, c, Lrows);
      int step = N / 6, cnt = 2;
      for (int l = 1; l <= s->b; ++l, cnt *= 3, step /= 3)
        ntt3n_radix3_fwd<<<dim3(rows, chunks(cnt * step)), 256, 0, st>>>(tmp, N, step, cnt, r3f + s->r3_off[l], s->r3_stride, l3, c, Lrows);
    }
    if (s->sub && !fuse) {
      // (poly, limb, block) rows of length n2: limb-major virtual limb index = limb*nb + c
      if (int rc = rh_std_ntt_launch(s->sub, tmp, tmp, npoly, Lrows * nb, limb0 * nb, false, false, 0)) return rc;
    }
    if (!(tiled && launch_perm_tiled<true>(r->perm_fwd_shape, rows, s->log_n2, nb, st, tmp, out, N, s->d_rank)))
      ntt3n_perm_fwd<<<dim3(rows, chunks(N)), 256, 0, st>>>(tmp, out, N, nb, s->log_n2, s->d_block_of_rank, c, Lrows, s->sub ? 0 : 1);
  } else {
    if (!(tiled && launch_perm_tiled<false>(r->perm_inv_shape, rows, s->log_n2, nb, st, in, tmp, N, s->d_rank)))
      ntt3n_perm_inv<<<dim3(rows, chunks(N)), 256, 0, st>>>(in, tmp, N, nb, s->log_n2, s->d_block_of_rank);
    const int S1sub = s->sub ? s->log_n2 - 12 : 0;
    const bool fuse = s->b == 1 && s->sub && r->fuse3n && S1sub >= 1 && S1sub <= 3;
    if (s->sub) {
      if (int rc = rh_std_ntt_launch(s->sub, tmp, tmp, npoly, Lrows * nb, limb0 * nb, true, false, fuse ? 2 : 0)) return rc;   // fused: tile stages only
    }
    if (fuse) {
      const tw2* stw = s->sub->d_tw_inv + ((size_t)limb0 * nb << s->log_n2);
      const dim3 g(rows * 16);
      launch_cols_post_inv(r, S1sub, g, st, tmp, out, N, r3i + s->r3_off[1], s->r3_stride, l3, c, Lrows, stw, s->log_n2);
    } else if (s->b == 1) {
      ntt3n_post_b1_inv<<<dim3(rows, chunks(N / 6)), 256, 0, st>>>(tmp, out, N, r3i + s->r3_off[1], s->r3_stride, l3, c, Lrows);
    } else {
      int cnt = nb / 3, step = s->n2;
      for (int l = s->b; l >= 1; --l, cnt /= 3, step *= 3)
        ntt3n_radix3_inv<<<dim3(rows, chunks(cnt * step)), 256, 0, st>>>(tmp, N, step, cnt, r3i + s->r3_off[l], s->r3_stride, l3, c, Lrows);
      ntt3n_split_inv<<<dim3(rows, chunks(N / 2)), 256, 0, st>>>(tmp, out, N, l3, c, Lrows);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rh_fail(RH_ERR_DEVICE, "3N transform launch failed: %s", hipGetErrorString(e));
  return RH_OK;
}

// tuning keys that must reach the radix-2 sub-ring as well (its tile stages choose their cache policy by size like any standard ring)
void rh_ring3n_set_nt_streams(rh_ring* r, bool on) { if (r->s3n && r->s3n->sub) r->s3n->sub->nt_streams = on; }

int rh_ring3n_reserve(rh_ring* r, int npoly) { return ensure_tmp(r->s3n, (size_t)npoly * r->L * r->N); }
