// ntt_kernels.hip.hpp -- negacyclic NTT / INTT over Z_q[X]/(X^N+1) for gfx950.
//
// Replaces ring/ntt.go: nttCoreLazy (:209-552), inttCoreLazy (:554-714) and the NTTStandard*/INTTStandard* wrappers
// (:174-206) of the reference.  Data layout in HBM: (poly, limb, coefficient) contiguous u64, one launch covers a
// whole batch.  Stage numbering follows the reference: forward stage s has m = 2^s blocks of 2t = N/2^s coefficients
// and block i uses RootsForward[m+i] (:240-255); inverse stages run t = 1,2,4.. and use RootsBackward[h+i] (:590-605).
//
// Decomposition for N >= 4096 (LT = 12):
//   forward:  K1 = the first S1 = log2(N)-12 stages, register radix-2^S1, each thread owns one column
//             {c + 4096*k}: every load/store instruction of a wave is one contiguous 512 B segment; twiddles are
//             wave-uniform (scalar loads).
//             K2 = the last 12 stages on one contiguous 4096-coefficient tile (32 KiB) per 256-thread workgroup:
//             three register radix-16 rounds, two exchanges through LDS (padded: bank-conflict free), a final LDS
//             transpose so global stores are contiguous.  Twiddles come from a per-limb table already permuted
//             into "kernel order" (engine.cpp: build_kernel_order) so every twiddle load is lane-contiguous.
//   inverse:  the mirror image (K2 first, then K1 with N^-1 folded into the last stage's twiddles).
// N < 4096: one workgroup per limb, all stages in LDS (ntt_small_*).
//
// Arithmetic policies (modarith.hip.hpp):
//   ShoupPolicy  -- fast path; values kept < 8q, outputs canonical.  Used by Forward, Backward, BackwardLazy
//                   (all canonical in the reference for N >= 16).
//   MontPolicy   -- the reference's MRedLazy butterfly with its reduce schedule (:315-318, :500-517); reproduces
//                   ForwardLazy's exact representatives.
#pragma once
#include "modarith.hip.hpp"

#include "ring_types.hip.hpp"

// ---------------------------------------------------------------------------------------------------------------
// policies
// ---------------------------------------------------------------------------------------------------------------
struct ShoupPolicy {
  typedef tw2 tw_t;
  u64 q, nq, q4;
  RH_DEV void init(const LimbConsts& c) { q = c.q; nq = c.nq; q4 = 4 * c.q; }
  // forward: U < 8q, V any -> X,Y < 8q
  RH_DEV void fwd(u64& U, u64& V, const tw_t& w, bool /*reduce*/) const {
    u64 u = csub(U, q4);
    u64 X = shoup_mul_acc(V, w.w, w.wp, nq, u);      // u + r, r < 4q
    V = ((u << 1) + q4) - X;                         // u + 4q - r
    U = X;
  }
  // inverse: U,V < 4q -> X,Y < 4q
  RH_DEV void inv(u64& U, u64& V, const tw_t& w) const {
    u64 d = U + q4 - V;
    U = csub(U + V, q4);
    V = shoup_mul(d, w.w, w.wp, nq);
  }
  RH_DEV u64 fwd_final(u64 x, bool canonical) const { return canonical ? canon8(x, q) : x; }
};

struct MontPolicy {
  typedef u64 tw_t;
  u64 q, qinv, q2, q4;
  RH_DEV void init(const LimbConsts& c) { q = c.q; qinv = c.qinv; q2 = 2 * c.q; q4 = 4 * c.q; }
  RH_DEV void fwd(u64& U, u64& V, const tw_t& w, bool reduce) const {     // butterfly, ring/ntt.go:155-161
    u64 u = U;
    if (reduce) { if (u >= q4) u -= q4; }
    u64 r = mred_lazy(V, w, q, qinv);
    U = u + r; V = u + q2 - r;
  }
  RH_DEV void inv(u64& U, u64& V, const tw_t& w) const {                  // invbutterfly, ring/ntt.go:164-171
    u64 x = U + V;
    if (x >= q2) x -= q2;
    V = mred_lazy(U + q4 - V, w, q, qinv);
    U = x;
  }
  RH_DEV u64 fwd_final(u64 x, bool) const { return x; }
};

// reduce schedule of the reference for forward stage s (ring/ntt.go:223-257 for N<16, :271-310, :315-318, :500-517)
RH_DEV bool ref_reduce(int s, int logN) {
  if (logN < 4) return true;
  if (s == 0) return false;
  if (s == logN - 1) return true;
  return (s & 1) == 0;          // bit-length(2^s) = s+1 odd
}

// one register radix-16 round: 4 stages over x[16]; stage u pairs (k, k + (8>>u)); twiddle slot (2^u - 1) + (k >> (4-u)).
// TW(slot) yields the twiddle.  s0 = global index of the round's first stage (for the reduce schedule).
template <class P, class TWF>
RH_DEV void round16_fwd(const P& p, u64 (&x)[16], TWF TW, int s0, int logN) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int h = 8 >> u;
    const bool red = ref_reduce(s0 + u, logN);
#pragma unroll
    for (int g = 0; g < (1 << u); ++g) {
      typename P::tw_t w = TW((1 << u) - 1 + g);
#pragma unroll
      for (int e = 0; e < h; ++e) {
        const int k = g * 2 * h + e;
        p.fwd(x[k], x[k + h], w, red);
      }
    }
  }
}
template <class P, class TWF>
RH_DEV void round16_inv(const P& p, u64 (&x)[16], TWF TW) {
#pragma unroll
  for (int u = 3; u >= 0; --u) {
    const int h = 8 >> u;
#pragma unroll
    for (int g = 0; g < (1 << u); ++g) {
      typename P::tw_t w = TW((1 << u) - 1 + g);
#pragma unroll
      for (int e = 0; e < h; ++e) {
        const int k = g * 2 * h + e;
        p.inv(x[k], x[k + h], w);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K2 forward: last 12 stages on a 4096-tile.  twk: kernel-order table, per limb N entries, per tile 4096 entries:
//   [0,15) round A slots (uniform) | [16,256) round B: slot*16 + hi4 | [256,4096) round C: slot*256 + tid
// ---------------------------------------------------------------------------------------------------------------
template <class P>
RH_DEV void fwd_tile_body(u64* lds, const u32 b, const u64* in, u64* out, const typename P::tw_t* __restrict__ twk,
                          const LimbConsts* __restrict__ consts, int L, int logN, int canonical, int npoly, int Ls, int Lso = 0) {
  if (Lso == 0) Lso = Ls;                  // rows per poly of the OUTPUT block when it differs from the input's (AtLevel views with two strides)
  const int tid = threadIdx.x;
  const u32 limb = b % (u32)L;
  const u32 r = b / (u32)L;
  const int S1 = logN - LT;
  // poly is the fast index: workgroups resident at the same time work on the SAME tile of different polys, so that
  // tile's twiddles (64 KiB per limb) are served by the XCD's L2 instead of being re-fetched per poly
  const u32 poly = r % (u32)npoly;
  const u32 tile = r / (u32)npoly;
  const size_t base = (((size_t)poly * Ls + limb) << logN) + ((size_t)tile << LT);   // Ls = rows per poly of the block (>= L)
  const size_t obase = (((size_t)poly * Lso + limb) << logN) + ((size_t)tile << LT);
  const typename P::tw_t* tw = twk + ((size_t)limb << logN) + ((size_t)tile << LT);
  P p; p.init(consts[limb]);

  u64 x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = in[base + tid + 256 * k];
  // round A: in-tile bits 11..8 (k); twiddles wave-uniform
  round16_fwd(p, x, [&](int slot) { return tw[slot]; }, S1, logN);
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[LDS_PAD(tid + 256 * k)] = x[k];
  __syncthreads();
  // round B: j = (hi4<<8) | (k<<4) | lo4
  const int hi4 = tid >> 4, lo4 = tid & 15;
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = lds[LDS_PAD((hi4 << 8) | (k << 4) | lo4)];
  round16_fwd(p, x, [&](int slot) { return tw[16 + slot * 16 + hi4]; }, S1 + 4, logN);
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[LDS_PAD((hi4 << 8) | (k << 4) | lo4)] = x[k];
  __syncthreads();
  // round C: j = tid*16 + k
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = lds[LDS_PAD(tid * 16 + k)];
  round16_fwd(p, x, [&](int slot) { return tw[256 + slot * 256 + tid]; }, S1 + 8, logN);
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[LDS_PAD(tid * 16 + k)] = p.fwd_final(x[k], canonical != 0);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) out[obase + tid + 256 * k] = lds[LDS_PAD(tid + 256 * k)];
}
template <class P>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))      // LDS allows 4 workgroups per CU: keep the registers within that
ntt_fwd_tile(const u64* in, u64* out, const typename P::tw_t* __restrict__ twk,
             const LimbConsts* __restrict__ consts, int L, int logN, int canonical, int npoly, int Ls, int Lso) {
  __shared__ u64 lds[LDS_WORDS];
  fwd_tile_body<P>(lds, blockIdx.x, in, out, twk, consts, L, logN, canonical, npoly, Ls, Lso);
}

// Forward tile stages with a fused epilogue: out = MRed(2q - y + NTT(in), s_limb) -- the subtract-multiply that follows a
// forward transform in ModDownQPtoQNTT (ring/basis_extension.go:255-257, SubThenMulScalarMontgomeryTwoModulus) and in
// DivRoundByLastModulusNTT (ring/scaling.go:120-124).  The canonical NTT values never go to memory: the element-wise
// pass (24 B per coefficient) disappears.  y and out are (poly, limb) blocks with their own row counts.
struct LimbScalars { u64 s[RH_MAX_LIMBS_K]; };
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))      // <= 128 VGPRs: 4 workgroups per CU like the asm body
ntt_fwd_tile_submul(const u64* in, const tw2* __restrict__ twk, const LimbConsts* __restrict__ consts, int L, int logN, int npoly,
                    const u64* y, int y_rows, u64* out, int out_rows, LimbScalars sc, const u64* z, int z_rows) {   // z != null: out = CRed(... + z) (the ring.Add that follows a key switch)
  __shared__ u64 lds[LDS_WORDS];
  const int tid = threadIdx.x;
  const u32 b = blockIdx.x;
  const u32 limb = b % (u32)L;
  const u32 r = b / (u32)L;
  const int S1 = logN - LT;
  const u32 poly = r % (u32)npoly;
  const u32 tile = r / (u32)npoly;
  const size_t base = (((size_t)poly * L + limb) << logN) + ((size_t)tile << LT);
  const tw2* tw = twk + ((size_t)limb << logN) + ((size_t)tile << LT);
  const LimbConsts c = consts[limb];
  ShoupPolicy p; p.init(c);
  u64 x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = in[base + tid + 256 * k];
  round16_fwd(p, x, [&](int slot) { return tw[slot]; }, S1, logN);
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[LDS_PAD(tid + 256 * k)] = x[k];
  __syncthreads();
  const int hi4 = tid >> 4, lo4 = tid & 15;
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = lds[LDS_PAD((hi4 << 8) | (k << 4) | lo4)];
  round16_fwd(p, x, [&](int slot) { return tw[16 + slot * 16 + hi4]; }, S1 + 4, logN);
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[LDS_PAD((hi4 << 8) | (k << 4) | lo4)] = x[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = lds[LDS_PAD(tid * 16 + k)];
  round16_fwd(p, x, [&](int slot) { return tw[256 + slot * 256 + tid]; }, S1 + 8, logN);
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[LDS_PAD(tid * 16 + k)] = p.fwd_final(x[k], true);
  __syncthreads();
  const size_t yb = (((size_t)poly * y_rows + limb) << logN) + ((size_t)tile << LT);
  const size_t ob = (((size_t)poly * out_rows + limb) << logN) + ((size_t)tile << LT);
  const u64 sl = sc.s[limb], q2 = 2 * c.q;
  const size_t zb = (((size_t)poly * z_rows + limb) << logN) + ((size_t)tile << LT);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const u64 v = lds[LDS_PAD(tid + 256 * k)];
    u64 rr = mred(q2 - y[yb + tid + 256 * k] + v, sl, c.q, c.qinv);
    if (z) rr = cred(rr + z[zb + tid + 256 * k], c.q);
    out[ob + tid + 256 * k] = rr;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K2 inverse: first 12 stages (t = 1..2048) on a 4096-tile.  If `last` (logN == 12) the N^-1 scaling and the
// canonical reduction happen here, else values leave < 4q for K1 inverse.
// ---------------------------------------------------------------------------------------------------------------
// MUL: the input is the pointwise product of two NTT-domain blocks, formed on load as MRedLazy(in, in2) = in*in2*2^-64 (< 2q for
// inputs < 2q): INTT(a . b) without the element-wise passes of MForm + MulCoeffsMontgomery (schemes/ckks/evaluator.go:821-834
// followed by INTT).  The missing factor 2^64 rides in the N^-1 constants of the LAST inverse stage (the caller passes the
// 2^64-scaled constant sets: consts here when `last`, else to the column kernel), so the canonical result is the same residue.
template <bool MUL>
RH_DEV void inv_tile_body(u64* lds, const u32 b, const u64* in, const u64* in2, u64* out, const tw2* __restrict__ twk,
                          const LimbConsts* __restrict__ consts, int L, int logN, int last, int npoly, int in_Ls = 0, int out_Ls = 0) {
  const int tid = threadIdx.x;
  const u32 limb = b % (u32)L;
  const u32 r = b / (u32)L;
  // poly is the fast index: workgroups resident at the same time work on the SAME tile of different polys, so that
  // tile's twiddles (64 KiB per limb) are served by the XCD's L2 instead of being re-fetched per poly
  const u32 poly = r % (u32)npoly;
  const u32 tile = r / (u32)npoly;
  const size_t base = (((size_t)poly * (in_Ls ? in_Ls : L) + limb) << logN) + ((size_t)tile << LT);      // in_Ls / out_Ls: rows per poly of the blocks (0: L)
  const size_t obase = (((size_t)poly * (out_Ls ? out_Ls : L) + limb) << logN) + ((size_t)tile << LT);
  const tw2* tw = twk + ((size_t)limb << logN) + ((size_t)tile << LT);
  const LimbConsts c = consts[limb];
  ShoupPolicy p; p.init(c);

  u64 x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    u64 v = in[base + tid + 256 * k];
    if (MUL) v = mred_lazy(v, in2[base + tid + 256 * k], c.q, c.qinv);
    lds[LDS_PAD(tid + 256 * k)] = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = lds[LDS_PAD(tid * 16 + k)];
  round16_inv(p, x, [&](int slot) { return tw[256 + slot * 256 + tid]; });
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[LDS_PAD(tid * 16 + k)] = x[k];
  __syncthreads();
  const int hi4 = tid >> 4, lo4 = tid & 15;
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = lds[LDS_PAD((hi4 << 8) | (k << 4) | lo4)];
  round16_inv(p, x, [&](int slot) { return tw[16 + slot * 16 + hi4]; });
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[LDS_PAD((hi4 << 8) | (k << 4) | lo4)] = x[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = lds[LDS_PAD(tid + 256 * k)];
  round16_inv(p, x, [&](int slot) { return tw[slot]; });
  if (last) {
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = canon4(shoup_mul(x[k], c.ninv_w, c.ninv_wp, c.nq), c.q);
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) out[obase + tid + 256 * k] = x[k];
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
ntt_inv_tile(const u64* in, u64* out, const tw2* __restrict__ twk,
             const LimbConsts* __restrict__ consts, int L, int logN, int last, int npoly, int in_Ls = 0, int out_Ls = 0) {
  __shared__ u64 lds[LDS_WORDS];
  inv_tile_body<false>(lds, blockIdx.x, in, nullptr, out, twk, consts, L, logN, last, npoly, in_Ls, out_Ls);
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
ntt_inv_tile_mul(const u64* in, const u64* in2, u64* out, const tw2* __restrict__ twk,
                 const LimbConsts* __restrict__ consts, int L, int logN, int last, int npoly) {
  __shared__ u64 lds[LDS_WORDS];
  inv_tile_body<true>(lds, blockIdx.x, in, in2, out, twk, consts, L, logN, last, npoly);
}

// ---------------------------------------------------------------------------------------------------------------
// K1 forward: first S1 stages, R = 2^S1 coefficients per thread at stride 4096.  twn: natural-order table
// (RootsForward index), entries [1, R) are used and are wave-uniform.
// ---------------------------------------------------------------------------------------------------------------
template <class P, int S1>
RH_DEV void fwd_cols_body(const u32 b, const u64* in, u64* out, const typename P::tw_t* __restrict__ twn,
                          const LimbConsts* __restrict__ consts, int L, int logN, int Ls = 0, int Lso = 0) {
  if (Ls == 0) Ls = L;                     // rows per poly of the block (>= L)
  if (Lso == 0) Lso = Ls;                  // ... of the output block, when it differs
  constexpr int R = 1 << S1;
  const u32 limb = b % (u32)L;
  const u32 r = b / (u32)L;
  const u32 cb = r & 15;            // 16 blocks of 256 columns per limb
  const u32 poly = r >> 4;
  const size_t base = (((size_t)poly * Ls + limb) << logN) + cb * 256 + threadIdx.x;
  const size_t obase = (((size_t)poly * Lso + limb) << logN) + cb * 256 + threadIdx.x;
  const typename P::tw_t* tw = twn + ((size_t)limb << logN);
  P p; p.init(consts[limb]);
  u64 x[R];
#pragma unroll
  for (int k = 0; k < R; ++k) x[k] = in[base + ((size_t)k << LT)];
#pragma unroll
  for (int s = 0; s < S1; ++s) {
    const int h = R >> (s + 1);
    const bool red = ref_reduce(s, logN);
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      typename P::tw_t w = tw[(1 << s) + g];
#pragma unroll
      for (int e = 0; e < h; ++e) p.fwd(x[g * 2 * h + e], x[g * 2 * h + e + h], w, red);
    }
  }
#pragma unroll
  for (int k = 0; k < R; ++k) out[obase + ((size_t)k << LT)] = x[k];
}
template <class P, int S1>
__global__ void __launch_bounds__(256)
ntt_fwd_cols(const u64* in, u64* out, const typename P::tw_t* __restrict__ twn,
             const LimbConsts* __restrict__ consts, int L, int logN, int Ls, int Lso) {
  fwd_cols_body<P, S1>(blockIdx.x, in, out, twn, consts, L, logN, Ls, Lso);
}
// Column stages fed by the re-expansion of a rescale step (ring/scaling.go:97-118): x = (t [+ h, recentred]) mod q_limb is
// computed on the fly from the coefficient-domain last limb t (one row per poly, L2-resident across the limbs) instead of
// being written to a buffer and read back.  Same arithmetic as rescale_expand_kernel followed by fwd_cols_body.
template <int S1>
__global__ void __launch_bounds__(256)
ntt_fwd_cols_expand(const u64* tmp, u64* out, const tw2* __restrict__ twn, const LimbConsts* __restrict__ consts,
                    const RescaleLimb* __restrict__ T, int L, int logN, int mode, u64 qL) {
  constexpr int R = 1 << S1;
  const u32 b = blockIdx.x;
  const u32 limb = b % (u32)L;
  const u32 r = b / (u32)L;
  const u32 cb = r & 15;
  const u32 poly = r >> 4;
  const size_t col = cb * 256 + threadIdx.x;
  const size_t base = (((size_t)poly * L + limb) << logN) + col;
  const tw2* tw = twn + ((size_t)limb << logN);
  const RescaleLimb l = T[limb];
  ShoupPolicy p; p.init(consts[limb]);
  u64 x[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    u64 t = tmp[((size_t)poly << logN) + col + ((size_t)k << LT)];
    if (mode == 1) t = cred(t + ((qL - 1) >> 1), qL) + l.s;              // AddScalarLazy (:104)
    x[k] = bred_add(t, l.q, l.bred0);
  }
#pragma unroll
  for (int s = 0; s < S1; ++s) {
    const int h = R >> (s + 1);
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      tw2 w = tw[(1 << s) + g];
#pragma unroll
      for (int e = 0; e < h; ++e) p.fwd(x[g * 2 * h + e], x[g * 2 * h + e + h], w, false);
    }
  }
#pragma unroll
  for (int k = 0; k < R; ++k) out[base + ((size_t)k << LT)] = x[k];
}

// Fused launch of a software pipeline over spans of polys: the workgroup first runs the (HBM-bound) column stages of
// one unit of span j, then the (VALU-bound) tile stages of one tile of span j-1, so that on every CU memory-phase and
// compute-phase workgroups are co-resident.  n1/n2 = number of column units / tiles in this launch.
template <class P, int S1>
__global__ void __launch_bounds__(256)
ntt_fwd_fused(const u64* in1, u64* out1, unsigned n1, u64* data2, unsigned n2, int npoly2,
              const typename P::tw_t* __restrict__ twn, const typename P::tw_t* __restrict__ twk,
              const LimbConsts* __restrict__ consts, int L, int logN, int canonical) {
  __shared__ u64 lds[LDS_WORDS];
  if (blockIdx.x < n1) fwd_cols_body<P, S1>(blockIdx.x, in1, out1, twn, consts, L, logN);
  if (blockIdx.x < n2) fwd_tile_body<P>(lds, blockIdx.x, data2, data2, twk, consts, L, logN, canonical, npoly2, L);
}

// K1 inverse: last S1 stages (t = 4096 .. N/2), in natural-order RootsBackward indexing, then N^-1 and canonical
// reduction.  The last stage (h = 1) folds N^-1 into both outputs: X = (U+V)*ninv, Y = (U-V)*(psi*ninv);
// lastw = Shoup pair of psi_bwd[1]*N^-1.
template <int S1>
RH_DEV void inv_cols_body(const u32 b, u64* data, const tw2* __restrict__ twn, const tw2* __restrict__ lastw,
                          const LimbConsts* __restrict__ consts, int L, int logN, int scale, int Ls = 0) {
  constexpr int R = 1 << S1;
  const u32 limb = b % (u32)L;
  const u32 r = b / (u32)L;
  const u32 cb = r & 15;
  const u32 poly = r >> 4;
  const size_t base = (((size_t)poly * (Ls ? Ls : L) + limb) << logN) + cb * 256 + threadIdx.x;          // Ls: rows per poly of the block (0: L)
  const tw2* tw = twn + ((size_t)limb << logN);
  const LimbConsts c = consts[limb];
  ShoupPolicy p; p.init(c);
  u64 x[R];
#pragma unroll
  for (int k = 0; k < R; ++k) x[k] = data[base + ((size_t)k << LT)];
#pragma unroll
  for (int s = S1 - 1; s >= 1; --s) {          // stage with h = 2^s blocks
    const int hh = R >> (s + 1);
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      tw2 w = tw[(1 << s) + g];
#pragma unroll
      for (int e = 0; e < hh; ++e) p.inv(x[g * 2 * hh + e], x[g * 2 * hh + e + hh], w);
    }
  }
  if (scale) {
    const tw2 wl = lastw[limb];
    constexpr int hh = R >> 1;
#pragma unroll
    for (int e = 0; e < hh; ++e) {
      u64 U = x[e], V = x[e + hh];
      u64 d = U + p.q4 - V;
      x[e] = canon4(shoup_mul(U + V, c.ninv_w, c.ninv_wp, c.nq), c.q);
      x[e + hh] = canon4(shoup_mul(d, wl.w, wl.wp, c.nq), c.q);
    }
  } else {       // plain last stage (h = 1), values stay < 4q (used by the 3N transform, which scales later)
    const tw2 w1 = tw[1];
    constexpr int hh = R >> 1;
#pragma unroll
    for (int e = 0; e < hh; ++e) p.inv(x[e], x[e + hh], w1);
  }
#pragma unroll
  for (int k = 0; k < R; ++k) data[base + ((size_t)k << LT)] = x[k];
}
template <int S1>
__global__ void __launch_bounds__(256)
ntt_inv_cols(u64* data, const tw2* __restrict__ twn, const tw2* __restrict__ lastw,
             const LimbConsts* __restrict__ consts, int L, int logN, int scale, int Ls = 0) {
  inv_cols_body<S1>(blockIdx.x, data, twn, lastw, consts, L, logN, scale, Ls);
}

// ---------------------------------------------------------------------------------------------------------------
// small rings (16 <= N <= 2048, also usable up to 4096): one workgroup per (poly, limb), all stages in LDS, natural-
// order tables.  Not a throughput path (configs use N >= 4096); it exists so every ring size the reference's tests
// use (N = 16..512, ring/ntt_test.go) runs on the device.
// ---------------------------------------------------------------------------------------------------------------
template <class P>
__global__ void __launch_bounds__(256)
ntt_fwd_small(const u64* in, u64* out, const typename P::tw_t* __restrict__ twn,
              const LimbConsts* __restrict__ consts, int L, int logN, int canonical, int Ls = 0, int Lso = 0) {   // Ls / Lso: rows per poly of the input / output block (0: L)
  __shared__ u64 lds[TILE];
  const int N = 1 << logN;
  const u32 limb = blockIdx.x % (u32)L, poly = blockIdx.x / (u32)L;
  const size_t base = ((size_t)poly * (Ls ? Ls : L) + limb) << logN, obase = ((size_t)poly * (Lso ? Lso : (Ls ? Ls : L)) + limb) << logN;
  const typename P::tw_t* tw = twn + ((size_t)limb << logN);
  P p; p.init(consts[limb]);
  for (int j = threadIdx.x; j < N; j += blockDim.x) lds[j] = in[base + j];
  __syncthreads();
  for (int s = 0; s < logN; ++s) {
    const int lt = logN - 1 - s;           // log2(t)
    const bool red = ref_reduce(s, logN);
    for (int bf = threadIdx.x; bf < (N >> 1); bf += blockDim.x) {
      const int i = bf >> lt;
      const int j = (i << (lt + 1)) + (bf & ((1 << lt) - 1));
      u64 U = lds[j], V = lds[j + (1 << lt)];
      p.fwd(U, V, tw[(1 << s) + i], red);
      lds[j] = U; lds[j + (1 << lt)] = V;
    }
    __syncthreads();
  }
  for (int j = threadIdx.x; j < N; j += blockDim.x) out[obase + j] = p.fwd_final(lds[j], canonical != 0);
}

__global__ void __launch_bounds__(256)
ntt_inv_small(const u64* in, u64* out, const tw2* __restrict__ twn,
              const LimbConsts* __restrict__ consts, int L, int logN, int scale, int Ls = 0, int Lso = 0) {   // Ls / Lso: rows per poly of the input / output block (0: L)
  __shared__ u64 lds[TILE];
  const int N = 1 << logN;
  const u32 limb = blockIdx.x % (u32)L, poly = blockIdx.x / (u32)L;
  const size_t base = ((size_t)poly * (Ls ? Ls : L) + limb) << logN, obase = ((size_t)poly * (Lso ? Lso : (Ls ? Ls : L)) + limb) << logN;
  const tw2* tw = twn + ((size_t)limb << logN);
  const LimbConsts c = consts[limb];
  ShoupPolicy p; p.init(c);
  for (int j = threadIdx.x; j < N; j += blockDim.x) lds[j] = in[base + j];
  __syncthreads();
  for (int lt = 0; lt < logN; ++lt) {      // t = 2^lt, h = N/(2t)
    const int h = N >> (lt + 1);
    for (int bf = threadIdx.x; bf < (N >> 1); bf += blockDim.x) {
      const int i = bf >> lt;
      const int j = (i << (lt + 1)) + (bf & ((1 << lt) - 1));
      u64 U = lds[j], V = lds[j + (1 << lt)];
      p.inv(U, V, tw[h + i]);
      lds[j] = U; lds[j + (1 << lt)] = V;
    }
    __syncthreads();
  }
  for (int j = threadIdx.x; j < N; j += blockDim.x)
    out[obase + j] = scale ? canon4(shoup_mul(lds[j], c.ninv_w, c.ninv_wp, c.nq), c.q) : lds[j];
}

// N < 16 (the reference accepts N = 8, ring/ring.go:318): BackwardLazy is NOT canonical there -- inttCoreLazy followed by
// MRedLazy(x, NInv) (ring/ntt.go:197-202), values in [0, 2q).  Reproduced with the reference's own Montgomery butterfly so the
// representatives are bit-identical.  twm: RootsBackward as handed over (Montgomery form), natural order.
__global__ void __launch_bounds__(64)
ntt_inv_small_lazy_mont(const u64* in, u64* out, const u64* __restrict__ twm, const LimbConsts* __restrict__ consts, int L, int logN) {
  __shared__ u64 lds[16];
  const int N = 1 << logN;
  const u32 limb = blockIdx.x % (u32)L;
  const size_t base = (size_t)blockIdx.x << logN;
  const u64* tw = twm + ((size_t)limb << logN);
  const LimbConsts c = consts[limb];
  MontPolicy p; p.init(c);
  if ((int)threadIdx.x < N) lds[threadIdx.x] = in[base + threadIdx.x];
  __syncthreads();
  for (int lt = 0; lt < logN; ++lt) {
    const int h = N >> (lt + 1);
    if ((int)threadIdx.x < (N >> 1)) {
      const int bf = threadIdx.x, i = bf >> lt, j = (i << (lt + 1)) + (bf & ((1 << lt) - 1));
      u64 U = lds[j], V = lds[j + (1 << lt)];
      p.inv(U, V, tw[h + i]);
      lds[j] = U; lds[j + (1 << lt)] = V;
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < N) out[base + threadIdx.x] = mred_lazy(lds[threadIdx.x], c.ninv_mont, c.q, c.qinv);
}
