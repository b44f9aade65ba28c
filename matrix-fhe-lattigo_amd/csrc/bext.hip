// bext.hip -- RNS basis extension on device-resident (poly, limb, coefficient) blocks.
//
// Replaces ring/basis_extension.go: GenModUpConstants (:93-164), genmodDownConstants (:25-49), ModUpQtoP/ModUpPtoQ
// (:188-217), ModDownQPtoQ (:223-234), ModDownQPtoQNTT (:241-258), ModDownQPtoP (:264-278), ModUpExact with
// reconstructRNS + multSum (:282-308, :550-673) and Decomposer.DecomposeAndSplit (:381-548).
//
// One kernel (bext_kernel) does the work of reconstructRNS(+Centered) and multSum: one thread per coefficient,
//   y_i = MRed(x_i [+ half_i], (Q/q_i)^-1),  v = trunc(sum_i double(y_i)/double(q_i))   (sequential IEEE adds),
//   out_j = (sum_i y_i * (Q/q_i mod p_j)) reduced once (lazy Montgomery) + vtimesqmodp[j][v]       -- NOT canonical,
// then the caller's post step (centred subtraction, ModDown's fused subtract-multiply).  The non-canonical values are
// reproduced exactly, so results are bit-identical to the reference even before its later Reduce.
// The y_i live in registers (kernel instantiated per source-limb count up to 8, bounded variants for 16 / 32; a generic
// fallback keeps them in LDS [limb][thread]).
// Traffic: 8*(nsrc + ntgt) bytes per coefficient; nsrc*ntgt 64x64->128 multiply-accumulates per coefficient.
#include <hip/hip_runtime.h>
#include <vector>
#include <map>
#include <array>
#include <cstring>
#include "engine_internal.hpp"
#include "bext_internal.hpp"
#include "hostmath.hpp"

// acc += a * b with the carry out of the 64-bit accumulator counted in cnt; b wave-uniform (a scalar-loaded constant)
RH_DEV void mac_carry(u64& acc, u32& cnt, u32 a, u32 b) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(cnt) : "v"(a), "s"(b) : "vcc");
}

// NS: compile-time bound on the source-limb count (y_i live in registers, loops fully unrolled); EXACT: nsrc == NS.
// NS == 0: generic fallback with the y_i in dynamic LDS ([limb][thread]).
template <int NS, bool EXACT>
RH_DEV void bext_body(const u64* in, int in_rows, int src_limb0, int nsrc, const BextSource* __restrict__ S,
                      int ntgt_c, int ntgt, const BextTarget* __restrict__ T, const u64* __restrict__ coef, const u64* __restrict__ vt,
                      u64* out0, int out0_rows, u64* out1, int out1_rows, const u64* other, int other_rows, int N, int add_mode, int post) {
  // targets [0, ntgt_c) are extended (all with the plan-uniform post step `post`); [ntgt_c, ntgt) are the digit limbs of
  // DecomposeAndSplit: no extension, the centred subtraction applies to what the output already holds
  // dynamic LDS: [ntgt_c * (nsrc + 1)] the vt table (-v*Q mod p for every target and v), then [nsrc][256] the y_i (NS == 0 only).
  // The per-target vt lookup must not be a global load: its s_waitcnt vmcnt would also wait for the STORES of the previous
  // targets (loads and stores share that counter on gfx9) and serialise the loop on write latency.
  extern __shared__ u64 dyn_lds[];
  const int tid = threadIdx.x;
  const int vt_words = ntgt_c * (nsrc + 1);
  u64* const vt_lds = dyn_lds;
  u64* const ylds = dyn_lds + vt_words;
  for (int i = tid; i < vt_words; i += 256) vt_lds[i] = vt[i];
  __syncthreads();
  const int k = blockIdx.x * 256 + tid;
  const int poly = blockIdx.y;
  const bool live = k < N;
  constexpr int NY = NS > 0 ? NS : 1;
  u64 yr[NY];
  double vi = 0.0;
  auto source = [&](int i) {
    const BextSource s = S[i];
    u64 x = live ? in[((size_t)poly * in_rows + src_limb0 + i) * N + k] : 0;
    if (add_mode == BEXT_ADD_CRED) x = cred(x + s.half, s.q);            // AddScalarBigint -> addscalarvec
    else if (add_mode == BEXT_ADD_RAW) x = x + s.half;                    // reconstructRNSCentered :522
    const u64 y = mred(x, s.qstar_inv, s.q, s.qinv);
    vi += (double)y / (double)s.q;                                        // :576-593, one rounding per op
    return y;
  };
  if constexpr (NS > 0) {
#pragma unroll
    for (int i = 0; i < NS; ++i) { yr[i] = 0; if (EXACT || i < nsrc) yr[i] = source(i); }
  } else {
    for (int i = 0; i < nsrc; ++i) ylds[i * 256 + tid] = source(i);
  }
  const u64 v = (u64)vi;
  auto one_target = [&](int j) {                    // branch-free body
    const BextTarget t = T[j];
    u64* outp = t.buf ? out1 : out0;
    const int rows = t.buf ? out1_rows : out0_rows;
    const size_t o = ((size_t)poly * rows + t.limb) * N + k;
    const u64* cj = coef + (size_t)j * nsrc;
    u64 rlo, rhi;
    if constexpr (NS > 0) {
      // multSum :612-649, the same 128-bit sum by columns: y = y1*2^32 + y0, c = c1*2^32 + c0 with y, c < 2^61, so
      // y1, c1 < 2^29.  L = sum y0*c0 (carries counted in cL), M1 = sum y0*c1, M2 = sum y1*c0 (each term < 2^61: no
      // overflow up to 8 terms, carries counted in cM beyond), H = sum y1*c1 (< 2^63 for 32 terms).  One multiply-add
      // per partial product instead of a 128-bit add with compare-and-select carries per term.
      u64 Lc = 0, M1 = 0, M2 = 0, H = 0;
      u32 cL = 0, cM = 0;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        if (EXACT || i < nsrc) {
          const u64 cw = cj[i];
          const u32 c0 = (u32)cw, c1 = (u32)(cw >> 32);
          const u32 y0 = (u32)yr[i], y1 = (u32)(yr[i] >> 32);
          mac_carry(Lc, cL, y0, c0);
          if constexpr (NS <= 8) {
            M1 += (u64)y0 * c1;
            M2 += (u64)y1 * c0;
          } else {
            mac_carry(M1, cM, y0, c1);
            mac_carry(M2, cM, y1, c0);
          }
          H += (u64)y1 * c1;
        }
      }
      const u64 mid = M1 + M2;
      const u64 cm = (u64)(mid < M1) + cM;                                // weight 2^96
      rlo = Lc + (mid << 32);
      rhi = H + cL + (mid >> 32) + (u64)(rlo < Lc) + (cm << 32);
    } else {
      u128 acc = (u128)ylds[tid] * cj[0];
      rlo = (u64)acc; rhi = (u64)(acc >> 64);
      for (int i = 1; i < nsrc; ++i) {
        const u128 m = (u128)ylds[i * 256 + tid] * cj[i];
        const u64 mlo = (u64)m, mhi = (u64)(m >> 64);
        const u64 s = rlo + mlo;
        rhi += mhi + (u64)(s < rlo);
        rlo = s;
      }
    }
    const u64 hhi = mulhi64(rlo * t.pinv, t.p);
    u64 r = rhi - hhi + t.p + vt_lds[j * (nsrc + 1) + (int)v];            // :651-672
    if (post >= 1) r = cred(r + t.p - t.half, t.p);                       // SubScalarBigint -> subscalarvec
    if (post == 2) {
      const u64 y = live ? other[((size_t)poly * other_rows + t.limb) * N + k] : 0;
      r = mred(2 * t.p - y + r, t.md_scalar, t.p, t.pinv);                // SubThenMulScalarMontgomeryTwoModulus
    }
    if (live) outp[o] = r;
  };
  int j = 0;
  for (; j + 1 < ntgt_c; j += 2) { one_target(j); one_target(j + 1); }   // two independent targets per trip: their constant loads,
  if (j < ntgt_c) one_target(j);                                         // vt gathers and multiply-add chains interleave
  if (post >= 1 && live) {
    for (int j = ntgt_c; j < ntgt; ++j) {
      const BextTarget t = T[j];
      u64* outp = t.buf ? out1 : out0;
      const int rows = t.buf ? out1_rows : out0_rows;
      const size_t o = ((size_t)poly * rows + t.limb) * N + k;
      outp[o] = cred(outp[o] + t.p - t.half, t.p);
    }
  }
}

template <int NS, bool EXACT>
__global__ void __launch_bounds__(256)
bext_kernel(const u64* in, int in_rows, int src_limb0, int nsrc, const BextSource* __restrict__ S,
            int ntgt_c, int ntgt, const BextTarget* __restrict__ T, const u64* __restrict__ coef, const u64* __restrict__ vt,
            u64* out0, int out0_rows, u64* out1, int out1_rows, const u64* other, int other_rows, int N, int add_mode, int post) {
  bext_body<NS, EXACT>(in, in_rows, src_limb0, nsrc, S, ntgt_c, ntgt, T, coef, vt, out0, out0_rows, out1, out1_rows, other, other_rows, N, add_mode, post);
}
// Several plans over the same input in ONE launch (blockIdx.z = the plan): every digit of a hybrid key-switch decomposition at once, for batches
// too small to fill the chip digit by digit (rh_bext_decompose_and_split_all).  Output blocks of plan z: out0 + z * stride0, out1 + z * stride1.
struct BextMulti {
  const BextSource* S[8]; const BextTarget* T[8]; const u64* coef[8]; const u64* vt[8];
  int src_limb0[8], nsrc[8], ntgt_c[8], ntgt[8];
};
template <int NS, bool EXACT>                      // EXACT: every plan has exactly NS source limbs (the usual case: all digits full)
__global__ void __launch_bounds__(256)
bext_multi_kernel(const u64* in, int in_rows, BextMulti m, u64* out0, size_t stride0, int out0_rows, u64* out1, size_t stride1, int out1_rows,
                  int N, int add_mode, int post) {
  const int z = blockIdx.z;
  bext_body<NS, EXACT>(in, in_rows, m.src_limb0[z], m.nsrc[z], m.S[z], m.ntgt_c[z], m.ntgt[z], m.T[z], m.coef[z], m.vt[z],
                      out0 + (size_t)z * stride0, out0_rows, out1 + (size_t)z * stride1, out1_rows, nullptr, 0, N, add_mode, post);
}

#define RH_BEXT_ARGS in, in_rows, src_limb0, p.nsrc, p.d_S, p.ntgt_c, p.ntgt, p.d_T, p.d_coef, p.d_vt, out0, out0_rows, out1, out1_rows, other, other_rows, N, add_mode, p.post
static void bext_dispatch(dim3 grid, hipStream_t st, const BextPlan& p, const u64* in, int in_rows, int src_limb0, u64* out0, int out0_rows,
                          u64* out1, int out1_rows, const u64* other, int other_rows, int N, int add_mode) {
  const size_t vt_bytes = (size_t)p.ntgt_c * (p.nsrc + 1) * 8;      // at most 64 targets x 33 entries = 16.5 KiB
  switch (p.nsrc) {
    case 1: bext_kernel<1, true><<<grid, 256, vt_bytes, st>>>(RH_BEXT_ARGS); return;
    case 2: bext_kernel<2, true><<<grid, 256, vt_bytes, st>>>(RH_BEXT_ARGS); return;
    case 3: bext_kernel<3, true><<<grid, 256, vt_bytes, st>>>(RH_BEXT_ARGS); return;
    case 4: bext_kernel<4, true><<<grid, 256, vt_bytes, st>>>(RH_BEXT_ARGS); return;
    case 5: bext_kernel<5, true><<<grid, 256, vt_bytes, st>>>(RH_BEXT_ARGS); return;
    case 6: bext_kernel<6, true><<<grid, 256, vt_bytes, st>>>(RH_BEXT_ARGS); return;
    case 7: bext_kernel<7, true><<<grid, 256, vt_bytes, st>>>(RH_BEXT_ARGS); return;
    case 8: bext_kernel<8, true><<<grid, 256, vt_bytes, st>>>(RH_BEXT_ARGS); return;
    default: break;
  }
  if (p.nsrc <= 16) bext_kernel<16, false><<<grid, 256, vt_bytes, st>>>(RH_BEXT_ARGS);
  else if (p.nsrc <= 32) bext_kernel<32, false><<<grid, 256, vt_bytes, st>>>(RH_BEXT_ARGS);
  else bext_kernel<0, false><<<grid, 256, vt_bytes + (size_t)p.nsrc * 256 * 8, st>>>(RH_BEXT_ARGS);
}

// DecomposeAndSplit, single-prime digit (decompLvl < 0): sign-aware copy/reduce into every limb (:402-436)
__global__ void __launch_bounds__(256)
bext_sign_copy_kernel(const u64* in, int in_rows, int src_limb, u64 qd, int ntgt, const SignTarget* __restrict__ T,
                      u64* out0, int out0_rows, u64* out1, int out1_rows, int N) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int poly = blockIdx.y;
  if (k >= N) return;
  u64 coeff = in[((size_t)poly * in_rows + src_limb) * N + k];
  u64 pos = 1, neg = 0;
  if (coeff >= (qd >> 1)) { coeff = qd - coeff; pos = 0; neg = 1; }
  for (int j = 0; j < ntgt; ++j) {
    const SignTarget t = T[j];
    const u64 tmp = bred_add(coeff, t.p, t.bred0);
    u64* outp = t.buf ? out1 : out0;
    const int rows = t.buf ? out1_rows : out0_rows;
    outp[((size_t)poly * rows + t.limb) * N + k] = tmp * pos + (t.p - tmp) * neg;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct rh_bext {
  rh_ring* Q = nullptr; rh_ring* P = nullptr;
  std::map<std::array<int, 5>, BextPlan> plans;
  u64* buf[9] = {}; size_t buf_words[9] = {};      // 0,1: ModDownNTT buffers; 2..8: gadget product (keyswitch.hip)
  std::recursive_mutex mu;
};

template <class T>
static int upv(T** d, const std::vector<T>& h) {
  if (hipMalloc((void**)d, (h.size() ? h.size() : 1) * sizeof(T)) != hipSuccess) return rh_fail(RH_ERR_NOMEM, "hipMalloc failed");
  if (!h.empty() && hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return rh_fail(RH_ERR_DEVICE, "hipMemcpy failed");
  return 0;
}

// floor(prod(M)/2) mod m, exact, by carrying the product as (value mod 2m): floor(X/2) mod m = ((X mod 2m) - (X & 1)) / 2
static u64 half_product_mod(const std::vector<u64>& M, u64 m) {
  const rh::u128 mm = (rh::u128)2 * m;             // < 2^62
  rh::u128 acc = 1;
  for (u64 f : M) acc = (acc * ((rh::u128)f % mm)) % mm;      // both factors < 2^62
  return (u64)((acc - 1) / 2);                     // every modulus is odd, so the product (and acc) is odd
}

// GenModUpConstants (:93-164) for source basis Qs and a list of target moduli
static void gen_modup(const std::vector<u64>& Qs, const std::vector<u64>& tg, std::vector<u64>& qstar_inv_mont,
                      std::vector<u64>& coef, std::vector<u64>& vt) {
  const int n = (int)Qs.size(), m = (int)tg.size();
  qstar_inv_mont.resize(n); coef.assign((size_t)m * n, 0); vt.assign((size_t)m * (n + 1), 0);
  for (int i = 0; i < n; ++i) {
    const u64 qi = Qs[i];
    u64 star = 1 % qi;
    for (int j = 0; j < n; ++j) if (j != i) star = rh::mulmod(star, Qs[j] % qi, qi);
    qstar_inv_mont[i] = rh::mform(rh::invmod_prime(star, qi), qi);
    for (int j = 0; j < m; ++j) {
      const u64 p = tg[j];
      u64 s = 1 % p;
      for (int u = 0; u < n; ++u) if (u != i) s = rh::mulmod(s, Qs[u] % p, p);
      coef[(size_t)j * n + i] = rh::mform(s, p);
    }
  }
  for (int j = 0; j < m; ++j) {
    const u64 p = tg[j];
    u64 QmodP = 1 % p;
    for (int i = 0; i < n; ++i) QmodP = rh::mulmod(QmodP, Qs[i] % p, p);
    const u64 v = p - QmodP;
    u64* row = &vt[(size_t)j * (n + 1)];
    row[0] = 0;
    for (int i = 1; i <= n; ++i) { u64 t = row[i - 1] + v; row[i] = t >= p ? t - p : t; }
  }
}
// genmodDownConstants (:25-49): prod_{j<=levelP} p_j^-1 mod q_i, Montgomery form
static u64 moddown_const(const std::vector<u64>& Ps, u64 qi) {
  u64 acc = 1 % qi;
  for (u64 p : Ps) acc = rh::mulmod(acc, rh::invmod_prime(p % qi, qi), qi);
  return rh::mform(acc, qi);
}

extern "C" int rh_bext_create(rh_bext** out, rh_ring* ringQ, rh_ring* ringP) {
  if (!out || !ringQ) return rh_fail(RH_ERR_ARG, "rh_bext_create: null argument");
  if (ringP && (ringP->N != ringQ->N || ringP->device != ringQ->device)) return rh_fail(RH_ERR_ARG, "rh_bext_create: rings differ in N or device");
  rh_bext* be = new rh_bext();
  be->Q = ringQ; be->P = ringP;
  *out = be;
  return RH_OK;
}
extern "C" void rh_bext_destroy(rh_bext* be) {
  if (!be) return;
  for (auto& kv : be->plans) {
    BextPlan& p = kv.second;
    void* ptrs[] = {p.d_S, p.d_T, p.d_coef, p.d_vt, p.d_sign};
    for (void* q : ptrs) if (q) (void)hipFree(q);
  }
  for (int i = 0; i < 9; ++i) if (be->buf[i]) (void)hipFree(be->buf[i]);
  delete be;
}

static int ensure_buf(rh_bext* be, int which, size_t words) {
  if (be->buf_words[which] >= words) return 0;
  if (be->buf[which]) (void)hipFree(be->buf[which]);
  be->buf[which] = nullptr; be->buf_words[which] = 0;
  if (hipMalloc((void**)&be->buf[which], words * 8) != hipSuccess) return rh_fail(RH_ERR_NOMEM, "hipMalloc(basis-extension scratch) failed");
  be->buf_words[which] = words;
  return 0;
}

std::recursive_mutex& rh_bext_mutex(rh_bext* be) { return be->mu; }
// Pre-sizes the extender's scratch for key switches / ModDowns of up to npoly polys at the rings' top levels
extern "C" int rh_bext_reserve(rh_bext* be, int npoly) {
  if (!be || npoly < 0) return rh_fail(RH_ERR_ARG, "rh_bext_reserve: bad argument");
  RhBextGuard guard(be);
  (void)hipSetDevice(be->Q->device);
  const size_t N = be->Q->N, wq = (size_t)npoly * be->Q->L * N, wp = be->P ? (size_t)npoly * be->P->L * N : 0;
  const size_t beta = be->P ? (size_t)(be->Q->L + be->P->L - 1) / be->P->L : 1;
  const size_t need[9] = {2 * wq, 2 * wp, wq, beta * wq, beta * wp, 2 * wp, 0, wq, wq};
  for (int i = 0; i < 9; ++i) if (need[i]) if (int rc = ensure_buf(be, i, need[i])) return rc;
  return RH_OK;
}
rh_ring* rh_bext_ringQ(rh_bext* be) { return be->Q; }
rh_ring* rh_bext_ringP(rh_bext* be) { return be->P; }
int rh_bext_scratch(rh_bext* be, int which, size_t words, u64** out) {
  if (int rc = ensure_buf(be, which, words)) return rc;
  *out = be->buf[which];
  return 0;
}

// plan kinds: 0 = ModUp src->tgt centred (post 1); 1 = same + fused ModDown (post 2); key = {kind, dir, lvlSrc, lvlTgt, 0}
//   dir 0: Q -> P, dir 1: P -> Q
static int get_modup_plan(rh_bext* be, int kind, int dir, int lvlS, int lvlT, BextPlan** out) {
  std::array<int, 5> key{kind, dir, lvlS, lvlT, 0};
  auto it = be->plans.find(key);
  if (it != be->plans.end()) { *out = &it->second; return 0; }
  rh_ring* RS = dir == 0 ? be->Q : be->P;
  rh_ring* RT = dir == 0 ? be->P : be->Q;
  std::vector<u64> Qs(RS->moduli.begin(), RS->moduli.begin() + lvlS + 1), tg(RT->moduli.begin(), RT->moduli.begin() + lvlT + 1);
  std::vector<u64> qsi, coef, vt;
  gen_modup(Qs, tg, qsi, coef, vt);
  std::vector<BextSource> S(Qs.size()); std::vector<BextTarget> T(tg.size());
  for (size_t i = 0; i < Qs.size(); ++i) S[i] = BextSource{Qs[i], RS->mred[i], qsi[i], half_product_mod(Qs, Qs[i])};
  for (size_t j = 0; j < tg.size(); ++j) {
    BextTarget t{};
    t.p = tg[j]; t.pinv = RT->mred[j]; t.half = half_product_mod(Qs, tg[j]);
    t.md_scalar = kind == 1 ? tg[j] - moddown_const(Qs, tg[j]) : 0;      // s.Modulus - modDownConstants[i]
    t.buf = 0; t.limb = (int)j; t.post = kind == 1 ? 2 : 1; t.skip = 0;
    T[j] = t;
  }
  BextPlan p;
  if (int rc = rh_bext_upload_plan(p, S, T, coef, vt)) return rc;
  auto ins = be->plans.emplace(key, p);
  *out = &ins.first->second;
  return 0;
}

static int launch_plan(rh_bext* be, const BextPlan& p, const u64* in, int in_rows, int src_limb0, u64* out0, int out0_rows,
                       u64* out1, int out1_rows, const u64* other, int other_rows, int npoly, int add_mode) {
  rh_ring* R = be->Q;
  const int N = R->N;
  if (npoly <= 0) return RH_OK;
  dim3 grid((N + 255) / 256, npoly);
  (void)hipGetLastError();
  bext_dispatch(grid, rh_stream(R), p, in, in_rows, src_limb0, out0, out0_rows, out1, out1_rows, other, other_rows, N, add_mode);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rh_fail(RH_ERR_DEVICE, "bext_kernel launch failed: %s", hipGetErrorString(e));
  return RH_OK;
}

static int check_levels(rh_bext* be, int levelQ, int levelP, bool needP) {
  if (!be) return rh_fail(RH_ERR_ARG, "null basis extender");
  if (levelQ < 0 || levelQ >= be->Q->L) return rh_fail(RH_ERR_ARG, "levelQ %d out of range [0,%d)", levelQ, be->Q->L);
  if (needP) {
    if (!be->P) return rh_fail(RH_ERR_ARG, "basis extender has no P ring");
    if (levelP < 0 || levelP >= be->P->L) return rh_fail(RH_ERR_ARG, "levelP %d out of range [0,%d)", levelP, be->P->L);
  }
  if (levelQ + 1 > 32 || (needP && levelP + 1 > 32)) return rh_fail(RH_ERR_ARG, "basis extension supports at most 32 source limbs (ring/basis_extension.go:285)");
  (void)hipSetDevice(be->Q->device);
  return 0;
}

extern "C" int rh_bext_modup_q_to_p(rh_bext* be, int levelQ, int levelP, const uint64_t* polQ, uint64_t* polP, int npoly) {
  if (!be) return rh_fail(RH_ERR_ARG, "null basis extender");
  RhBextGuard guard(be);
  if (int rc = check_levels(be, levelQ, levelP, true)) return rc;
  BextPlan* p; if (int rc = get_modup_plan(be, 0, 0, levelQ, levelP, &p)) return rc;
  return launch_plan(be, *p, polQ, levelQ + 1, 0, polP, levelP + 1, nullptr, 0, nullptr, 0, npoly, BEXT_ADD_CRED);
}
extern "C" int rh_bext_modup_p_to_q(rh_bext* be, int levelP, int levelQ, const uint64_t* polP, uint64_t* polQ, int npoly) {
  if (!be) return rh_fail(RH_ERR_ARG, "null basis extender");
  RhBextGuard guard(be);
  if (int rc = check_levels(be, levelQ, levelP, true)) return rc;
  BextPlan* p; if (int rc = get_modup_plan(be, 0, 1, levelP, levelQ, &p)) return rc;
  return launch_plan(be, *p, polP, levelP + 1, 0, polQ, levelQ + 1, nullptr, 0, nullptr, 0, npoly, BEXT_ADD_CRED);
}
extern "C" int rh_bext_moddown_qp_to_q(rh_bext* be, int levelQ, int levelP, const uint64_t* p1Q, const uint64_t* p1P, uint64_t* p2Q, int npoly) {
  if (!be) return rh_fail(RH_ERR_ARG, "null basis extender");
  RhBextGuard guard(be);
  if (int rc = check_levels(be, levelQ, levelP, true)) return rc;
  BextPlan* p; if (int rc = get_modup_plan(be, 1, 1, levelP, levelQ, &p)) return rc;
  return launch_plan(be, *p, p1P, levelP + 1, 0, p2Q, levelQ + 1, nullptr, 0, p1Q, levelQ + 1, npoly, BEXT_ADD_CRED);
}
extern "C" int rh_bext_moddown_qp_to_p(rh_bext* be, int levelQ, int levelP, const uint64_t* p1Q, const uint64_t* p1P, uint64_t* p2P, int npoly) {
  if (!be) return rh_fail(RH_ERR_ARG, "null basis extender");
  RhBextGuard guard(be);
  if (int rc = check_levels(be, levelQ, levelP, true)) return rc;
  BextPlan* p; if (int rc = get_modup_plan(be, 1, 0, levelQ, levelP, &p)) return rc;
  return launch_plan(be, *p, p1Q, levelQ + 1, 0, p2P, levelP + 1, nullptr, 0, p1P, levelP + 1, npoly, BEXT_ADD_CRED);
}

// ModDownQPtoQNTT (:241-258): INTTLazy on P, ModUpPtoQ, NTTLazy on Q, fused subtract-multiply
extern "C" int rh_bext_moddown_qp_to_q_ntt(rh_bext* be, int levelQ, int levelP, const uint64_t* p1Q, const uint64_t* p1P, uint64_t* p2Q, int npoly) {
  return rh_bext_moddown_ntt_add(be, levelQ, levelP, p1Q, p1P, p2Q, npoly, nullptr);
}
int rh_bext_moddown_ntt_add(rh_bext* be, int levelQ, int levelP, const u64* p1Q, const u64* p1P, u64* p2Q, int npoly, const u64* addend) {
  if (!be) return rh_fail(RH_ERR_ARG, "null basis extender");
  RhBextGuard guard(be);
  if (int rc = check_levels(be, levelQ, levelP, true)) return rc;
  if (be->Q->kind != be->P->kind) return rh_fail(RH_ERR_ARG, "ModDownQPtoQNTT: ringQ and ringP differ in ring type");
  const size_t N = be->Q->N;
  if (int rc = ensure_buf(be, 0, (size_t)npoly * (levelQ + 1) * N)) return rc;
  if (int rc = ensure_buf(be, 1, (size_t)npoly * (levelP + 1) * N)) return rc;
  u64* buffQ = be->buf[0]; u64* buffP = be->buf[1];
  if (int rc = rh_ring_ntt_any(be->P, p1P, buffP, npoly, levelP + 1, 0, true)) return rc;          // ringP.INTTLazy (canonical for N >= 16; any ring type)
  BextPlan* p; if (int rc = get_modup_plan(be, 0, 1, levelP, levelQ, &p)) return rc;
  if (int rc = launch_plan(be, *p, buffP, levelP + 1, 0, buffQ, levelQ + 1, nullptr, 0, nullptr, 0, npoly, BEXT_ADD_CRED)) return rc;
  // ringQ.NTTLazy(buffQ, buffQ): the following MRed yields the canonical residue for any representative (< 8q) of the
  // NTT values, so the canonical forward transform is used.
  std::vector<u64> sc(levelQ + 1);
  std::vector<u64> Ps(be->P->moduli.begin(), be->P->moduli.begin() + levelP + 1);
  for (int i = 0; i <= levelQ; ++i) sc[i] = be->Q->moduli[i] - moddown_const(Ps, be->Q->moduli[i]);
  if (rh_can_fuse_submul(be->Q))                 // the subtract-multiply rides in the forward tile kernel's epilogue
    return rh_std_ntt_submul_launch(be->Q, buffQ, npoly, levelQ + 1, 0, p1Q, levelQ + 1, p2Q, levelQ + 1, sc.data(), false, addend, levelQ + 1);
  if (int rc = rh_ring_ntt_any(be->Q, buffQ, buffQ, npoly, levelQ + 1, 0, false)) return rc;
  if (!addend) return rh_vec_launch(be->Q, RH_OP_SUB_THEN_MUL_SCALAR_MONT_TWO_MODULUS, buffQ, p1Q, p2Q, npoly, levelQ + 1, 0, sc.data(), nullptr);
  // the output may be the addend's own buffer: finish in the scratch, add last
  if (int rc = rh_vec_launch(be->Q, RH_OP_SUB_THEN_MUL_SCALAR_MONT_TWO_MODULUS, buffQ, p1Q, buffQ, npoly, levelQ + 1, 0, sc.data(), nullptr)) return rc;
  return rh_vec_launch(be->Q, RH_OP_ADD, buffQ, addend, p2Q, npoly, levelQ + 1, 0, nullptr, nullptr);
}

// The two ModDowns that end a key switch (one per ciphertext component) as one batch of 2*npoly polys up to the tile stages:
// p1P holds both P parts back to back ([2][npoly][levelP+1][N]); Q parts, outputs and addends are separate blocks.
int rh_bext_moddown_ntt_pair(rh_bext* be, int levelQ, int levelP, const u64* q0, const u64* q1, const u64* p1P, u64* out0, u64* out1,
                             int npoly, const u64* add0, const u64* add1) {
  if (!be) return rh_fail(RH_ERR_ARG, "null basis extender");
  RhBextGuard guard(be);
  if (int rc = check_levels(be, levelQ, levelP, true)) return rc;
  const size_t N = be->Q->N, wq = (size_t)npoly * (levelQ + 1) * N, wp = (size_t)npoly * (levelP + 1) * N;
  if (!rh_can_fuse_submul(be->Q) || be->P->kind != RH_RING_STANDARD) {          // small rings: component by component
    if (int rc = rh_bext_moddown_ntt_add(be, levelQ, levelP, q0, p1P, out0, npoly, add0)) return rc;
    return rh_bext_moddown_ntt_add(be, levelQ, levelP, q1, p1P + wp, out1, npoly, add1);
  }
  if (int rc = ensure_buf(be, 0, 2 * wq)) return rc;
  if (int rc = ensure_buf(be, 1, 2 * wp)) return rc;
  u64* buffQ = be->buf[0]; u64* buffP = be->buf[1];
  if (int rc = rh_std_ntt_launch(be->P, p1P, buffP, 2 * npoly, levelP + 1, 0, true, true, 0)) return rc;       // ringP.INTTLazy, both components
  BextPlan* p; if (int rc = get_modup_plan(be, 0, 1, levelP, levelQ, &p)) return rc;
  if (int rc = launch_plan(be, *p, buffP, levelP + 1, 0, buffQ, levelQ + 1, nullptr, 0, nullptr, 0, 2 * npoly, BEXT_ADD_CRED)) return rc;
  if (be->Q->logN > 12) if (int rc = rh_std_ntt_launch(be->Q, buffQ, buffQ, 2 * npoly, levelQ + 1, 0, false, false, 1)) return rc;   // column stages
  std::vector<u64> sc(levelQ + 1);
  std::vector<u64> Ps(be->P->moduli.begin(), be->P->moduli.begin() + levelP + 1);
  for (int i = 0; i <= levelQ; ++i) sc[i] = be->Q->moduli[i] - moddown_const(Ps, be->Q->moduli[i]);
  if (be->Q->asm_tile && be->Q->pair_submul && (add0 == nullptr) == (add1 == nullptr))       // both components in one launch
    return rh_std_ntt_submul_launch_pair(be->Q, buffQ, npoly, levelQ + 1, q0, q1, levelQ + 1, out0, out1, levelQ + 1, sc.data(), add0, add1, levelQ + 1);
  if (int rc = rh_std_ntt_submul_launch(be->Q, buffQ, npoly, levelQ + 1, 0, q0, levelQ + 1, out0, levelQ + 1, sc.data(), true, add0, levelQ + 1)) return rc;
  return rh_std_ntt_submul_launch(be->Q, buffQ + wq, npoly, levelQ + 1, 0, q1, levelQ + 1, out1, levelQ + 1, sc.data(), true, add1, levelQ + 1);
}

// DecomposeAndSplit (:381-502)
extern "C" int rh_bext_decompose_and_split(rh_bext* be, int levelQ, int levelP, int nbPi, int digit, const uint64_t* p0Q,
                                           uint64_t* p1Q, uint64_t* p1P, int npoly) {
  if (!be) return rh_fail(RH_ERR_ARG, "null basis extender");
  RhBextGuard guard(be);
  if (int rc = check_levels(be, levelQ, levelP, be && be->P != nullptr)) return rc;
  if (nbPi < 1 || digit < 0) return rh_fail(RH_ERR_ARG, "DecomposeAndSplit: bad nbPi/digit");
  rh_ring* RQ = be->Q; rh_ring* RP = be->P;
  const int N = RQ->N;
  const int st = digit * nbPi;
  if (st > levelQ) return rh_fail(RH_ERR_ARG, "DecomposeAndSplit: digit %d starts past levelQ %d", digit, levelQ);
  int decompLvl = (levelQ > nbPi * (digit + 1) - 1) ? nbPi - 2 : (levelQ % nbPi) - 1;            // :394-399
  const int nP = RP ? levelP + 1 : 0;
  std::array<int, 5> key{2, levelQ, RP ? levelP : -1, nbPi, digit};
  auto it = be->plans.find(key);
  if (it == be->plans.end()) {
    BextPlan p;
    if (decompLvl < 0) {
      std::vector<SignTarget> T;
      for (int i = 0; i <= levelQ; ++i) T.push_back(SignTarget{RQ->moduli[i], RQ->bred[2 * i], 0, i});
      for (int i = 0; i < nP; ++i) T.push_back(SignTarget{RP->moduli[i], RP->bred[2 * i], 1, i});
      p.ntgt = (int)T.size(); p.qd = RQ->moduli[st];
      if (int rc = upv(&p.d_sign, T)) return rc;
    } else {
      if (!RP) return rh_fail(RH_ERR_ARG, "DecomposeAndSplit: multi-prime digit needs ringP");
      if (nbPi > RP->L) return rh_fail(RH_ERR_ARG, "DecomposeAndSplit: nbPi %d exceeds the P ring (%d limbs)", nbPi, RP->L);
      int ed = st + nbPi; if (ed > levelQ + 1) ed = levelQ + 1;
      std::vector<u64> Qs(RQ->moduli.begin() + st, RQ->moduli.begin() + ed);
      // constants: GenModUpConstants(Q[st:ed], Q_all ++ P[:nbPi]) (NewDecomposer :345-372); only the rows of the
      // limbs actually written are kept.  Row u of the reference = limb u of Q (u < len(Q)) or len(Q)+j for P_j.
      std::vector<u64> tg; std::vector<BextTarget> T;
      for (int pass = 0; pass < 2; ++pass) {                              // extended limbs first, the digit's own limbs last
        for (int j = 0; j <= levelQ; ++j) {
          const int skip = (j >= st && j < ed) ? 1 : 0;
          if (skip != pass) continue;
          BextTarget t{}; t.p = RQ->moduli[j]; t.pinv = RQ->mred[j]; t.half = half_product_mod(Qs, t.p);
          t.buf = 0; t.limb = j; t.post = 1; t.skip = skip;
          T.push_back(t); tg.push_back(t.p);
        }
        if (pass == 0) for (int j = 0; j < nP; ++j) {
          BextTarget t{}; t.p = RP->moduli[j]; t.pinv = RP->mred[j]; t.half = half_product_mod(Qs, t.p);
          t.buf = 1; t.limb = j; t.post = 1; t.skip = 0;
          T.push_back(t); tg.push_back(t.p);
        }
      }
      std::vector<u64> qsi, coef, vt;
      gen_modup(Qs, tg, qsi, coef, vt);
      std::vector<BextSource> S(Qs.size());
      for (size_t i = 0; i < Qs.size(); ++i) S[i] = BextSource{Qs[i], RQ->mred[st + i], qsi[i], half_product_mod(Qs, Qs[i])};
      if (int rc = rh_bext_upload_plan(p, S, T, coef, vt)) return rc;
    }
    it = be->plans.emplace(key, p).first;
  }
  const BextPlan& p = it->second;
  if (npoly <= 0) return RH_OK;
  if (decompLvl < 0) {
    dim3 grid((N + 255) / 256, npoly);
    (void)hipGetLastError();
    bext_sign_copy_kernel<<<grid, 256, 0, rh_stream(RQ)>>>(p0Q, levelQ + 1, st, p.qd, p.ntgt, p.d_sign, p1Q, levelQ + 1, p1P, nP, N);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rh_fail(RH_ERR_DEVICE, "bext_sign_copy_kernel launch failed: %s", hipGetErrorString(e));
    return RH_OK;
  }
  return launch_plan(be, p, p0Q, levelQ + 1, st, p1Q, levelQ + 1, p1P, nP, nullptr, 0, npoly, BEXT_ADD_RAW);
}

// DecomposeAndSplit for digits 0 .. beta-1 in ONE launch (small batches; keyswitch.hip).  Returns 1 (and launches nothing) when a digit is not a
// register-resident multi-prime plan (single-prime digits, more than 8 source limbs, more than 8 digits): the caller goes digit by digit.
int rh_bext_decompose_and_split_all(rh_bext* be, int levelQ, int levelP, int nbPi, int beta, const u64* p0Q, u64* p1Q, size_t strideQ,
                                    u64* p1P, size_t strideP, int npoly) {
  if (beta < 1 || beta > 8 || !be->P) return 1;
  RhBextGuard guard(be);
  BextMulti m; memset(&m, 0, sizeof m);
  int post = -1; size_t vt_bytes = 0;
  for (int d = 0; d < beta; ++d) {
    const int decompLvl = (levelQ > nbPi * (d + 1) - 1) ? nbPi - 2 : (levelQ % nbPi) - 1;
    if (decompLvl < 0) return 1;
    if (int rc = rh_bext_decompose_and_split(be, levelQ, levelP, nbPi, d, p0Q, p1Q, p1P, 0)) return rc;      // builds the plan, launches nothing
    auto it = be->plans.find(std::array<int, 5>{2, levelQ, levelP, nbPi, d});
    if (it == be->plans.end()) return 1;
    const BextPlan& p = it->second;
    if (p.nsrc > 8 || p.ntgt == 0 || (post >= 0 && p.post != post)) return 1;
    post = p.post;
    m.S[d] = p.d_S; m.T[d] = p.d_T; m.coef[d] = p.d_coef; m.vt[d] = p.d_vt;
    m.src_limb0[d] = d * nbPi; m.nsrc[d] = p.nsrc; m.ntgt_c[d] = p.ntgt_c; m.ntgt[d] = p.ntgt;
    const size_t vb = (size_t)p.ntgt_c * (p.nsrc + 1) * 8;
    if (vb > vt_bytes) vt_bytes = vb;
  }
  if (npoly <= 0) return RH_OK;
  rh_ring* RQ = be->Q;
  (void)hipGetLastError();
  bool uniform = true;
  for (int d = 1; d < beta; ++d) uniform &= m.nsrc[d] == m.nsrc[0];
  const dim3 grid((RQ->N + 255) / 256, npoly, beta);
  hipStream_t st = rh_stream(RQ);
#define RH_BM(NS, EX) bext_multi_kernel<NS, EX><<<grid, 256, vt_bytes, st>>>(p0Q, levelQ + 1, m, p1Q, strideQ, levelQ + 1, p1P, strideP, levelP + 1, RQ->N, BEXT_ADD_RAW, post)
  switch (uniform ? m.nsrc[0] : 0) {
    case 2: RH_BM(2, true); break; case 3: RH_BM(3, true); break; case 4: RH_BM(4, true); break; case 5: RH_BM(5, true); break;
    case 6: RH_BM(6, true); break; case 7: RH_BM(7, true); break; case 8: RH_BM(8, true); break;
    default: RH_BM(8, false); break;
  }
#undef RH_BM
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rh_fail(RH_ERR_DEVICE, "bext_multi_kernel launch failed: %s", hipGetErrorString(e));
  return RH_OK;
}

// ---- internals shared with kshard.hip (bext_internal.hpp) ----
u64 rh_half_product_mod(const std::vector<u64>& M, u64 m) { return half_product_mod(M, m); }
void rh_gen_modup(const std::vector<u64>& Qs, const std::vector<u64>& tg, std::vector<u64>& qstar_inv_mont, std::vector<u64>& coef,
                  std::vector<u64>& vt) { gen_modup(Qs, tg, qstar_inv_mont, coef, vt); }
u64 rh_moddown_const(const std::vector<u64>& Ps, u64 qi) { return moddown_const(Ps, qi); }
int rh_bext_upload_plan(BextPlan& p, const std::vector<BextSource>& S, const std::vector<BextTarget>& T, const std::vector<u64>& coef,
                        const std::vector<u64>& vt) {
  p.nsrc = (int)S.size(); p.ntgt = (int)T.size();
  p.ntgt_c = 0; p.post = T.empty() ? 0 : T[0].post;
  for (const BextTarget& t : T) {                  // extended targets first, then the skipped (digit) limbs; one post step per plan
    if (!t.skip) { if (p.ntgt_c != (int)(&t - T.data())) return rh_fail(RH_ERR_ARG, "basis-extension plan: skipped limbs must come last"); ++p.ntgt_c; }
    if (t.post != p.post) return rh_fail(RH_ERR_ARG, "basis-extension plan: mixed post steps");
  }
  int rc = upv(&p.d_S, S); if (!rc) rc = upv(&p.d_T, T); if (!rc) rc = upv(&p.d_coef, coef); if (!rc) rc = upv(&p.d_vt, vt);
  return rc;
}
int rh_bext_upload_sign_plan(BextPlan& p, const std::vector<SignTarget>& T, u64 qd) {
  p.ntgt = (int)T.size(); p.qd = qd;
  return upv(&p.d_sign, T);
}
void rh_bext_free_plan(BextPlan& p) {
  void* ptrs[] = {p.d_S, p.d_T, p.d_coef, p.d_vt, p.d_sign};
  for (void* q : ptrs) if (q) (void)hipFree(q);
  p = BextPlan();
}
int rh_bext_launch_raw(hipStream_t st, int N, const BextPlan& p, const u64* in, int in_rows, int src_limb0, u64* out0, int out0_rows,
                       u64* out1, int out1_rows, const u64* other, int other_rows, int npoly, int add_mode) {
  if (npoly <= 0 || p.ntgt == 0) return RH_OK;
  dim3 grid((N + 255) / 256, npoly);
  (void)hipGetLastError();
  bext_dispatch(grid, st, p, in, in_rows, src_limb0, out0, out0_rows, out1, out1_rows, other, other_rows, N, add_mode);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rh_fail(RH_ERR_DEVICE, "bext_kernel launch failed: %s", hipGetErrorString(e));
  return RH_OK;
}
int rh_bext_launch_sign(hipStream_t st, int N, const BextPlan& p, const u64* in, int in_rows, int src_limb, u64* out0, int out0_rows,
                        u64* out1, int out1_rows, int npoly) {
  if (npoly <= 0 || p.ntgt == 0) return RH_OK;
  dim3 grid((N + 255) / 256, npoly);
  (void)hipGetLastError();
  bext_sign_copy_kernel<<<grid, 256, 0, st>>>(in, in_rows, src_limb, p.qd, p.ntgt, p.d_sign, out0, out0_rows, out1, out1_rows, N);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rh_fail(RH_ERR_DEVICE, "bext_sign_copy_kernel launch failed: %s", hipGetErrorString(e));
  return RH_OK;
}
