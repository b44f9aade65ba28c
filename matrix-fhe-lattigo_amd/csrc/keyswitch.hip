// keyswitch.hip -- device-side orchestration of the hybrid key-switch gadget product.
//
// Replaces, for NTT-domain inputs and levelP >= 1, rlwe.Evaluator.GadgetProduct (core/rlwe/evaluator_gadget_product.go:16-30)
// = gadgetProductMultiplePLazy (:122-188) + ModDown NTT->NTT (:33-46), including DecomposeSingleNTT (:455-478).
// Everything it sequences already exists on the device (INTT, DecomposeAndSplit, NTT, ModDownQPtoQNTT); the one new
// kernel is the multiply-accumulate that streams the evaluation key once per digit for BOTH output components
// (MulCoeffsMontgomeryLazy / ...LazyThenAddLazy of ringqp, ring/ringqp/operations.go:115-155).
// Outputs are canonical (the reference ends with ring.Reduce and a full MRed in ModDown), so they are bit-identical.
#include <hip/hip_runtime.h>
#include <vector>
#include "engine_internal.hpp"
#include "bext_internal.hpp"

// acc_c[row] (=|+=) MRedLazy(evk_c[limb], c2[row]) for c = 0,1.  rows = npoly * L; evk is shared by every poly.
__global__ void __launch_bounds__(256)
gadget_mac_kernel(const u64* c2, const u64* __restrict__ evk0, const u64* __restrict__ evk1, u64* acc0, u64* acc1,
                  unsigned n, const LimbConsts* __restrict__ consts, int L, int first) {
  const u32 row = blockIdx.x, limb = row % (u32)L;
  const u64 q = consts[limb].q, qi = consts[limb].qinv;
  const size_t ro = (size_t)row * n, eo = (size_t)limb * n;
  const unsigned npairs = n >> 1;
  for (unsigned i = blockIdx.y * blockDim.x + threadIdx.x; i < npairs; i += gridDim.y * blockDim.x) {
    const ulonglong2 x = *reinterpret_cast<const ulonglong2*>(c2 + ro + 2 * (size_t)i);
    const ulonglong2 k0 = *reinterpret_cast<const ulonglong2*>(evk0 + eo + 2 * (size_t)i);
    const ulonglong2 k1 = *reinterpret_cast<const ulonglong2*>(evk1 + eo + 2 * (size_t)i);
    ulonglong2 a, b;
    a.x = mred_lazy(k0.x, x.x, q, qi); a.y = mred_lazy(k0.y, x.y, q, qi);
    b.x = mred_lazy(k1.x, x.x, q, qi); b.y = mred_lazy(k1.y, x.y, q, qi);
    if (!first) {
      const ulonglong2 pa = *reinterpret_cast<const ulonglong2*>(acc0 + ro + 2 * (size_t)i);
      const ulonglong2 pb = *reinterpret_cast<const ulonglong2*>(acc1 + ro + 2 * (size_t)i);
      a.x += pa.x; a.y += pa.y; b.x += pb.x; b.y += pb.y;
    }
    *reinterpret_cast<ulonglong2*>(acc0 + ro + 2 * (size_t)i) = a;
    *reinterpret_cast<ulonglong2*>(acc1 + ro + 2 * (size_t)i) = b;
  }
}

int rh_gadget_mac(rh_ring* r, const u64* c2, const u64* e0, const u64* e1, u64* a0, u64* a1, int npoly, int L, int first) {
  const unsigned rows = (unsigned)npoly * L, n = (unsigned)r->N;
  unsigned chunks = (n / 2 + 1023) / 1024; if (chunks < 1) chunks = 1; if (chunks > 64) chunks = 64;
  gadget_mac_kernel<<<dim3(rows, chunks), 256, 0, rh_stream(r)>>>(c2, e0, e1, a0, a1, n, r->d_consts, L, first);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rh_fail(RH_ERR_DEVICE, "gadget_mac_kernel launch failed: %s", hipGetErrorString(e));
  return RH_OK;
}

// All digits in one pass (used when the whole decomposition is resident: the hoisted layout [digit][poly][limb][N]):
// acc_c = sum_i MRedLazy(evk_c[i], c2[i]) with the reference's Reduce schedule (after every `overf` digits and at the end,
// core/rlwe/evaluator_gadget_product.go:166-187) kept in registers -- the accumulators are written once instead of being
// read and rewritten per digit.  poly is the fast block index, so the workgroups that share a key row run together.
struct LimbDigit { signed char d[RH_MAX_LIMBS]; };   // per limb: the digit whose own limbs include it (-1: none)
// BETA > 0: the digit count at compile time -- the 3 * BETA 16-byte loads of an element pair are issued before the first multiply.
// Not limited by the vector ALUs (a variant with ONE Montgomery reduction per output, 0.4 x the instructions, ran 8 % slower at
// 68 VGPRs) nor by key re-reads (PMC: fetch + write = the algorithmic 3.1 GB per launch): 4.8 TB/s over 4 + 2 streams.
template <int BETA, int PP>
__global__ void __launch_bounds__(256)
gadget_mac_all_kernel(const u64* c2, size_t digit_stride, const u64* __restrict__ evk, size_t evk_stride, int beta, int overf,
                      u64* acc0, u64* acc1, unsigned n, const LimbConsts* __restrict__ consts, int L, int npoly,
                      const u64* cx, LimbDigit own) {   // cx != null: limb l is read from cx (:467-468), not from c2, for digit own.d[l]
  // PP polys per workgroup (BETA > 0): the 2 * BETA key words of an element pair are loaded once and used for PP polys
  const u32 groups = ((u32)npoly + PP - 1) / PP;
  const u32 poly0 = (blockIdx.x % groups) * PP, limb = blockIdx.x / groups;
  const LimbConsts c = consts[limb];
  const size_t eo = (size_t)limb * n;
  const unsigned npairs = n >> 1;
  const unsigned per = (npairs + gridDim.y - 1) / gridDim.y;        // a contiguous run of the row per workgroup
  const unsigned lim = min(npairs, (blockIdx.y + 1) * per);
  for (unsigned i = blockIdx.y * per + threadIdx.x; i < lim; i += blockDim.x) {
    ulonglong2 a, b;
    int red;
    auto term = [&](const ulonglong2& x, const ulonglong2& k0, const ulonglong2& k1) {
      a.x += mred_lazy(k0.x, x.x, c.q, c.qinv); a.y += mred_lazy(k0.y, x.y, c.q, c.qinv);
      b.x += mred_lazy(k1.x, x.x, c.q, c.qinv); b.y += mred_lazy(k1.y, x.y, c.q, c.qinv);
      if (red % overf == overf - 1) {
        a.x = bred_add(a.x, c.q, c.bred0); a.y = bred_add(a.y, c.q, c.bred0);
        b.x = bred_add(b.x, c.q, c.bred0); b.y = bred_add(b.y, c.q, c.bred0);
      }
      ++red;
    };
    auto ld = [&](const u64* p) { return *reinterpret_cast<const ulonglong2*>(p + 2 * (size_t)i); };
    auto finish = [&](size_t ro) {
      if (red % overf != 0) {
        a.x = bred_add(a.x, c.q, c.bred0); a.y = bred_add(a.y, c.q, c.bred0);
        b.x = bred_add(b.x, c.q, c.bred0); b.y = bred_add(b.y, c.q, c.bred0);
      }
      *reinterpret_cast<ulonglong2*>(acc0 + ro + 2 * (size_t)i) = a;
      *reinterpret_cast<ulonglong2*>(acc1 + ro + 2 * (size_t)i) = b;
    };
    if constexpr (BETA > 0) {
      ulonglong2 k0[BETA], k1[BETA];
#pragma unroll
      for (int d = 0; d < BETA; ++d) {
        k0[d] = ld(evk + ((size_t)d * 2) * evk_stride + eo);
        k1[d] = ld(evk + ((size_t)d * 2 + 1) * evk_stride + eo);
      }
      for (u32 pp = 0; pp < (u32)PP && poly0 + pp < (u32)npoly; ++pp) {
        const size_t ro = ((size_t)(poly0 + pp) * L + limb) * n;
        ulonglong2 x[BETA];
#pragma unroll
        for (int d = 0; d < BETA; ++d) x[d] = ld((cx && own.d[limb] == d) ? cx + ro : c2 + (size_t)d * digit_stride + ro);
        a = {0, 0}; b = {0, 0}; red = 0;
#pragma unroll
        for (int d = 0; d < BETA; ++d) term(x[d], k0[d], k1[d]);
        finish(ro);
      }
    } else {
      const size_t ro = ((size_t)poly0 * L + limb) * n;               // PP == 1
      a = {0, 0}; b = {0, 0}; red = 0;
      for (int d = 0; d < beta; ++d)
        term(ld((cx && own.d[limb] == d) ? cx + ro : c2 + (size_t)d * digit_stride + ro),
             ld(evk + ((size_t)d * 2) * evk_stride + eo), ld(evk + ((size_t)d * 2 + 1) * evk_stride + eo));
      finish(ro);
    }
  }
}

// own_digit[l] (l < L): the digit whose own limbs include limb l (read from cx instead of c2), -1: none; NULL with cx: limb l belongs to
// digit l / digit_limbs (the unsharded chain)
int rh_gadget_mac_all(rh_ring* r, const u64* c2, size_t digit_stride, const u64* evk, int beta, int overf, u64* a0, u64* a1, int npoly, int L,
                      const u64* cx, const int* own_digit, int digit_limbs) {
  const unsigned n = (unsigned)r->N;
  unsigned chunks = (n / 2 + 1023) / 1024; if (chunks < 1) chunks = 1; if (chunks > 64) chunks = 64;
  LimbDigit own;
  for (int l = 0; l < RH_MAX_LIMBS; ++l) own.d[l] = (signed char)(l < L && cx ? (own_digit ? own_digit[l] : l / (digit_limbs > 0 ? digit_limbs : 1)) : -1);
  const int PPE = npoly >= 8 ? 4 : 1;             // polys per workgroup (same-box A/B at batch 64: 1 / 2 / 4 / 8 -> 9.04 / 8.93 / 8.89 / 8.92 ms per product)
#define RH_MAC_ALL(B, PP) gadget_mac_all_kernel<B, PP><<<dim3(((unsigned)npoly + PP - 1) / PP * L, chunks), 256, 0, rh_stream(r)>>>( \
    c2, digit_stride, evk, (size_t)r->L * n, beta, overf, a0, a1, n, r->d_consts, L, npoly, cx, own)
#define RH_MAC_PP(B) do { if (PPE == 4) RH_MAC_ALL(B, 4); else RH_MAC_ALL(B, 1); } while (0)
  switch (beta) {
    case 2: RH_MAC_PP(2); break;
    case 3: RH_MAC_PP(3); break;
    case 4: RH_MAC_PP(4); break;
    default: RH_MAC_ALL(0, 1);
  }
#undef RH_MAC_PP
#undef RH_MAC_ALL
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rh_fail(RH_ERR_DEVICE, "gadget_mac_all_kernel launch failed: %s", hipGetErrorString(e));
  return RH_OK;
}
static int mac_all(rh_ring* r, const u64* c2, size_t digit_stride, const u64* evk, int beta, int overf, u64* a0, u64* a1, int npoly, int L,
                   const u64* cx = nullptr, int digit_limbs = 1) {
  return rh_gadget_mac_all(r, c2, digit_stride, evk, beta, overf, a0, a1, npoly, L, cx, nullptr, digit_limbs);
}

int rh_overflow_margin(const std::vector<u64>& m, int level) {     // QiOverflowMargin / PiOverflowMargin, core/rlwe/params.go
  u64 mx = 0; for (int i = 0; i <= level; ++i) if (m[i] > mx) mx = m[i];
  return (int)(18446744073709551616.0 / (double)mx);
}

// DecomposeSingleNTT (:455-478): digit i of cx -> c2Q (levelQ+1 limbs), c2P (levelP+1 limbs), NTT domain.
// cx: NTT-domain input (its digit limbs are copied as they are, :467-468), cxInv: its inverse transform.
static int decompose_single_ntt(rh_bext* be, int levelQ, int levelP, int i, const u64* cx, const u64* cxInv, u64* c2Q, u64* c2P, int npoly,
                                bool copy_digit = true) {   // false: the consumer reads the digit's own limbs from cx itself
  rh_ring* RQ = rh_bext_ringQ(be); rh_ring* RP = rh_bext_ringP(be);
  const int LQ = levelQ + 1, LP = levelP + 1, N = RQ->N;
  if (int rc = rh_bext_decompose_and_split(be, levelQ, levelP, LP, i, cxInv, c2Q, c2P, npoly)) return rc;
  const int st = i * LP; int ed = st + LP; if (ed > LQ) ed = LQ;
  if (RQ->kind == RH_RING_STANDARD && RQ->logN >= 12) {          // the digit's own limbs are overwritten below: transform only the limbs around them
    if (int rc = rh_std_ntt_fwd_strided(RQ, c2Q, npoly, st, 0, LQ)) return rc;
    if (int rc = rh_std_ntt_fwd_strided(RQ, c2Q + (size_t)ed * N, npoly, LQ - ed, ed, LQ)) return rc;
  } else if (int rc = rh_ring_ntt_any(RQ, c2Q, c2Q, npoly, LQ, 0, false)) return rc;
  if (copy_digit && hipMemcpy2DAsync(c2Q + (size_t)st * N, (size_t)LQ * N * 8, cx + (size_t)st * N, (size_t)LQ * N * 8, (size_t)(ed - st) * N * 8, npoly,
                                     hipMemcpyDeviceToDevice, rh_stream(RQ)) != hipSuccess) return rh_fail(RH_ERR_DEVICE, "gadget_product: digit copy failed");
  return rh_ring_ntt_any(RP, c2P, c2P, npoly, LP, 0, false);
}

// small batch: rows of one block (polys x limbs) at or below the ring's tuning value ks_small_rows (0: never)
static bool ks_small(const rh_ring* RQ, int npoly, int LQ) { return RQ->ks_small_rows > 0 && (long)npoly * LQ <= RQ->ks_small_rows; }

// DecomposeSingleNTT for every digit: the basis extensions digit by digit, then ONE pipelined transform of all the Q blocks
// (rh_std_ntt_fwd_digits) and one of all the P blocks (contiguous: beta * npoly polys of LP limbs).
static int decompose_all_ntt(rh_bext* be, int levelQ, int levelP, int beta, const u64* cx, const u64* cxInv, u64* decQ, u64* decP, int npoly,
                             bool copy_digit) {
  rh_ring* RQ = rh_bext_ringQ(be); rh_ring* RP = rh_bext_ringP(be);
  const int LQ = levelQ + 1, LP = levelP + 1, N = RQ->N;
  const size_t wq = (size_t)npoly * LQ * N, wp = (size_t)npoly * LP * N;
  if (!rh_can_ntt_digits(RQ)) {
    for (int i = 0; i < beta; ++i)
      if (int rc = decompose_single_ntt(be, levelQ, levelP, i, cx, cxInv, decQ + (size_t)i * wq, decP + (size_t)i * wp, npoly, copy_digit)) return rc;
    return RH_OK;
  }
  // a few polys (one ciphertext per call, the way the reference's callers issue work): a digit's launches fill a quarter of the chip and the
  // digit-by-digit sequence is a chain of ~10 dependent launches -- ALL digits in one extension launch (blockIdx.z = digit) and, inside
  // rh_std_ntt_fwd_digits, one launch pair of the transforms (blockIdx.y = digit block)
  bool extended = false;
  const int small = ks_small(RQ, npoly, LQ) ? 1 : 0;        // ringQ's tuning decides for both rings
  if (small && beta > 1) {
    const int rc = rh_bext_decompose_and_split_all(be, levelQ, levelP, LP, beta, cxInv, decQ, wq, decP, wp, npoly);
    if (rc < 0) return rc;
    extended = rc == 0;
  }
  for (int i = 0; i < beta && !extended; ++i)
    if (int rc = rh_bext_decompose_and_split(be, levelQ, levelP, LP, i, cxInv, decQ + (size_t)i * wq, decP + (size_t)i * wp, npoly)) return rc;
  // internal product (copy_digit == false): the blocks feed the key multiply-accumulate only, whose MRedLazy takes any 64-bit
  // operand and whose closing Reduce is canonical -- the transforms skip their final reduction (outputs < 8q)
  const bool lazy = !copy_digit;
  const bool both = small && beta <= 8 && rh_can_ntt_digits(RP) && RP->logN == RQ->logN;      // small batch: the Q blocks and the P blocks in ONE launch pair
  if (both) {
    int g0[8], gl[8];
    for (int j = 0; j < beta; ++j) { int n = LQ - j * LP; if (n > LP) n = LP; if (n < 0) n = 0; g0[j] = j * LP; gl[j] = n; }      // digit j skips its own limbs
    if (int rc = rh_std_ntt_fwd_blocks_small(RQ, decQ, wq, beta, LQ, g0, gl, RP, decP, wp, beta, LP, npoly, lazy)) return rc;
  } else if (int rc = rh_std_ntt_fwd_digits(RQ, decQ, wq, npoly, beta, LQ, LP, lazy, small)) return rc;
  if (copy_digit)
    for (int i = 0; i < beta; ++i) {
      const int st = i * LP; int ed = st + LP; if (ed > LQ) ed = LQ;
      if (ed > st && hipMemcpy2DAsync(decQ + (size_t)i * wq + (size_t)st * N, (size_t)LQ * N * 8, cx + (size_t)st * N, (size_t)LQ * N * 8,
                                      (size_t)(ed - st) * N * 8, npoly, hipMemcpyDeviceToDevice, rh_stream(RQ)) != hipSuccess)
        return rh_fail(RH_ERR_DEVICE, "gadget_product: digit copy failed");
    }
  if (both) return RH_OK;
  if (rh_can_ntt_digits(RP)) return rh_std_ntt_fwd_digits(RP, decP, wp, npoly, beta, LP, 0, lazy, small);   // no limb skipped
  return rh_ring_ntt_any(RP, decP, decP, beta * npoly, LP, 0, false);
}

// the Reduce schedule of gadgetProductMultiplePLazy(Hoisted) (:166-187, :408-428)
struct ReduceSchedule {
  int QiOverF, PiOverF;
  ReduceSchedule(rh_ring* RQ, int levelQ, rh_ring* RP, int levelP)
      : QiOverF(rh_overflow_margin(RQ->moduli, levelQ) >> 1), PiOverF(rh_overflow_margin(RP->moduli, levelP) >> 1) {}
};
static int ks_check(rh_bext* be, int levelQ, int levelP, int beta_key, const char* who, int* beta) {
  if (!be) return rh_fail(RH_ERR_ARG, "%s: null basis extender", who);
  rh_ring* RQ = rh_bext_ringQ(be); rh_ring* RP = rh_bext_ringP(be);
  if (!RP) return rh_fail(RH_ERR_ARG, "%s: basis extender has no P ring", who);
  if (RQ->kind != RP->kind) return rh_fail(RH_ERR_ARG, "%s: ringQ and ringP differ in ring type", who);   // 3N / conjugate-invariant rings: the same steps through the ring's own transform
  if (levelQ < 0 || levelQ >= RQ->L || levelP < 1 || levelP >= RP->L) return rh_fail(RH_ERR_ARG, "%s: need 0 <= levelQ < %d and 1 <= levelP < %d", who, RQ->L, RP->L);
  *beta = (levelQ + levelP + 1) / (levelP + 1);                    // BaseRNSDecompositionVectorSize, params.go:635-642
  if (beta_key >= 0 && *beta > beta_key) return rh_fail(RH_ERR_ARG, "%s: key has %d digits, level needs %d", who, beta_key, *beta);
  (void)hipSetDevice(RQ->device);
  return RH_OK;
}

// Evaluator.DecomposeNTT (:431-453): decompQ [beta][npoly][levelQ+1][N], decompP [beta][npoly][levelP+1][N], NTT domain
extern "C" int rh_bext_decompose_ntt(rh_bext* be, int levelQ, int levelP, const uint64_t* c2, int c2_is_ntt, uint64_t* decompQ,
                                     uint64_t* decompP, int npoly) {
  if (!be || !c2 || !decompQ || !decompP) return rh_fail(RH_ERR_ARG, "decompose_ntt: null argument");
  RhBextGuard guard(be);
  int beta; if (int rc = ks_check(be, levelQ, levelP, -1, "decompose_ntt", &beta)) return rc;
  if (npoly <= 0) return RH_OK;
  rh_ring* RQ = rh_bext_ringQ(be);
  const int LQ = levelQ + 1, LP = levelP + 1, N = RQ->N;
  const size_t wq = (size_t)npoly * LQ * N, wp = (size_t)npoly * LP * N;
  u64* other;
  if (int rc = rh_bext_scratch(be, 2, wq, &other)) return rc;
  // :437-445: the missing domain of c2 goes to BuffInvNTT
  if (int rc = rh_ring_ntt_any(RQ, c2, other, npoly, LQ, 0, c2_is_ntt != 0)) return rc;
  const u64* polyNTT = c2_is_ntt ? c2 : other;
  const u64* polyInv = c2_is_ntt ? other : c2;
  (void)wp;
  return decompose_all_ntt(be, levelQ, levelP, beta, polyNTT, polyInv, decompQ, decompP, npoly, true);
}

// Evaluator.GadgetProductHoisted (:326-349) = gadgetProductMultiplePLazyHoisted (:373-429) + ModDown NTT->NTT (:33-46).
// cx != null (internal, direct product): the digits' own limbs of decompQ were not filled; the multiply-accumulate reads them from cx.
static int hoisted_tail(rh_bext* be, int levelQ, int levelP, const uint64_t* decompQ, const uint64_t* decompP, const uint64_t* evkQ,
                        const uint64_t* evkP, int beta_key, uint64_t* ct0, uint64_t* ct1, int npoly, const uint64_t* cx,
                        const uint64_t* add0 = nullptr, const uint64_t* add1 = nullptr, uint64_t* out0 = nullptr, uint64_t* out1 = nullptr,
                        bool out_ntt = true) {
  // ct0 / ct1 hold the Q-part accumulators; results go to out_c (default: ct_c itself) as [add_c +] ModDown(ct_c, P part)
  if (!out0) out0 = ct0;
  if (!out1) out1 = ct1;
  if (!be || !decompQ || !decompP || !evkQ || !evkP || !ct0 || !ct1) return rh_fail(RH_ERR_ARG, "gadget_product_hoisted: null argument");
  RhBextGuard guard(be);
  int beta; if (int rc = ks_check(be, levelQ, levelP, beta_key, "gadget_product_hoisted", &beta)) return rc;
  if (npoly <= 0) return RH_OK;
  rh_ring* RQ = rh_bext_ringQ(be); rh_ring* RP = rh_bext_ringP(be);
  const int LQ = levelQ + 1, LP = levelP + 1, N = RQ->N;
  const size_t wq = (size_t)npoly * LQ * N, wp = (size_t)npoly * LP * N;
  u64 *aP0, *aP1;                                  // both P-part accumulators back to back: ModDown transforms them as one batch
  if (int rc = rh_bext_scratch(be, 5, 2 * wp, &aP0)) return rc;
  aP1 = aP0 + wp;
  const size_t evq_stride = (size_t)RQ->L * N, evp_stride = (size_t)RP->L * N;
  ReduceSchedule rs(RQ, levelQ, RP, levelP);
  if (int rc = mac_all(RQ, decompQ, wq, evkQ, beta, rs.QiOverF, ct0, ct1, npoly, LQ, cx, LP)) return rc;
  if (int rc = mac_all(RP, decompP, wp, evkP, beta, rs.PiOverF, aP0, aP1, npoly, LP)) return rc;
  if (out_ntt) return rh_bext_moddown_ntt_pair(be, levelQ, levelP, ct0, ct1, aP0, out0, out1, npoly, add0, add1);
  // coefficient-domain ciphertext (:114-118, then ModDown INTT -> INTT :62-66): ringQP.INTT on both components, ModDownQPtoQ
  if (add0 || add1) return rh_fail(RH_ERR_UNSUPPORTED, "gadget product: the fused Add exists for NTT-domain ciphertexts only");
  if (int rc = rh_ring_ntt_any(RQ, ct0, ct0, npoly, LQ, 0, true)) return rc;
  if (int rc = rh_ring_ntt_any(RQ, ct1, ct1, npoly, LQ, 0, true)) return rc;
  if (int rc = rh_ring_ntt_any(RP, aP0, aP0, 2 * npoly, LP, 0, true)) return rc;
  if (int rc = rh_bext_moddown_qp_to_q(be, levelQ, levelP, ct0, aP0, out0, npoly)) return rc;
  return rh_bext_moddown_qp_to_q(be, levelQ, levelP, ct1, aP1, out1, npoly);
}
// GadgetProductHoistedLazy (:351-371) = gadgetProductMultiplePLazyHoisted (:373-429) without the ModDown: the accumulators modulo Q and
// modulo P, canonical after the closing Reduce, P parts as separate blocks ([npoly][levelP+1][N]).  "Lazy" = still scaled by P.
extern "C" int rh_bext_gadget_product_hoisted_lazy(rh_bext* be, int levelQ, int levelP, const uint64_t* decompQ, const uint64_t* decompP,
                                                   const uint64_t* evkQ, const uint64_t* evkP, int beta_key, uint64_t* ctQ0, uint64_t* ctQ1,
                                                   uint64_t* ctP0, uint64_t* ctP1, int npoly) {
  if (!be || !decompQ || !decompP || !evkQ || !evkP || !ctQ0 || !ctQ1 || !ctP0 || !ctP1) return rh_fail(RH_ERR_ARG, "gadget_product_hoisted_lazy: null argument");
  RhBextGuard guard(be);
  int beta; if (int rc = ks_check(be, levelQ, levelP, beta_key, "gadget_product_hoisted_lazy", &beta)) return rc;
  if (npoly <= 0) return RH_OK;
  rh_ring* RQ = rh_bext_ringQ(be); rh_ring* RP = rh_bext_ringP(be);
  const int LQ = levelQ + 1, LP = levelP + 1, N = RQ->N;
  const size_t wq = (size_t)npoly * LQ * N, wp = (size_t)npoly * LP * N;
  ReduceSchedule rs(RQ, levelQ, RP, levelP);
  if (int rc = mac_all(RQ, decompQ, wq, evkQ, beta, rs.QiOverF, ctQ0, ctQ1, npoly, LQ)) return rc;
  return mac_all(RP, decompP, wp, evkP, beta, rs.PiOverF, ctP0, ctP1, npoly, LP);
}
// Evaluator.ModDown (:33-98) for both components of a QP element, NTT -> NTT: ct_c = ModDownQPtoQNTT(ctQ_c, ctP_c); outputs may alias ctQ_c
extern "C" int rh_bext_moddown_qp_to_q_ntt_pair(rh_bext* be, int levelQ, int levelP, const uint64_t* ctQ0, const uint64_t* ctQ1,
                                                const uint64_t* ctP0, const uint64_t* ctP1, uint64_t* ct0, uint64_t* ct1, int npoly) {
  if (!be || !ctQ0 || !ctQ1 || !ctP0 || !ctP1 || !ct0 || !ct1) return rh_fail(RH_ERR_ARG, "moddown pair: null argument");
  if (npoly <= 0) return RH_OK;
  rh_ring* RP = rh_bext_ringP(be);
  if (!RP) return rh_fail(RH_ERR_ARG, "moddown pair: basis extender has no P ring");
  if (levelP < 0 || levelP >= RP->L) return rh_fail(RH_ERR_ARG, "moddown pair: levelP out of range");
  const size_t wp = (size_t)npoly * (levelP + 1) * RP->N;
  if (ctP1 == ctP0 + wp) return rh_bext_moddown_ntt_pair(be, levelQ, levelP, ctQ0, ctQ1, ctP0, ct0, ct1, npoly, nullptr, nullptr);   // back to back: one batch
  if (int rc = rh_bext_moddown_ntt_add(be, levelQ, levelP, ctQ0, ctP0, ct0, npoly, nullptr)) return rc;
  return rh_bext_moddown_ntt_add(be, levelQ, levelP, ctQ1, ctP1, ct1, npoly, nullptr);
}
extern "C" int rh_bext_gadget_product_hoisted(rh_bext* be, int levelQ, int levelP, const uint64_t* decompQ, const uint64_t* decompP,
                                              const uint64_t* evkQ, const uint64_t* evkP, int beta_key, uint64_t* ct0, uint64_t* ct1, int npoly) {
  return hoisted_tail(be, levelQ, levelP, decompQ, decompP, evkQ, evkP, beta_key, ct0, ct1, npoly, nullptr);
}
// hoisted product with the ring.Add that follows it in AutomorphismHoisted (core/rlwe/evaluator_automorphism.go:88-89)
extern "C" int rh_bext_gadget_product_hoisted_then_add(rh_bext* be, int levelQ, int levelP, const uint64_t* decompQ, const uint64_t* decompP,
                                                       const uint64_t* evkQ, const uint64_t* evkP, int beta_key, const uint64_t* add0,
                                                       const uint64_t* add1, uint64_t* ct0, uint64_t* ct1, int npoly) {
  if (!add0 && !add1) return hoisted_tail(be, levelQ, levelP, decompQ, decompP, evkQ, evkP, beta_key, ct0, ct1, npoly, nullptr);
  if (!be || npoly <= 0) return npoly == 0 && be ? RH_OK : rh_fail(RH_ERR_ARG, "gadget_product_hoisted_then_add: bad argument");
  RhBextGuard guard(be);
  rh_ring* RQ = rh_bext_ringQ(be);
  if (levelQ < 0 || levelQ >= RQ->L) return rh_fail(RH_ERR_ARG, "gadget_product_hoisted_then_add: levelQ out of range");
  const size_t wq = (size_t)npoly * (levelQ + 1) * RQ->N;
  u64 *acc0, *acc1;
  if (int rc = rh_bext_scratch(be, 7, wq, &acc0)) return rc;
  if (int rc = rh_bext_scratch(be, 8, wq, &acc1)) return rc;
  return hoisted_tail(be, levelQ, levelP, decompQ, decompP, evkQ, evkP, beta_key, acc0, acc1, npoly, nullptr, add0, add1, ct0, ct1);
}

static int gadget_product_impl(rh_bext* be, int levelQ, int levelP, const uint64_t* cx, const uint64_t* evkQ, const uint64_t* evkP,
                               int beta_key, uint64_t* ct0, uint64_t* ct1, int npoly, const uint64_t* add0, const uint64_t* add1,
                               bool is_ntt = true) {
  if (!be || !cx || !evkQ || !evkP || !ct0 || !ct1) return rh_fail(RH_ERR_ARG, "gadget_product: null argument");
  RhBextGuard guard(be);
  int beta; if (int rc = ks_check(be, levelQ, levelP, beta_key, "gadget_product", &beta)) return rc;
  if (npoly <= 0) return RH_OK;
  rh_ring* RQ = rh_bext_ringQ(be);
  const int LQ = levelQ + 1, LP = levelP + 1, N = RQ->N;
  const size_t wq = (size_t)npoly * LQ * N, wp = (size_t)npoly * LP * N;
  // the whole decomposition stays resident (beta x the size of one digit) so that the key multiply-accumulate is ONE
  // pass with the accumulators in registers: DecomposeNTT (:431-453) then the hoisted product (:373-429) -- the same
  // arithmetic and Reduce schedule as gadgetProductMultiplePLazy (:122-188), digit by digit
  u64 *decQ, *decP;
  if (int rc = rh_bext_scratch(be, 3, (size_t)beta * wq, &decQ)) return rc;
  if (int rc = rh_bext_scratch(be, 4, (size_t)beta * wp, &decP)) return rc;
  u64* other;
  if (int rc = rh_bext_scratch(be, 2, wq, &other)) return rc;
  // ctQP.IsNTT: cxNTT = cx, cxInvNTT = INTT(cx) (:134-138); else cxInvNTT = cx, cxNTT = NTT(cx) (:139-143)
  if (int rc = rh_ring_ntt_any(RQ, cx, other, npoly, LQ, 0, is_ntt)) return rc;
  const u64* cxNTT = is_ntt ? cx : other; const u64* cxInv = is_ntt ? other : cx;
  if (int rc = decompose_all_ntt(be, levelQ, levelP, beta, cxNTT, cxInv, decQ, decP, npoly, false)) return rc;
  if (!is_ntt) return hoisted_tail(be, levelQ, levelP, decQ, decP, evkQ, evkP, beta_key, ct0, ct1, npoly, cxNTT, nullptr, nullptr, nullptr, nullptr, false);
  if (!add0 && !add1) return hoisted_tail(be, levelQ, levelP, decQ, decP, evkQ, evkP, beta_key, ct0, ct1, npoly, cx);
  u64 *acc0, *acc1;                                  // accumulate beside the outputs: they may alias the addends (or cx)
  if (int rc = rh_bext_scratch(be, 7, wq, &acc0)) return rc;
  if (int rc = rh_bext_scratch(be, 8, wq, &acc1)) return rc;
  return hoisted_tail(be, levelQ, levelP, decQ, decP, evkQ, evkP, beta_key, acc0, acc1, npoly, cx, add0, add1, ct0, ct1);
}
extern "C" int rh_bext_gadget_product(rh_bext* be, int levelQ, int levelP, const uint64_t* cx, const uint64_t* evkQ,
                                      const uint64_t* evkP, int beta_key, uint64_t* ct0, uint64_t* ct1, int npoly) {
  return gadget_product_impl(be, levelQ, levelP, cx, evkQ, evkP, beta_key, ct0, ct1, npoly, nullptr, nullptr);
}
// ct_c = add_c + GadgetProduct(cx)_c (ring.Add, canonical): the Add that follows the product in Relinearize / mulRelin /
// applyEvaluationKey / Automorphism rides in ModDown's tile epilogue.  add0 / add1 may be NULL (no addend for that component)
// and may alias ct0 / ct1 (the accumulation happens in scratch; outputs are written last).
extern "C" int rh_bext_gadget_product_then_add(rh_bext* be, int levelQ, int levelP, const uint64_t* cx, const uint64_t* evkQ,
                                               const uint64_t* evkP, int beta_key, const uint64_t* add0, const uint64_t* add1,
                                               uint64_t* ct0, uint64_t* ct1, int npoly) {
  return gadget_product_impl(be, levelQ, levelP, cx, evkQ, evkP, beta_key, ct0, ct1, npoly, add0, add1);
}

// rlwe.Evaluator.GadgetProduct for a COEFFICIENT-domain ciphertext (ct.IsNTT == false), levelP >= 1: gadgetProductMultiplePLazy with
// cxInvNTT = cx and cxNTT = NTT(cx) (:139-143), the accumulators brought back with ringQP.INTT (:114-118), ModDown INTT -> INTT
// = ModDownQPtoQ on both components (:62-66).  cx, ct0, ct1: coefficient domain.
extern "C" int rh_bext_gadget_product_coeff(rh_bext* be, int levelQ, int levelP, const uint64_t* cx, const uint64_t* evkQ,
                                            const uint64_t* evkP, int beta_key, uint64_t* ct0, uint64_t* ct1, int npoly) {
  return gadget_product_impl(be, levelQ, levelP, cx, evkQ, evkP, beta_key, ct0, ct1, npoly, nullptr, nullptr, false);
}

// ---- gadgetProductSinglePAndBitDecompLazy (:190-324) + ModDown (:33-98): gadget ciphertexts with at most one P modulus, optionally
// with a power-of-two decomposition on top of the RNS one.  One digit per Q modulus i; pw2 == 0: the digit is the sign-aware
// re-embedding of limb i (DecomposeAndSplit, single-prime branch); pw2 > 0: digits_per_limb[i] base-2^pw2 windows of limb i
// (ring.MaskVec), the same small vector under every modulus.  Each digit is transformed and multiply-accumulated against its key
// row with the reference's Reduce schedule.  The reference transforms with NTTLazy and accumulates MRedLazy values; the closing
// Reduce / ModDown outputs are canonical, so the canonical forward transform used here gives the same bits.
// The window of limb i is < q_i (or < 2^pw2), which may exceed another modulus q_u by any factor (mixed-size chains, e.g. 60- and
// 40-bit primes): each target limb gets the canonical residue BRedAdd(v, q_u) (ring/modular_reduction.go:110-117), inside the
// range the forward transform's first stage assumes.  The reference feeds the raw window to NTTLazy; its closing Reduce makes both
// routes agree bit for bit.
__global__ void __launch_bounds__(256)
mask_broadcast_kernel(const u64* in, int in_rows, int src_limb, int shift, u64 mask, u64* outQ, int LQ, u64* outP, int LP, int N,
                      const LimbConsts* __restrict__ cQ, const LimbConsts* __restrict__ cP) {
  const int k = blockIdx.x * 256 + threadIdx.x, poly = blockIdx.y;
  if (k >= N) return;
  const u64 v = (in[((size_t)poly * in_rows + src_limb) * N + k] >> shift) & mask;           // MaskVec (ring/vec_ops.go:870)
  for (int u = 0; u < LQ; ++u) outQ[((size_t)poly * LQ + u) * N + k] = bred_add(v, cQ[u].q, cQ[u].bred0);
  for (int u = 0; u < LP; ++u) outP[((size_t)poly * LP + u) * N + k] = bred_add(v, cP[u].q, cP[u].bred0);
}
static int reduce_pair(rh_ring* R, int L, u64* a0, u64* a1, int npoly) {
  if (int rc = rh_vec_launch(R, RH_OP_REDUCE, a0, nullptr, a0, npoly, L, 0, nullptr, nullptr)) return rc;
  return rh_vec_launch(R, RH_OP_REDUCE, a1, nullptr, a1, npoly, L, 0, nullptr, nullptr);
}
// lazy = false: the whole product (ModDown / CopyLvl at the end, ctP0 / ctP1 unused);  lazy = true: stop after the closing Reduce, the
// accumulators modulo Q in ct0 / ct1 and modulo P in ctP0 / ctP1 (canonical residues, NTT domain).  raw_limb_digits: the digits of
// rgsw's externalProductInPlaceSinglePAndBitDecomp (core/rgsw/evaluator.go:119-186): MaskVec of limb i even without a power-of-two
// decomposition (mask = all ones: the limb's own residues under every modulus, no DecomposeAndSplit).
static int single_p_core(rh_bext* be, int levelQ, int levelP, const uint64_t* cx, int cx_is_ntt, int pw2,
                         const int* digits_per_limb, const uint64_t* evkQ, const uint64_t* evkP, int key_rows,
                         uint64_t* ct0, uint64_t* ct1, int npoly, bool lazy, bool raw_limb_digits, uint64_t* ctP0, uint64_t* ctP1);
extern "C" int rh_bext_gadget_product_single_p(rh_bext* be, int levelQ, int levelP, const uint64_t* cx, int cx_is_ntt, int pw2,
                                               const int* digits_per_limb, const uint64_t* evkQ, const uint64_t* evkP, int key_rows,
                                               uint64_t* ct0, uint64_t* ct1, int npoly) {
  return single_p_core(be, levelQ, levelP, cx, cx_is_ntt, pw2, digits_per_limb, evkQ, evkP, key_rows, ct0, ct1, npoly, false, false, nullptr, nullptr);
}
extern "C" int rh_bext_gadget_product_single_p_lazy(rh_bext* be, int levelQ, int levelP, const uint64_t* cx, int cx_is_ntt, int pw2,
                                                    const int* digits_per_limb, const uint64_t* evkQ, const uint64_t* evkP, int key_rows,
                                                    int raw_limb_digits, uint64_t* ctQ0, uint64_t* ctQ1, uint64_t* ctP0, uint64_t* ctP1, int npoly) {
  if (levelP == 0 && (!ctP0 || !ctP1)) return rh_fail(RH_ERR_ARG, "gadget_product_single_p_lazy: levelP = 0 needs the P accumulators");
  return single_p_core(be, levelQ, levelP, cx, cx_is_ntt, pw2, digits_per_limb, evkQ, evkP, key_rows, ctQ0, ctQ1, npoly, true, raw_limb_digits != 0, ctP0, ctP1);
}
static int single_p_core(rh_bext* be, int levelQ, int levelP, const uint64_t* cx, int cx_is_ntt, int pw2,
                         const int* digits_per_limb, const uint64_t* evkQ, const uint64_t* evkP, int key_rows,
                         uint64_t* ct0, uint64_t* ct1, int npoly, bool lazy, bool raw_limb_digits, uint64_t* ctP0, uint64_t* ctP1) {
  if (!be || !cx || !evkQ || !ct0 || !ct1) return rh_fail(RH_ERR_ARG, "gadget_product_single_p: null argument");
  RhBextGuard guard(be);
  rh_ring* RQ = rh_bext_ringQ(be); rh_ring* RP = rh_bext_ringP(be);
  if (RP && RQ->kind != RP->kind) return rh_fail(RH_ERR_ARG, "gadget_product_single_p: ringQ and ringP differ in ring type");
  if (levelQ < 0 || levelQ >= RQ->L) return rh_fail(RH_ERR_ARG, "gadget_product_single_p: levelQ %d out of range [0,%d)", levelQ, RQ->L);
  if (levelP != 0 && levelP != -1) return rh_fail(RH_ERR_ARG, "gadget_product_single_p: levelP must be 0 or -1 (levelP >= 1: rh_bext_gadget_product)");
  if (levelP == 0 && (!RP || !evkP)) return rh_fail(RH_ERR_ARG, "gadget_product_single_p: levelP = 0 needs a P ring and the key's P part");
  if (pw2 < 0 || pw2 > 63 || (pw2 > 0 && !digits_per_limb)) return rh_fail(RH_ERR_ARG, "gadget_product_single_p: bad BaseTwoDecomposition");
  // pw2 == 0 without a P modulus: the reference calls DecomposeAndSplit with nbPi = levelP + 1 = 0, which degenerates (every digit
  // reads limb 0); such gadget ciphertexts are built with a power-of-two decomposition (core/rlwe/params.go:615-633)
  if (pw2 == 0 && levelP < 0 && !raw_limb_digits) return rh_fail(RH_ERR_UNSUPPORTED, "gadget_product_single_p: no P modulus needs BaseTwoDecomposition > 0");
  if (npoly <= 0) return RH_OK;
  if (npoly > 65535) return rh_fail(RH_ERR_ARG, "gadget_product_single_p: at most 65535 polys per call (split the batch)");
  (void)hipSetDevice(RQ->device);
  (void)hipGetLastError();
  const int LQ = levelQ + 1, LP = levelP + 1, N = RQ->N;
  const size_t wq = (size_t)npoly * LQ * N, wp = (size_t)npoly * LP * N;
  int rows_needed = 0;
  for (int i = 0; i < LQ; ++i) rows_needed += pw2 ? digits_per_limb[i] : 1;
  if (key_rows < rows_needed) return rh_fail(RH_ERR_ARG, "gadget_product_single_p: key has %d rows, level needs %d", key_rows, rows_needed);
  u64 *cxInvBuf, *c2Q, *c2P = nullptr, *aP0 = nullptr, *aP1 = nullptr;
  if (int rc = rh_bext_scratch(be, 3, wq, &c2Q)) return rc;
  if (LP) {
    if (int rc = rh_bext_scratch(be, 4, wp, &c2P)) return rc;
    if (lazy) { aP0 = ctP0; aP1 = ctP1; }
    else { if (int rc = rh_bext_scratch(be, 5, 2 * wp, &aP0)) return rc; aP1 = aP0 + wp; }
  }
  const bool maskform = pw2 != 0 || raw_limb_digits;                  // digits by MaskVec of the limb (rlwe with pw2 > 0; rgsw always)
  const u64 mask_all = ~(u64)0;
  const u64* cxInv = cx;
  if (cx_is_ntt) {
    if (int rc = rh_bext_scratch(be, 2, wq, &cxInvBuf)) return rc;
    if (int rc = rh_ring_ntt_any(RQ, cx, cxInvBuf, npoly, LQ, 0, true)) return rc;          // ringQ.INTT(cx, cxInvNTT) (:201-203)
    cxInv = cxInvBuf;
  }
  const int QiOverF = rh_overflow_margin(RQ->moduli, levelQ) >> 1;
  const int PiOverF = LP ? rh_overflow_margin(RP->moduli, levelP) >> 1 : 1;
  const size_t evq = (size_t)RQ->L * N, evp = LP ? (size_t)RP->L * N : 0;
  const u64 mask = pw2 ? (((u64)1 << pw2) - 1) : 0;
  const dim3 grid((N + 255) / 256, npoly);                          // polys on gridDim.y: a batch is far below its 65535 cap
  int e = 0, reduce = 0;
  for (int i = 0; i < LQ; ++i) {
    const int nd = pw2 ? digits_per_limb[i] : 1;
    if (!maskform) if (int rc = rh_bext_decompose_and_split(be, levelQ, levelP, LP, i, cxInv, c2Q, c2P, npoly)) return rc;   // (:243-245)
    for (int j = 0; j < nd; ++j, ++e) {
      if (maskform) {
        mask_broadcast_kernel<<<grid, 256, 0, rh_stream(RQ)>>>(cxInv, LQ, i, j * pw2, pw2 ? mask : mask_all, c2Q, LQ, c2P, LP, N, RQ->d_consts, LP ? RP->d_consts : nullptr);   // (:249-252)
        if (hipGetLastError() != hipSuccess) return rh_fail(RH_ERR_DEVICE, "mask_broadcast_kernel launch failed");
      }
      if (maskform || j == 0) {                                        // s.NTTLazy under every modulus (:258-262, :285-289)
        if (int rc = rh_ring_ntt_any(RQ, c2Q, c2Q, npoly, LQ, 0, false)) return rc;
        if (LP) if (int rc = rh_ring_ntt_any(RP, c2P, c2P, npoly, LP, 0, false)) return rc;
      }
      if (int rc = rh_gadget_mac(RQ, c2Q, evkQ + ((size_t)e * 2) * evq, evkQ + ((size_t)e * 2 + 1) * evq, ct0, ct1, npoly, LQ, e == 0)) return rc;
      if (LP) if (int rc = rh_gadget_mac(RP, c2P, evkP + ((size_t)e * 2) * evp, evkP + ((size_t)e * 2 + 1) * evp, aP0, aP1, npoly, LP, e == 0)) return rc;
      if (reduce % QiOverF == QiOverF - 1) if (int rc = reduce_pair(RQ, LQ, ct0, ct1, npoly)) return rc;
      if (LP && reduce % PiOverF == PiOverF - 1) if (int rc = reduce_pair(RP, LP, aP0, aP1, npoly)) return rc;
      ++reduce;
    }
  }
  if (reduce % QiOverF != 0) if (int rc = reduce_pair(RQ, LQ, ct0, ct1, npoly)) return rc;
  if (LP && reduce % PiOverF != 0) if (int rc = reduce_pair(RP, LP, aP0, aP1, npoly)) return rc;
  if (lazy) return RH_OK;
  if (cx_is_ntt) {
    if (!LP) return RH_OK;                                             // levelP = -1, NTT -> NTT: CopyLvl (:72-75)
    return rh_bext_moddown_ntt_pair(be, levelQ, levelP, ct0, ct1, aP0, ct0, ct1, npoly, nullptr, nullptr);
  }
  // coefficient-domain ciphertext: ringQP.INTT (:114-118), then ModDownQPtoQ / plain copy
  if (int rc = rh_ring_ntt_any(RQ, ct0, ct0, npoly, LQ, 0, true)) return rc;
  if (int rc = rh_ring_ntt_any(RQ, ct1, ct1, npoly, LQ, 0, true)) return rc;
  if (!LP) return RH_OK;
  if (int rc = rh_ring_ntt_any(RP, aP0, aP0, 2 * npoly, LP, 0, true)) return rc;
  if (int rc = rh_bext_moddown_qp_to_q(be, levelQ, levelP, ct0, aP0, ct0, npoly)) return rc;
  return rh_bext_moddown_qp_to_q(be, levelQ, levelP, ct1, aP1, ct1, npoly);
}
