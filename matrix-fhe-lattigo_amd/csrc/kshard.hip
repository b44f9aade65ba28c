// kshard.hip -- limb-sharded hybrid key switch (BASELINE config 5): one rank owns a subset of the limbs of
// Q and P (and the matching slice of the evaluation key) and computes only those limbs of the gadget product.
//
// Same arithmetic as rh_bext_gadget_product (keyswitch.hip) = rlwe.Evaluator.GadgetProduct
// (core/rlwe/evaluator_gadget_product.go:16-30, :122-188, :33-46, :455-478), cut at the two places where a coefficient
// needs limbs of other owners (reconstructRNS, ring/basis_extension.go:550-594):
//   * per digit: the digit's alpha source limbs of INTT(cx)      -> all-gather, then rh_kshard_digit
//   * ModDown:   all k+1 limbs of the P part (coefficient domain) -> all-gather, then rh_kshard_moddown
// The exchange itself is the host's (RCCL all-gather through torch.distributed, matrix-fhe-lattigo_amd/sharding.py);
// this file never communicates.  Every limb this rank produces is bit-identical to the same limb of the unsharded
// product (same kernels, same constants, same Reduce schedule), which is what tests/test_gpu_kshard.py checks.
//
// Layout: local rings hold the OWNED moduli only, in ascending global order; local blocks are (poly, owned limb, N).
// Gathered source blocks are (poly, source limb in global order, N).
#include <hip/hip_runtime.h>
#include <vector>
#include <map>
#include "engine_internal.hpp"
#include "bext_internal.hpp"
#include "hostmath.hpp"

struct rh_kshard {
  rh_ring* Q = nullptr; rh_ring* P = nullptr;            // local rings (P may be null: this rank owns no P limb)
  std::vector<u64> allQ, allP;                           // moduli 0..levelQ / 0..levelP of the full chain
  std::vector<int> ownQ, ownP;                           // global indices of the owned limbs, ascending
  int levelQ = 0, levelP = 0, beta = 0;
  std::map<int, BextPlan> digit_plans;
  BextPlan md_plan; bool have_md = false;
  std::vector<u64> md_scalars;                           // q_k - (P^-1 mod q_k) Montgomery form, per owned Q limb
  // scratch per pipeline slot (rh_kshard_gadget_product runs chunks of the batch on two streams): 0 c2Q, 1 c2P, 2 buffQ; 3, 4: all digits'
  // c2Q / c2P (rh_kshard_product); 5 cxinv, 6 accP (both components), 7 gathered source limbs in chain order, 8 send, 9 recv (exchange)
  u64* buf[2][10] = {}; size_t buf_words[2][10] = {};
  int slot = 0;                                          // slot the calling orchestrator is enqueueing (under mu)
  int reduce = 0, QiOverF = 1, PiOverF = 1;
  // whole-product orchestration (rh_kshard_gadget_product): who owns which limb of Q ++ P, side streams for the chunk pipeline
  int world = 1, rank = 0;
  std::vector<int> owner;                                // owner[i] for i < levelQ+1 (Q), owner[levelQ+1+j] (P)
  hipStream_t side[2] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
  u64* arena = nullptr; size_t arena_words = 0;          // caller-registered exchange memory (send / recv blocks are carved from it)
  std::recursive_mutex mu;                               // one key switch at a time per handle (digits are fed in order)
};

static int ks_buf(rh_kshard* ks, int which, size_t words, u64** out) {
  const int sl = ks->slot;
  if (ks->buf_words[sl][which] < words) {
    if (ks->buf[sl][which]) (void)hipFree(ks->buf[sl][which]);           // hipFree waits for the device: never frees under a running kernel
    ks->buf[sl][which] = nullptr; ks->buf_words[sl][which] = 0;
    if (hipMalloc((void**)&ks->buf[sl][which], (words ? words : 1) * 8) != hipSuccess) return rh_fail(RH_ERR_NOMEM, "hipMalloc(key-switch shard scratch) failed");
    ks->buf_words[sl][which] = words;
  }
  *out = ks->buf[sl][which];
  return 0;
}

extern "C" int rh_kshard_create(rh_kshard** out, rh_ring* ringQ_loc, rh_ring* ringP_loc, const uint64_t* allQ, int levelQ,
                                const uint64_t* allP, int levelP, const int* ownQ, int nownQ, const int* ownP, int nownP) {
  if (!out || !ringQ_loc || !allQ || !allP || !ownQ || nownQ < 1) return rh_fail(RH_ERR_ARG, "rh_kshard_create: null argument or no owned Q limb");
  if (levelQ < 0 || levelP < 1 || levelQ + 1 > RH_MAX_LIMBS || levelP + 1 > 32) return rh_fail(RH_ERR_ARG, "rh_kshard_create: need levelQ >= 0 and 1 <= levelP < 32");
  if (ringQ_loc->kind != RH_RING_STANDARD || (ringP_loc && ringP_loc->kind != RH_RING_STANDARD)) return rh_fail(RH_ERR_UNSUPPORTED, "rh_kshard_create: standard rings only");
  if (ringQ_loc->L != nownQ) return rh_fail(RH_ERR_ARG, "rh_kshard_create: local Q ring has %d limbs, %d owned", ringQ_loc->L, nownQ);
  if ((nownP > 0) != (ringP_loc != nullptr) || (ringP_loc && ringP_loc->L != nownP)) return rh_fail(RH_ERR_ARG, "rh_kshard_create: local P ring does not match the %d owned P limbs", nownP);
  if (ringP_loc && (ringP_loc->N != ringQ_loc->N || ringP_loc->device != ringQ_loc->device)) return rh_fail(RH_ERR_ARG, "rh_kshard_create: local rings differ in N or device");
  for (int k = 0; k < nownQ; ++k) {
    if (ownQ[k] < 0 || ownQ[k] > levelQ || (k && ownQ[k] <= ownQ[k - 1])) return rh_fail(RH_ERR_ARG, "rh_kshard_create: owned Q indices must be ascending in [0,%d]", levelQ);
    if (ringQ_loc->moduli[k] != allQ[ownQ[k]]) return rh_fail(RH_ERR_MODULUS, "rh_kshard_create: local Q limb %d is not Q[%d]", k, ownQ[k]);
  }
  for (int k = 0; k < nownP; ++k) {
    if (ownP[k] < 0 || ownP[k] > levelP || (k && ownP[k] <= ownP[k - 1])) return rh_fail(RH_ERR_ARG, "rh_kshard_create: owned P indices must be ascending in [0,%d]", levelP);
    if (ringP_loc->moduli[k] != allP[ownP[k]]) return rh_fail(RH_ERR_MODULUS, "rh_kshard_create: local P limb %d is not P[%d]", k, ownP[k]);
  }
  rh_kshard* ks = new rh_kshard();
  ks->Q = ringQ_loc; ks->P = ringP_loc;
  ks->allQ.assign(allQ, allQ + levelQ + 1); ks->allP.assign(allP, allP + levelP + 1);
  ks->ownQ.assign(ownQ, ownQ + nownQ); if (nownP) ks->ownP.assign(ownP, ownP + nownP);
  ks->levelQ = levelQ; ks->levelP = levelP;
  ks->beta = (levelQ + levelP + 1) / (levelP + 1);                      // BaseRNSDecompositionVectorSize, core/rlwe/params.go:635-642
  ks->QiOverF = rh_overflow_margin(ks->allQ, levelQ) >> 1;             // margins of the FULL chain: same Reduce schedule
  ks->PiOverF = rh_overflow_margin(ks->allP, levelP) >> 1;
  ks->md_scalars.resize(nownQ);
  for (int k = 0; k < nownQ; ++k) ks->md_scalars[k] = ringQ_loc->moduli[k] - rh_moddown_const(ks->allP, ringQ_loc->moduli[k]);
  *out = ks;
  return RH_OK;
}

extern "C" void rh_kshard_destroy(rh_kshard* ks) {
  if (!ks) return;
  for (auto& kv : ks->digit_plans) rh_bext_free_plan(kv.second);
  if (ks->have_md) rh_bext_free_plan(ks->md_plan);
  for (int sl = 0; sl < 2; ++sl) for (int i = 0; i < 10; ++i) if (ks->buf[sl][i]) (void)hipFree(ks->buf[sl][i]);
  for (int k = 0; k < 2; ++k) { if (ks->side[k]) (void)hipStreamDestroy(ks->side[k]); if (ks->ev_join[k]) (void)hipEventDestroy(ks->ev_join[k]); }
  if (ks->ev_fork) (void)hipEventDestroy(ks->ev_fork);
  delete ks;
}

extern "C" int rh_kshard_num_digits(const rh_kshard* ks) { return ks ? ks->beta : 0; }

// global limb range [st, ed) of digit i (DecomposeSingleNTT :455-478 with nbPi = levelP + 1)
extern "C" int rh_kshard_digit_range(const rh_kshard* ks, int digit, int* st, int* ed) {
  if (!ks || !st || !ed || digit < 0 || digit >= ks->beta) return rh_fail(RH_ERR_ARG, "rh_kshard_digit_range: bad digit");
  const int LP = ks->levelP + 1, LQ = ks->levelQ + 1;
  *st = digit * LP; *ed = *st + LP > LQ ? LQ : *st + LP;
  return RH_OK;
}

static int digit_plan(rh_kshard* ks, int digit, BextPlan** out, bool* single) {
  const int nbPi = ks->levelP + 1, levelQ = ks->levelQ;
  const int st = digit * nbPi; int ed = st + nbPi; if (ed > levelQ + 1) ed = levelQ + 1;
  const int decompLvl = (levelQ > nbPi * (digit + 1) - 1) ? nbPi - 2 : (levelQ % nbPi) - 1;      // basis_extension.go:394-399
  *single = decompLvl < 0;
  auto it = ks->digit_plans.find(digit);
  if (it != ks->digit_plans.end()) { *out = &it->second; return 0; }
  BextPlan p;
  rh_ring* RQ = ks->Q; rh_ring* RP = ks->P;
  if (decompLvl < 0) {                                                    // single-prime digit: sign-aware copy (:402-436)
    std::vector<SignTarget> T;
    for (size_t k = 0; k < ks->ownQ.size(); ++k) T.push_back(SignTarget{RQ->moduli[k], RQ->bred[2 * k], 0, (int)k});
    for (size_t k = 0; k < ks->ownP.size(); ++k) T.push_back(SignTarget{RP->moduli[k], RP->bred[2 * k], 1, (int)k});
    if (int rc = rh_bext_upload_sign_plan(p, T, ks->allQ[st])) return rc;
  } else {
    std::vector<u64> Qs(ks->allQ.begin() + st, ks->allQ.begin() + ed), tg;
    std::vector<BextTarget> T;
    for (size_t k = 0; k < ks->ownQ.size(); ++k) {
      if (ks->ownQ[k] >= st && ks->ownQ[k] < ed) continue;              // digit limbs come from the NTT-domain input (:467-468)
      BextTarget t{}; t.p = RQ->moduli[k]; t.pinv = RQ->mred[k]; t.half = rh_half_product_mod(Qs, t.p);
      t.buf = 0; t.limb = (int)k; t.post = 1; t.skip = 0;
      T.push_back(t); tg.push_back(t.p);
    }
    for (size_t k = 0; k < ks->ownP.size(); ++k) {
      BextTarget t{}; t.p = RP->moduli[k]; t.pinv = RP->mred[k]; t.half = rh_half_product_mod(Qs, t.p);
      t.buf = 1; t.limb = (int)k; t.post = 1; t.skip = 0;
      T.push_back(t); tg.push_back(t.p);
    }
    std::vector<u64> qsi, coef, vt;
    rh_gen_modup(Qs, tg, qsi, coef, vt);
    std::vector<BextSource> S(Qs.size());
    for (size_t i = 0; i < Qs.size(); ++i) S[i] = BextSource{Qs[i], rh::gen_mred_constant(Qs[i]), qsi[i], rh_half_product_mod(Qs, Qs[i])};
    if (int rc = rh_bext_upload_plan(p, S, T, coef, vt)) return rc;
  }
  *out = &ks->digit_plans.emplace(digit, p).first->second;
  return 0;
}

static int reduce_accs(rh_kshard* ks, bool q, uint64_t* a0, uint64_t* a1, int npoly) {
  rh_ring* R = q ? ks->Q : ks->P;
  if (!R) return RH_OK;
  if (int rc = rh_vec_launch(R, RH_OP_REDUCE, a0, nullptr, a0, npoly, R->L, 0, nullptr, nullptr)) return rc;
  return rh_vec_launch(R, RH_OP_REDUCE, a1, nullptr, a1, npoly, R->L, 0, nullptr, nullptr);
}

// One digit of gadgetProductMultiplePLazy (:154-175) for the owned limbs.  src: (npoly, ed-st, N) = limbs [st, ed) of
// INTT(cx) gathered from their owners; cx_loc: this rank's limbs of the NTT-domain input; evk*_loc: [digit][2][owned
// limb][N].  Digits must be fed in order 0 .. beta-1; the last one applies the closing Reduce (:177-187).
extern "C" int rh_kshard_digit(rh_kshard* ks, int digit, const uint64_t* src, const uint64_t* cx_loc, const uint64_t* evkQ_loc,
                               const uint64_t* evkP_loc, uint64_t* ct0_loc, uint64_t* ct1_loc, uint64_t* accP0_loc,
                               uint64_t* accP1_loc, int npoly) {
  if (!ks || !src || !cx_loc || !evkQ_loc || !ct0_loc || !ct1_loc) return rh_fail(RH_ERR_ARG, "rh_kshard_digit: null argument");
  if (ks->P && (!evkP_loc || !accP0_loc || !accP1_loc)) return rh_fail(RH_ERR_ARG, "rh_kshard_digit: null P-part argument");
  if (digit < 0 || digit >= ks->beta) return rh_fail(RH_ERR_ARG, "rh_kshard_digit: digit %d out of range [0,%d)", digit, ks->beta);
  if (npoly <= 0) return RH_OK;
  rh_ring* RQ = ks->Q; rh_ring* RP = ks->P;
  (void)hipSetDevice(RQ->device);
  std::lock_guard<std::recursive_mutex> lk(ks->mu);
  RhCallScope scope(rh_stream(RQ));                        // both local rings launch into ringQ's stream
  const int N = RQ->N, nQ = RQ->L, nP = RP ? RP->L : 0;
  int st, ed; (void)rh_kshard_digit_range(ks, digit, &st, &ed);
  u64 *c2Q, *c2P = nullptr;
  if (int rc = ks_buf(ks, 0, (size_t)npoly * nQ * N, &c2Q)) return rc;
  if (nP) if (int rc = ks_buf(ks, 1, (size_t)npoly * nP * N, &c2P)) return rc;
  BextPlan* p; bool single;
  if (int rc = digit_plan(ks, digit, &p, &single)) return rc;
  if (single) { if (int rc = rh_bext_launch_sign(rh_stream(RQ), N, *p, src, ed - st, 0, c2Q, nQ, c2P, nP, npoly)) return rc; }
  else if (int rc = rh_bext_launch_raw(rh_stream(RQ), N, *p, src, ed - st, 0, c2Q, nQ, c2P, nP, nullptr, 0, npoly, BEXT_ADD_RAW)) return rc;
  if (int rc = rh_std_ntt_launch(RQ, c2Q, c2Q, npoly, nQ, 0, false, false, 0)) return rc;
  for (int k = 0; k < nQ; ++k) {
    if (ks->ownQ[k] < st || ks->ownQ[k] >= ed) continue;
    if (hipMemcpy2DAsync(c2Q + (size_t)k * N, (size_t)nQ * N * 8, cx_loc + (size_t)k * N, (size_t)nQ * N * 8, (size_t)N * 8, npoly,
                         hipMemcpyDeviceToDevice, rh_stream(RQ)) != hipSuccess) return rh_fail(RH_ERR_DEVICE, "rh_kshard_digit: digit copy failed");
  }
  if (nP) if (int rc = rh_std_ntt_launch(RP, c2P, c2P, npoly, nP, 0, false, false, 0)) return rc;
  const size_t evq = (size_t)nQ * N, evp = (size_t)nP * N;
  if (digit == 0) ks->reduce = 0;
  if (int rc = rh_gadget_mac(RQ, c2Q, evkQ_loc + ((size_t)digit * 2) * evq, evkQ_loc + ((size_t)digit * 2 + 1) * evq, ct0_loc, ct1_loc, npoly, nQ, digit == 0)) return rc;
  if (nP) if (int rc = rh_gadget_mac(RP, c2P, evkP_loc + ((size_t)digit * 2) * evp, evkP_loc + ((size_t)digit * 2 + 1) * evp, accP0_loc, accP1_loc, npoly, nP, digit == 0)) return rc;
  if (ks->reduce % ks->QiOverF == ks->QiOverF - 1) if (int rc = reduce_accs(ks, true, ct0_loc, ct1_loc, npoly)) return rc;
  if (ks->reduce % ks->PiOverF == ks->PiOverF - 1) if (int rc = reduce_accs(ks, false, accP0_loc, accP1_loc, npoly)) return rc;
  ++ks->reduce;
  if (digit == ks->beta - 1) {
    if (ks->reduce % ks->QiOverF != 0) if (int rc = reduce_accs(ks, true, ct0_loc, ct1_loc, npoly)) return rc;
    if (ks->reduce % ks->PiOverF != 0) if (int rc = reduce_accs(ks, false, accP0_loc, accP1_loc, npoly)) return rc;
  }
  return RH_OK;
}

// All digits of gadgetProductMultiplePLazy (:154-188) for the owned limbs in one call.  src_all: (npoly, levelQ + 1, N) = EVERY limb of
// INTT(cx) in chain order (one all-gather of the owners' limbs instead of one per digit); the other arguments as for rh_kshard_digit.
// Per digit the basis extension onto the owned limbs, then ONE pipelined transform of all digit blocks (each skips the digit's own
// limbs, which the multiply-accumulate reads from cx_loc) and one multiply-accumulate over all digits with the accumulators in
// registers and the reference's Reduce schedule: the structure of the single-GPU product (keyswitch.hip), same bits.
extern "C" int rh_kshard_product(rh_kshard* ks, const uint64_t* src_all, const uint64_t* cx_loc, const uint64_t* evkQ_loc,
                                 const uint64_t* evkP_loc, uint64_t* ct0_loc, uint64_t* ct1_loc, uint64_t* accP0_loc, uint64_t* accP1_loc,
                                 int npoly) {
  if (!ks || !src_all || !cx_loc || !evkQ_loc || !ct0_loc || !ct1_loc) return rh_fail(RH_ERR_ARG, "rh_kshard_product: null argument");
  if (ks->P && (!evkP_loc || !accP0_loc || !accP1_loc)) return rh_fail(RH_ERR_ARG, "rh_kshard_product: null P-part argument");
  if (npoly <= 0) return RH_OK;
  rh_ring* RQ = ks->Q; rh_ring* RP = ks->P;
  (void)hipSetDevice(RQ->device);
  std::lock_guard<std::recursive_mutex> lk(ks->mu);
  RhCallScope scope(rh_stream(RQ));
  const int N = RQ->N, nQ = RQ->L, nP = RP ? RP->L : 0, beta = ks->beta, LQ = ks->levelQ + 1;
  const size_t wq = (size_t)npoly * nQ * N, wp = (size_t)npoly * nP * N;
  u64 *c2Q, *c2P = nullptr;
  if (int rc = ks_buf(ks, 3, (size_t)beta * wq, &c2Q)) return rc;
  if (nP) if (int rc = ks_buf(ks, 4, (size_t)beta * wp, &c2P)) return rc;
  std::vector<int> gap0(beta), gap_len(beta), own_digit(nQ, -1), zero(beta, 0);
  for (int d = 0; d < beta; ++d) {
    int st, ed; (void)rh_kshard_digit_range(ks, d, &st, &ed);
    BextPlan* p; bool single;
    if (int rc = digit_plan(ks, d, &p, &single)) return rc;
    u64* q = c2Q + (size_t)d * wq; u64* pp = nP ? c2P + (size_t)d * wp : nullptr;
    if (single) { if (int rc = rh_bext_launch_sign(rh_stream(RQ), N, *p, src_all, LQ, st, q, nQ, pp, nP, npoly)) return rc; }
    else if (int rc = rh_bext_launch_raw(rh_stream(RQ), N, *p, src_all, LQ, st, q, nQ, pp, nP, nullptr, 0, npoly, BEXT_ADD_RAW)) return rc;
    gap0[d] = nQ; gap_len[d] = 0;                       // the owned limbs of the digit are a contiguous run of local indices (ownQ ascends)
    for (int k = 0; k < nQ; ++k)
      if (ks->ownQ[k] >= st && ks->ownQ[k] < ed) { if (!gap_len[d]) gap0[d] = k; ++gap_len[d]; own_digit[k] = d; }
    if (!gap_len[d]) gap0[d] = 0;
  }
  // (a single-prime digit's sign-aware copy also fills the digit's own limb of its block; like the others it is neither transformed nor read)
  if (rh_can_ntt_digits(RQ)) { if (int rc = rh_std_ntt_fwd_blocks(RQ, c2Q, wq, npoly, beta, nQ, gap0.data(), gap_len.data(), true)) return rc; }
  else if (int rc = rh_std_ntt_launch(RQ, c2Q, c2Q, beta * npoly, nQ, 0, false, false, 0)) return rc;
  if (nP) {
    if (rh_can_ntt_digits(RP)) { if (int rc = rh_std_ntt_fwd_blocks(RP, c2P, wp, npoly, beta, nP, zero.data(), zero.data(), true)) return rc; }
    else if (int rc = rh_std_ntt_launch(RP, c2P, c2P, beta * npoly, nP, 0, false, false, 0)) return rc;
  }
  if (int rc = rh_gadget_mac_all(RQ, c2Q, wq, evkQ_loc, beta, ks->QiOverF, ct0_loc, ct1_loc, npoly, nQ, cx_loc, own_digit.data(), 1)) return rc;
  if (nP) if (int rc = rh_gadget_mac_all(RP, c2P, wp, evkP_loc, beta, ks->PiOverF, accP0_loc, accP1_loc, npoly, nP, nullptr, nullptr, 1)) return rc;
  return RH_OK;
}

// ModDownQPtoQNTT (ring/basis_extension.go:241-258) for the owned Q limbs.  srcP: (npoly, levelP+1, N) = INTTLazy of the
// P part gathered from its owners (each owner runs rh_ring_intt(lazy) on its limbs first); ctQ_in/out: owned Q limbs.
extern "C" int rh_kshard_moddown(rh_kshard* ks, const uint64_t* srcP, const uint64_t* ctQ_in, uint64_t* ctQ_out, int npoly) {
  if (!ks || !srcP || !ctQ_in || !ctQ_out) return rh_fail(RH_ERR_ARG, "rh_kshard_moddown: null argument");
  if (npoly <= 0) return RH_OK;
  rh_ring* RQ = ks->Q;
  (void)hipSetDevice(RQ->device);
  std::lock_guard<std::recursive_mutex> lk(ks->mu);
  RhCallScope scope(rh_stream(RQ));
  const int N = RQ->N, nQ = RQ->L, LP = ks->levelP + 1;
  if (!ks->have_md) {
    std::vector<u64> tg(RQ->moduli.begin(), RQ->moduli.end()), qsi, coef, vt;
    rh_gen_modup(ks->allP, tg, qsi, coef, vt);
    std::vector<BextSource> S(LP); std::vector<BextTarget> T(nQ);
    for (int i = 0; i < LP; ++i) S[i] = BextSource{ks->allP[i], rh::gen_mred_constant(ks->allP[i]), qsi[i], rh_half_product_mod(ks->allP, ks->allP[i])};
    for (int k = 0; k < nQ; ++k) {
      BextTarget t{}; t.p = tg[k]; t.pinv = RQ->mred[k]; t.half = rh_half_product_mod(ks->allP, tg[k]);
      t.buf = 0; t.limb = k; t.post = 1; t.skip = 0;
      T[k] = t;
    }
    if (int rc = rh_bext_upload_plan(ks->md_plan, S, T, coef, vt)) return rc;
    ks->have_md = true;
  }
  u64* buffQ;
  if (int rc = ks_buf(ks, 2, (size_t)npoly * nQ * N, &buffQ)) return rc;
  if (int rc = rh_bext_launch_raw(rh_stream(RQ), N, ks->md_plan, srcP, LP, 0, buffQ, nQ, nullptr, 0, nullptr, 0, npoly, BEXT_ADD_CRED)) return rc;
  if (rh_can_fuse_submul(RQ)) return rh_std_ntt_submul_launch(RQ, buffQ, npoly, nQ, 0, ctQ_in, nQ, ctQ_out, nQ, ks->md_scalars.data());
  if (int rc = rh_std_ntt_launch(RQ, buffQ, buffQ, npoly, nQ, 0, false, false, 0)) return rc;
  return rh_vec_launch(RQ, RH_OP_SUB_THEN_MUL_SCALAR_MONT_TWO_MODULUS, buffQ, ctQ_in, ctQ_out, npoly, nQ, 0, ks->md_scalars.data(), nullptr);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The whole limb-sharded product behind the C ABI: rlwe.Evaluator.GadgetProduct (core/rlwe/evaluator_gadget_product.go:16-30) =
// ringQ.INTT(cx) (:138) -> [exchange: every limb of INTT(cx)] -> gadgetProductMultiplePLazy on the owned limbs (:122-188, DecomposeSingleNTT
// :455-478) -> INTTLazy of the owned P limbs -> [exchange: the P part of both accumulators] -> ModDownQPtoQNTT on the owned Q limbs (:33-46).
// The library still never communicates: the two exchanges are calls of the host's all-gather (RCCL ncclAllGather from a cgo host,
// torch.distributed from Python), stream-ordered on the stream the library passes.  The batch is cut into chunks that alternate
// between two side streams, so chunk k+1's exchange runs under chunk k's arithmetic (collectives of one communicator execute in
// issue order; what overlaps is an exchange with the OTHER chunk's kernels).
// ---------------------------------------------------------------------------------------------------------------------------------
struct GatherMap { unsigned char owner[RH_MAX_LIMBS], slot[RH_MAX_LIMBS]; };
// recv: (world, npoly, cmax, N) as an all-gather of (npoly, cmax, N) blocks leaves it; out: (npoly, nl, N) in chain order
__global__ void __launch_bounds__(256)
kshard_unpack_kernel(const u64* __restrict__ recv, u64* __restrict__ out, int npoly, int nl, int cmax, int N, GatherMap m) {
  const int row = blockIdx.x, poly = row / nl, i = row - poly * nl;
  const u64* src = recv + (((size_t)m.owner[i] * npoly + poly) * cmax + m.slot[i]) * (size_t)N;
  u64* dst = out + (size_t)row * N;
  for (int j = (blockIdx.y * 256 + threadIdx.x) * 2; j < N; j += gridDim.y * 512)
    *reinterpret_cast<ulonglong2*>(dst + j) = *reinterpret_cast<const ulonglong2*>(src + j);
}

extern "C" int rh_kshard_set_world(rh_kshard* ks, int world, int rank, const int* owner) {
  if (!ks || !owner || world < 1 || world > 255 || rank < 0 || rank >= world) return rh_fail(RH_ERR_ARG, "rh_kshard_set_world: bad argument");
  const int LQ = ks->levelQ + 1, LP = ks->levelP + 1;
  std::vector<int> oq, op;
  for (int i = 0; i < LQ + LP; ++i) {
    if (owner[i] < 0 || owner[i] >= world) return rh_fail(RH_ERR_ARG, "rh_kshard_set_world: owner[%d] = %d outside [0,%d)", i, owner[i], world);
    if (owner[i] == rank) (i < LQ ? oq : op).push_back(i < LQ ? i : i - LQ);
  }
  if (oq != ks->ownQ || op != ks->ownP) return rh_fail(RH_ERR_ARG, "rh_kshard_set_world: rank %d's limbs in the owner map are not the owned limbs the handle was created with", rank);
  std::lock_guard<std::recursive_mutex> lk(ks->mu);
  ks->world = world; ks->rank = rank; ks->owner.assign(owner, owner + LQ + LP);
  return RH_OK;
}

// exchange geometry of one chain (Q: base 0, P: base levelQ+1): padded limb count per rank and the unpack map
static int chain_map(const rh_kshard* ks, int base, int nl, GatherMap* m) {
  std::vector<int> cnt(ks->world, 0);
  for (int i = 0; i < nl; ++i) { const int r = ks->owner[base + i]; m->owner[i] = (unsigned char)r; m->slot[i] = (unsigned char)cnt[r]++; }
  int cmax = 1;
  for (int r = 0; r < ks->world; ++r) if (cnt[r] > cmax) cmax = cnt[r];
  return cmax;
}
static size_t slot_exchange_words(const rh_kshard* ks, int pc) {                    // send + recv of the larger of the two exchanges of a chunk
  if (ks->world == 1) return 0;
  GatherMap m;
  const size_t N = (size_t)ks->Q->N;
  const size_t sq = (size_t)pc * chain_map(ks, 0, ks->levelQ + 1, &m) * N, sp = (size_t)2 * pc * chain_map(ks, ks->levelQ + 1, ks->levelP + 1, &m) * N;
  return (sq > sp ? sq : sp) * (size_t)(1 + ks->world);
}
static int pick_chunks(const rh_kshard* ks, int npoly, int chunks) {
  if (chunks <= 0) chunks = (ks->world > 1 && npoly >= 4) ? 4 : 1;                 // auto: four chunks hide three of four exchanges
  return chunks > npoly ? npoly : chunks;
}
extern "C" int rh_kshard_exchange_words(const rh_kshard* ks, int npoly, int chunks, size_t* words) {
  if (!ks || !words || npoly < 0) return rh_fail(RH_ERR_ARG, "rh_kshard_exchange_words: bad argument");
  if (ks->owner.empty()) return rh_fail(RH_ERR_ARG, "rh_kshard_exchange_words: call rh_kshard_set_world first");
  const int nc = npoly ? pick_chunks(ks, npoly, chunks) : 1, pc = npoly ? (npoly + nc - 1) / nc : 0;
  *words = slot_exchange_words(ks, pc) * (size_t)(nc > 1 ? 2 : 1);
  return RH_OK;
}
extern "C" int rh_kshard_set_exchange(rh_kshard* ks, uint64_t* arena_dev, size_t words) {
  if (!ks) return rh_fail(RH_ERR_ARG, "rh_kshard_set_exchange: null handle");
  std::lock_guard<std::recursive_mutex> lk(ks->mu);
  ks->arena = arena_dev; ks->arena_words = arena_dev ? words : 0;
  return RH_OK;
}

// one exchange: the owned limbs of `loc` (npoly, nown, N) -> every limb of the chain in chain order, `out` (npoly, nl, N)
static int exchange_chain(rh_kshard* ks, hipStream_t st, const u64* loc, int nown, int npoly, int base, int nl, u64* xsend, u64* xrecv,
                          u64* out, rh_allgather_fn ag, void* ctx) {
  GatherMap m;
  const int cmax = chain_map(ks, base, nl, &m), N = ks->Q->N;
  const u64* send = loc;
  if (nown != cmax || ks->arena) {                                                 // pad this rank's block to the largest per-rank limb count (with a
                                                                                   // registered arena always: the host maps BOTH pointers back to it)
    if (nown && hipMemcpy2DAsync(xsend, (size_t)cmax * N * 8, loc, (size_t)nown * N * 8, (size_t)nown * N * 8, (size_t)npoly, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return rh_fail(RH_ERR_DEVICE, "rh_kshard_gadget_product: packing the exchange block failed");
    send = xsend;
  }
  if (int rc = ag(ctx, send, xrecv, (size_t)npoly * cmax * N, (void*)st)) return rh_fail(RH_ERR_DEVICE, "rh_kshard_gadget_product: the caller's all-gather returned %d", rc);
  unsigned cy = ((unsigned)N + 2047) / 2048; if (cy > 16) cy = 16;
  kshard_unpack_kernel<<<dim3((unsigned)npoly * nl, cy), 256, 0, st>>>(xrecv, out, npoly, nl, cmax, N, m);
  if (hipGetLastError() != hipSuccess) return rh_fail(RH_ERR_DEVICE, "kshard_unpack_kernel launch failed");
  return RH_OK;
}

static int product_chunk(rh_kshard* ks, hipStream_t st, const u64* cx, const u64* evkQ, const u64* evkP, u64* ct0, u64* ct1, int pc, u64* xsend,
                         u64* xrecv, rh_allgather_fn ag, void* ctx) {
  rh_ring* RQ = ks->Q; rh_ring* RP = ks->P;
  const int N = RQ->N, nQ = RQ->L, nP = RP ? RP->L : 0, LQ = ks->levelQ + 1, LP = ks->levelP + 1;
  RhCallScope scope(st, nullptr, 0);
  u64 *cxinv, *acc, *src;
  if (int rc = ks_buf(ks, 5, (size_t)pc * nQ * N, &cxinv)) return rc;
  if (int rc = ks_buf(ks, 6, (size_t)2 * pc * (nP ? nP : 1) * N, &acc)) return rc;
  if (int rc = rh_std_ntt_launch(RQ, cx, cxinv, pc, nQ, 0, true, false, 0)) return rc;                    // ringQ.INTT(cx, cxInvNTT) (:138)
  const u64* srcQ = cxinv;
  if (ks->world > 1) {
    if (int rc = ks_buf(ks, 7, (size_t)pc * (LQ > 2 * LP ? LQ : 2 * LP) * N, &src)) return rc;
    if (int rc = exchange_chain(ks, st, cxinv, nQ, pc, 0, LQ, xsend, xrecv, src, ag, ctx)) return rc;
    srcQ = src;
  }
  u64* a0 = acc; u64* a1 = acc + (size_t)pc * nP * N;
  if (int rc = rh_kshard_product(ks, srcQ, cx, evkQ, evkP, ct0, ct1, nP ? a0 : nullptr, nP ? a1 : nullptr, pc)) return rc;
  if (nP) if (int rc = rh_std_ntt_launch(RP, acc, acc, 2 * pc, nP, 0, true, true, 0)) return rc;           // INTTLazy of the owned P limbs, both components (:241-246)
  const u64* srcP = acc;
  if (ks->world > 1) {
    if (int rc = exchange_chain(ks, st, acc, nP, 2 * pc, LQ, LP, xsend, xrecv, src, ag, ctx)) return rc;
    srcP = src;
  }
  if (int rc = rh_kshard_moddown(ks, srcP, ct0, ct0, pc)) return rc;
  return rh_kshard_moddown(ks, srcP + (size_t)pc * LP * N, ct1, ct1, pc);
}

extern "C" int rh_kshard_gadget_product(rh_kshard* ks, const uint64_t* cx_loc, const uint64_t* evkQ_loc, const uint64_t* evkP_loc,
                                        uint64_t* ct0_loc, uint64_t* ct1_loc, int npoly, rh_allgather_fn allgather, void* ctx, int chunks) {
  if (!ks || !cx_loc || !evkQ_loc || !ct0_loc || !ct1_loc) return rh_fail(RH_ERR_ARG, "rh_kshard_gadget_product: null argument");
  if (ks->P && !evkP_loc) return rh_fail(RH_ERR_ARG, "rh_kshard_gadget_product: null P-part key");
  if (npoly <= 0) return RH_OK;
  rh_ring* RQ = ks->Q;
  (void)hipSetDevice(RQ->device);
  std::lock_guard<std::recursive_mutex> lk(ks->mu);
  if (ks->owner.empty()) {                                  // no map given: legal only when this rank owns everything
    if ((int)ks->ownQ.size() != ks->levelQ + 1 || (int)ks->ownP.size() != ks->levelP + 1)
      return rh_fail(RH_ERR_ARG, "rh_kshard_gadget_product: call rh_kshard_set_world first (this handle owns a subset of the limbs)");
    ks->world = 1; ks->rank = 0; ks->owner.assign(ks->levelQ + ks->levelP + 2, 0);
  }
  if (ks->world > 1 && !allgather) return rh_fail(RH_ERR_ARG, "rh_kshard_gadget_product: %d ranks need the caller's all-gather", ks->world);
  const int N = RQ->N, nQ = RQ->L, nc = pick_chunks(ks, npoly, chunks), pc = (npoly + nc - 1) / nc;
  const size_t xw = slot_exchange_words(ks, pc);
  hipStream_t main = rh_stream(RQ);
  if (nc > 1 && !ks->side[0]) {
    bool ok = hipEventCreateWithFlags(&ks->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < 2 && ok; ++k)
      ok = hipStreamCreateWithFlags(&ks->side[k], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&ks->ev_join[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) return rh_fail(RH_ERR_DEVICE, "rh_kshard_gadget_product: stream / event creation failed");
  }
  // exchange memory per slot: the caller's arena when it is large enough (a host that must map the pointers back to its own buffers), else ours
  u64 *xs[2] = {nullptr, nullptr}, *xr[2] = {nullptr, nullptr};
  const int nslots = nc > 1 ? 2 : 1;
  for (int sl = 0; sl < nslots && xw; ++sl) {
    const size_t sw = xw / (size_t)(1 + ks->world);
    if (ks->arena && ks->arena_words >= xw * nslots) { xs[sl] = ks->arena + (size_t)sl * xw; xr[sl] = xs[sl] + sw; }
    else {
      ks->slot = sl;
      if (int rc = ks_buf(ks, 8, sw, &xs[sl])) return rc;
      if (int rc = ks_buf(ks, 9, xw - sw, &xr[sl])) return rc;
    }
  }
  int rc = RH_OK;
  if (nc == 1) { ks->slot = 0; rc = product_chunk(ks, main, cx_loc, evkQ_loc, evkP_loc, ct0_loc, ct1_loc, npoly, xs[0], xr[0], allgather, ctx); return rc; }
  if (hipEventRecord(ks->ev_fork, main) != hipSuccess) return rh_fail(RH_ERR_DEVICE, "rh_kshard_gadget_product: fork failed");
  for (int k = 0; k < 2; ++k) (void)hipStreamWaitEvent(ks->side[k], ks->ev_fork, 0);
  for (int k = 0; k < nc && !rc; ++k) {
    const int p0 = k * pc, n = npoly - p0 < pc ? npoly - p0 : pc;
    if (n <= 0) break;
    const size_t off = (size_t)p0 * nQ * N;
    ks->slot = k & 1;
    rc = product_chunk(ks, ks->side[k & 1], cx_loc + off, evkQ_loc, evkP_loc, ct0_loc + off, ct1_loc + off, n, xs[k & 1], xr[k & 1], allgather, ctx);
  }
  ks->slot = 0;
  for (int k = 0; k < 2; ++k) {                             // join even after an error: the caller's stream must not run ahead of the side streams
    (void)hipEventRecord(ks->ev_join[k], ks->side[k]);
    (void)hipStreamWaitEvent(main, ks->ev_join[k], 0);
  }
  return rc;
}
