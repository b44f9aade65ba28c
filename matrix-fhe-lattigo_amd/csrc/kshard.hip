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
  u64* buf[5] = {}; size_t buf_words[5] = {};           // c2Q, c2P, buffQ; 3, 4: all digits' c2Q / c2P (rh_kshard_product)
  int reduce = 0, QiOverF = 1, PiOverF = 1;
  std::recursive_mutex mu;                               // one key switch at a time per handle (digits are fed in order)
};

static int ks_buf(rh_kshard* ks, int which, size_t words, u64** out) {
  if (ks->buf_words[which] < words) {
    if (ks->buf[which]) (void)hipFree(ks->buf[which]);
    ks->buf[which] = nullptr; ks->buf_words[which] = 0;
    if (hipMalloc((void**)&ks->buf[which], (words ? words : 1) * 8) != hipSuccess) return rh_fail(RH_ERR_NOMEM, "hipMalloc(key-switch shard scratch) failed");
    ks->buf_words[which] = words;
  }
  *out = ks->buf[which];
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
  for (int i = 0; i < 5; ++i) if (ks->buf[i]) (void)hipFree(ks->buf[i]);
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
