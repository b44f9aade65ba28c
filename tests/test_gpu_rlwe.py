"""GPU: the key-switch callers of core/rlwe on device batches -- hoisted decomposition / gadget product and the
automorphism (rotation) of a degree-1 ciphertext -- against the oracle composition of tests/test_gpu_keyswitch.py.
Evaluation keys are uniformly random: arithmetic parity needs no key generation; the end-to-end noise
statistics of a real key switch stay "parity unpinned" (core/rlwe/rlwe_test.go:690-798)."""
import numpy as np
import pytest

from conftest import QI60, PI60, uniform_mod
from test_gpu_keyswitch import oracle_gadget_product

pytestmark = pytest.mark.gpu


def make_case(rh, N, nq, np_, npoly, seed):
    Q, P = QI60[:nq], PI60[:np_]
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    rng = np.random.default_rng(seed)
    beta = (nq - 1 + np_) // np_
    evkQ = np.stack([np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(2)]) for _ in range(beta)])
    evkP = np.stack([np.stack([np.stack([uniform_mod(rng, p, N) for p in P]) for _ in range(2)]) for _ in range(beta)])
    c0 = np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(npoly)])
    c1 = np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(npoly)])
    return Q, P, rq, rp, beta, evkQ, evkP, c0, c1


# N >= 2^14: all digit blocks transformed by one pipelined stream of launches (rh_std_ntt_fwd_digits), (7, 3): digits of 3, 3, 1 limbs
@pytest.mark.parametrize("N,nq,np_", [(64, 6, 2), (4096, 7, 3), (8192, 5, 2), (16384, 7, 3), (16384, 6, 2), (32768, 5, 2)])
def test_hoisted_equals_direct_and_oracle(rh, oracle, N, nq, np_):
    Q, P, rq, rp, beta, evkQ, evkP, _c0, cx = make_case(rh, N, nq, np_, 2, N + nq)
    ev = rh.rlwe.Evaluator(rq, rp)
    gct = rh.rlwe.GadgetCiphertext(rq, rp, evkQ, evkP)
    levelQ, levelP = nq - 1, np_ - 1
    pcx = rh.DevicePoly.from_numpy(rq, cx)
    direct = rh.Ciphertext([rq.NewPoly(2), rq.NewPoly(2)], is_ntt=True)
    ev.GadgetProduct(levelQ, pcx, gct, direct)
    d0, d1 = direct.Value[0].numpy(), direct.Value[1].numpy()
    for k in range(2):
        e0, e1 = oracle_gadget_product(oracle, rh, N, Q, P, levelQ, levelP, cx[k], evkQ, evkP)
        assert np.array_equal(d0[k], e0) and np.array_equal(d1[k], e1)
    # hoisted: decomposition from the NTT-domain input, and from its coefficient-domain form
    pinv = rq.NewPoly(2); rq.INTT(pcx, pinv)
    for src, is_ntt in ((pcx, True), (pinv, False)):
        dq, dp = ev.DecomposeNTT(levelQ, levelP, src, is_ntt)
        assert dq.npoly == beta * 2 and dp.npoly == beta * 2
        h = rh.Ciphertext([rq.NewPoly(2), rq.NewPoly(2)], is_ntt=True)
        ev.GadgetProductHoisted(levelQ, (dq, dp), gct, h)
        assert np.array_equal(h.Value[0].numpy(), d0) and np.array_equal(h.Value[1].numpy(), d1)
    # the decomposition itself: digit 1 of poly 1 against the oracle (DecomposeSingleNTT :455-478)
    srQ = [oracle.SubRingConsts(N, q) for q in Q]; srP = [oracle.SubRingConsts(N, p) for p in P]
    cxinv = np.stack([oracle.intt(cx[1, i], srQ[i]) for i in range(nq)])
    LP = np_
    c2q, c2p = oracle.decompose_and_split(levelQ, levelP, LP, 1, cxinv, Q, P)
    st, ed = LP, min(2 * LP, nq)
    expq = np.stack([cx[1, i] if st <= i < ed else oracle.ntt(c2q[i], srQ[i]) for i in range(nq)])
    expp = np.stack([oracle.ntt(c2p[j], srP[j]) for j in range(np_)])
    assert np.array_equal(dq.numpy()[1 * 2 + 1], expq) and np.array_equal(dp.numpy()[1 * 2 + 1], expp)
    ev.close(); rq.close(); rp.close()


@pytest.mark.parametrize("N,nq,np_,gal", [(64, 6, 2, 5), (4096, 5, 2, 3), (2048, 4, 2, 2 * 2048 - 1)])
def test_automorphism_keyswitch_vs_oracle(rh, oracle, N, nq, np_, gal):
    Q, P, rq, rp, beta, evkQ, evkP, c0, c1 = make_case(rh, N, nq, np_, 2, N + gal)
    gct = rh.rlwe.GadgetCiphertext(rq, rp, evkQ, evkP)
    ev = rh.rlwe.Evaluator(rq, rp, galois_keys={gal: gct})
    levelQ, levelP = nq - 1, np_ - 1
    ct = rh.Ciphertext([rh.DevicePoly.from_numpy(rq, c0), rh.DevicePoly.from_numpy(rq, c1)], is_ntt=True)
    out = rh.Ciphertext([rq.NewPoly(2), rq.NewPoly(2)], is_ntt=True)
    ev.Automorphism(ct, gal, out)
    g0, g1 = out.Value[0].numpy(), out.Value[1].numpy()
    OPS = rh.OPS
    for k in range(2):
        e0, e1 = oracle_gadget_product(oracle, rh, N, Q, P, levelQ, levelP, c1[k], evkQ, evkP)
        for i, q in enumerate(Q):
            s = oracle.vec_op(OPS["ADD"], e0[i], c0[k, i], e0[i], 0, 0, q)
            assert np.array_equal(g0[k, i], oracle.automorphism_ntt(s, gal))
            assert np.array_equal(g1[k, i], oracle.automorphism_ntt(e1[i], gal))
    # hoisted form gives the same bits; inputs untouched
    dq_dp = ev.DecomposeNTT(levelQ, levelP, ct.Value[1], True)
    out2 = rh.Ciphertext([rq.NewPoly(2), rq.NewPoly(2)], is_ntt=True)
    ev.AutomorphismHoisted(levelQ, ct, dq_dp, gal, out2)
    assert np.array_equal(out2.Value[0].numpy(), g0) and np.array_equal(out2.Value[1].numpy(), g1)
    assert np.array_equal(ct.Value[0].numpy(), c0) and np.array_equal(ct.Value[1].numpy(), c1)
    # galEl == 1 is a copy (:20-25); a missing key is an error (:27-30); degree must be 1 (:16-18)
    out3 = rh.Ciphertext([rq.NewPoly(2), rq.NewPoly(2)], is_ntt=True)
    ev.Automorphism(ct, 1, out3)
    assert np.array_equal(out3.Value[0].numpy(), c0) and np.array_equal(out3.Value[1].numpy(), c1)
    with pytest.raises(rh.RingHipError):
        ev.Automorphism(ct, gal + 2, out3)
    with pytest.raises(rh.RingHipError):
        ev.Automorphism(rh.Ciphertext([ct.Value[0]], is_ntt=True), gal, out3)
    ev.close(); rq.close(); rp.close()


def test_relinearize_and_apply_evaluation_key(rh, oracle):
    # Relinearize (evaluator_evaluationkey.go:125-153) and ApplyEvaluationKey, same ring degree (:97-112)
    N, nq, np_ = 4096, 5, 2
    Q, P, rq, rp, beta, evkQ, evkP, c0, c1 = make_case(rh, N, nq, np_, 2, 77)
    rng = np.random.default_rng(5)
    c2 = np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(2)])
    gct = rh.rlwe.GadgetCiphertext(rq, rp, evkQ, evkP)
    ev = rh.rlwe.Evaluator(rq, rp, galois_keys={"rlk": gct})
    dp = lambda a: rh.DevicePoly.from_numpy(rq, a)
    ct2 = rh.Ciphertext([dp(c0), dp(c1), dp(c2)], is_ntt=True)
    out = rh.Ciphertext([rq.NewPoly(2), rq.NewPoly(2)], is_ntt=True)
    ev.Relinearize(ct2, out)
    ct1 = rh.Ciphertext([dp(c0), dp(c2)], is_ntt=True)
    out2 = rh.Ciphertext([rq.NewPoly(2), rq.NewPoly(2)], is_ntt=True)
    ev.ApplyEvaluationKey(ct1, gct, out2)
    OPS = rh.OPS
    for k in range(2):
        e0, e1 = oracle_gadget_product(oracle, rh, N, Q, P, nq - 1, np_ - 1, c2[k], evkQ, evkP)
        for i, q in enumerate(Q):
            assert np.array_equal(out.Value[0].numpy()[k, i], oracle.vec_op(OPS["ADD"], c0[k, i], e0[i], e0[i], 0, 0, q))
            assert np.array_equal(out.Value[1].numpy()[k, i], oracle.vec_op(OPS["ADD"], c1[k, i], e1[i], e1[i], 0, 0, q))
            assert np.array_equal(out2.Value[0].numpy()[k, i], oracle.vec_op(OPS["ADD"], c0[k, i], e0[i], e0[i], 0, 0, q))
            assert np.array_equal(out2.Value[1].numpy()[k, i], e1[i])
    with pytest.raises(rh.RingHipError):
        ev.Relinearize(ct1, out)                       # degree must be 2
    with pytest.raises(rh.RingHipError):
        rh.rlwe.Evaluator(rq, rp).Relinearize(ct2, out)   # key missing
    ev.close(); rq.close(); rp.close()


@pytest.mark.parametrize("N,nq,np_", [(4096, 5, 2), (16384, 6, 3)])
def test_lazy_hoisted_product_and_rotation_then_moddown(rh, oracle, N, nq, np_):
    # GadgetProductHoistedLazy (:351-371) + Evaluator.ModDown (:33-46) == GadgetProductHoisted, bit for bit (same accumulators, same ModDown);
    # AutomorphismHoistedLazy (:103-160) + ModDown == AutomorphismHoisted up to the centred rounding of the division by P: the automorphism
    # negates coefficients, and the rounding of (x - [x]_P) / P is odd-symmetric except on its boundary, so the two agree exactly on random data.
    Q, P, rq, rp, beta, evkQ, evkP, c0, c1 = make_case(rh, N, nq, np_, 2, 3 * N + nq)
    gal = 5
    gct = rh.rlwe.GadgetCiphertext(rq, rp, evkQ, evkP)
    ev = rh.rlwe.Evaluator(rq, rp, galois_keys={gal: gct})
    levelQ, levelP = nq - 1, np_ - 1
    ct = rh.Ciphertext([rh.DevicePoly.from_numpy(rq, c0), rh.DevicePoly.from_numpy(rq, c1)], is_ntt=True)
    dec = ev.ALlocateDecompositionBuffer(levelQ, levelP, 2)
    ev.DecomposeNTT(levelQ, levelP, ct.Value[1], True, dec)
    ref = rh.Ciphertext([rq.NewPoly(2), rq.NewPoly(2)], is_ntt=True)
    ev.GadgetProductHoisted(levelQ, dec, gct, ref)
    lazy = rh.rlwe.ElementQP.alloc(rq, rp, 2, levelQ, levelP)
    ev.GadgetProductHoistedLazy(levelQ, dec, gct, lazy)
    assert lazy.IsNTT and lazy.LevelP() == levelP
    # the accumulators themselves: canonical residues of sum_d evk_d * dec_d * 2^-64 (one limb of Q and one of P against plain integers)
    dq, dp = dec[0].numpy(), dec[1].numpy()
    for which, acc, key, d, mods, i in (("Q", lazy.Value[1].Q, evkQ, dq, Q, nq - 1), ("P", lazy.Value[1].P, evkP, dp, P, 0)):
        q = int(mods[i]); rinv = pow(1 << 64, -1, q)
        want = sum(key[dd, 1, i].astype(object) * d[dd * 2 + 1, i].astype(object) for dd in range(beta)) * rinv % q
        assert [int(x) for x in acc.numpy()[1, i][:64]] == [int(x) for x in want[:64]], which
    out = rh.Ciphertext([rq.NewPoly(2), rq.NewPoly(2)], is_ntt=True)
    ev.ModDown(levelQ, levelP, lazy, out)
    assert np.array_equal(out.Value[0].numpy(), ref.Value[0].numpy()) and np.array_equal(out.Value[1].numpy(), ref.Value[1].numpy())
    rot = rh.Ciphertext([rq.NewPoly(2), rq.NewPoly(2)], is_ntt=True)
    ev.AutomorphismHoisted(levelQ, ct, dec, gal, rot)
    lz = rh.rlwe.ElementQP.alloc(rq, rp, 2, levelQ, levelP)
    ev.AutomorphismHoistedLazy(levelQ, ct, dec, gal, lz)
    got = rh.Ciphertext([rq.NewPoly(2), rq.NewPoly(2)], is_ntt=True)
    ev.ModDown(levelQ, levelP, lz, got)
    assert np.array_equal(got.Value[1].numpy(), rot.Value[1].numpy())
    assert np.array_equal(got.Value[0].numpy(), rot.Value[0].numpy())
    ev.close(); rq.close(); rp.close()


def test_gadget_product_lazy_and_decompose_single(rh, oracle):
    # GadgetProductLazy (:100-120) == GadgetProductHoistedLazy on DecomposeNTT's output; DecomposeSingleNTT (:455-478) == that digit of DecomposeNTT
    N, nq, np_ = 4096, 6, 2
    Q, P, rq, rp, beta, evkQ, evkP, c0, c1 = make_case(rh, N, nq, np_, 2, 77)
    gct = rh.rlwe.GadgetCiphertext(rq, rp, evkQ, evkP)
    ev = rh.rlwe.Evaluator(rq, rp)
    levelQ, levelP = nq - 1, np_ - 1
    pcx = rh.DevicePoly.from_numpy(rq, c1)
    dec = ev.DecomposeNTT(levelQ, levelP, pcx, True)
    a, b = rh.rlwe.ElementQP.alloc(rq, rp, 2, levelQ, levelP), rh.rlwe.ElementQP.alloc(rq, rp, 2, levelQ, levelP)
    ev.GadgetProductHoistedLazy(levelQ, dec, gct, a)
    ev.GadgetProductLazy(levelQ, pcx, gct, b)
    for c in (0, 1):
        assert np.array_equal(a.Value[c].Q.numpy(), b.Value[c].Q.numpy()) and np.array_equal(a.Value[c].P.numpy(), b.Value[c].P.numpy())
    inv = rq.NewPoly(2); rq.INTT(pcx, inv)
    dq, dp = dec[0].numpy(), dec[1].numpy()
    for d in range(beta):
        oq, op = rq.NewPoly(2), rp.NewPoly(2)
        ev.DecomposeSingleNTT(levelQ, levelP, np_, d, pcx, inv, oq, op)
        assert np.array_equal(oq.numpy(), dq[d * 2:(d + 1) * 2]) and np.array_equal(op.numpy(), dp[d * 2:(d + 1) * 2])
    ev.close(); rq.close(); rp.close()


def test_lazy_forms_refuse_batches_with_more_p_limbs_and_keep_the_flag(rh, oracle):
    # the C entries stride the P accumulators by levelP+1 rows per poly, so a BATCH whose P part carries more limbs than
    # gadgetCt.LevelP()+1 (allowed by the reference: ctQP.LevelP() >= levelP) must be refused, not strided wrongly; a single poly is fine.
    N, nq = 4096, 5
    Q, P3 = QI60[:nq], PI60[:3]
    Q_, P, rq, rp, beta, evkQ, evkP, c0, c1 = make_case(rh, N, nq, 2, 2, 991)
    rp3 = rh.Ring(N, P3)                                         # a P ring with one more limb than the key's
    gal = 5
    gct = rh.rlwe.GadgetCiphertext(rq, rp, evkQ, evkP)
    ev = rh.rlwe.Evaluator(rq, rp, galois_keys={gal: gct})
    levelQ, levelP = nq - 1, 1
    ct = rh.Ciphertext([rh.DevicePoly.from_numpy(rq, c0), rh.DevicePoly.from_numpy(rq, c1)], is_ntt=True)
    dec = ev.DecomposeNTT(levelQ, levelP, ct.Value[1], True)
    wide = rh.rlwe.ElementQP.alloc(rq, rp3, 2, levelQ, 2)          # batch of 2, P blocks of 3 limbs: LevelP() = 2 > levelP = 1
    assert wide.LevelP() == 2
    with pytest.raises(rh.RingHipError):
        ev.GadgetProductHoistedLazy(levelQ, dec, gct, wide)
    out = rh.Ciphertext([rq.NewPoly(2), rq.NewPoly(2)], is_ntt=True)
    with pytest.raises(rh.RingHipError):
        ev.ModDown(levelQ, levelP, wide, out)
    # one poly with more P limbs: leading limbs are contiguous, accepted and equal to the exact-size call
    one = rh.Ciphertext([rh.DevicePoly.from_numpy(rq, c0[:1]), rh.DevicePoly.from_numpy(rq, c1[:1])], is_ntt=True)
    dec1 = ev.DecomposeNTT(levelQ, levelP, one.Value[1], True)
    w1 = rh.rlwe.ElementQP.alloc(rq, rp3, 1, levelQ, 2)
    e1 = rh.rlwe.ElementQP.alloc(rq, rp, 1, levelQ, levelP)
    ev.GadgetProductHoistedLazy(levelQ, dec1, gct, w1)
    ev.GadgetProductHoistedLazy(levelQ, dec1, gct, e1)
    for c in (0, 1):
        assert np.array_equal(w1.Value[c].Q.numpy(), e1.Value[c].Q.numpy())
        assert np.array_equal(w1.Value[c].P.numpy()[:, :2], e1.Value[c].P.numpy())
    # AutomorphismHoistedLazy with ctQP.IsNTT == False (evaluator_automorphism.go:146-157): the coefficient-domain index map of
    # ring.Automorphism is applied to the accumulators, and the caller's flag is left as it was
    lzn = rh.rlwe.ElementQP.alloc(rq, rp, 2, levelQ, levelP); lzn.IsNTT = False
    ev.AutomorphismHoistedLazy(levelQ, ct, dec, gal, lzn)
    assert lzn.IsNTT is False
    acc = rh.rlwe.ElementQP.alloc(rq, rp, 2, levelQ, levelP)
    ev.GadgetProductHoistedLazy(levelQ, dec, gct, acc)
    Pbig = int(P[0]) * int(P[1])
    for c in (0, 1):
        aq, ap = acc.Value[c].Q.numpy(), acc.Value[c].P.numpy()
        for k in range(2):
            for i, q in enumerate(Q):
                x = aq[k, i]
                if c == 0:
                    x = ((x.astype(object) + c0[k, i].astype(object) * (Pbig % int(q))) % int(q)).astype(np.uint64)
                assert np.array_equal(lzn.Value[c].Q.numpy()[k, i], oracle.automorphism(x, gal, int(q))), (c, k, i)
            for j, p in enumerate(P):
                assert np.array_equal(lzn.Value[c].P.numpy()[k, j], oracle.automorphism(ap[k, j], gal, int(p))), (c, k, j)
    ev.close(); rq.close(); rp.close(); rp3.close()
