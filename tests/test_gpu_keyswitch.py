"""GPU: hybrid key-switch gadget product (core/rlwe/evaluator_gadget_product.go:16-188) against the same sequence composed
from oracle pieces (each pinned elsewhere): INTT -> per digit DecomposeAndSplit, NTT, MulCoeffsMontgomeryLazy(ThenAddLazy)
with periodic Reduce -> ModDownQPtoQNTT.  The evaluation key is uniformly random: arithmetic
parity needs no key generation.  End-to-end key-switch semantics are pinned only statistically in the reference
(core/rlwe/rlwe_test.go:690-798): that part stays "parity unpinned"."""
import numpy as np
import pytest

from conftest import QI60, PI60, uniform_mod

pytestmark = pytest.mark.gpu


def oracle_gadget_product(oracle, rh, N, Q, P, levelQ, levelP, cx, evkQ, evkP):
    """the composition lives in oracle/compose.py (bench.py's verification leg uses it too)"""
    from oracle import compose
    assert compose.OPS == rh.OPS
    return compose.gadget_product(N, Q, P, levelQ, levelP, cx, evkQ, evkP)


@pytest.mark.parametrize("N,nq,np_,levelQ,levelP", [(64, 6, 2, 5, 1), (8192, 8, 3, 7, 2), (4096, 24, 6, 23, 5), (64, 9, 2, 6, 1), (8192, 9, 3, 6, 2), (8192, 8, 3, 7, 1), (4096, 7, 4, 3, 2),
                                                   (16384, 8, 3, 6, 2), (16384, 4, 4, 3, 3), (65536, 5, 2, 4, 1)])
def test_gadget_product_vs_oracle_composition(rh, oracle, N, nq, np_, levelQ, levelP):
    Q, P = QI60[:nq], PI60[:np_]
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    be = rh.BasisExtender(rq, rp)
    rng = np.random.default_rng(N + nq + levelQ)
    LQ, LP = levelQ + 1, levelP + 1
    beta_key = (nq - 1 + np_) // np_ if levelP == np_ - 1 else (levelQ + levelP + 1) // (levelP + 1)
    beta_key = max(beta_key, (levelQ + levelP + 1) // (levelP + 1))
    npoly = 2
    cx = np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(q) for q in Q[:LQ]]) for _ in range(npoly)])
    evkQ = np.stack([np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(q) for q in Q]) for _ in range(2)]) for _ in range(beta_key)])
    evkP = np.stack([np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(p) for p in P]) for _ in range(2)]) for _ in range(beta_key)])
    pcx = rh.DevicePoly.from_numpy(rq.AtLevel(levelQ), cx)
    dq = rh.DevicePoly.from_numpy(rq, evkQ.reshape(beta_key * 2, nq, N))
    dp = rh.DevicePoly.from_numpy(rp, evkP.reshape(beta_key * 2, np_, N))
    ct0, ct1 = rh.DevicePoly(rq, npoly, LQ), rh.DevicePoly(rq, npoly, LQ)
    rq.set_tuning("ks_small_rows", 0)                 # the digit-by-digit extension and the pipelined stream of transforms (large batches)
    be.GadgetProduct(levelQ, levelP, pcx, dq.ptr, dp.ptr, beta_key, ct0, ct1)
    g0, g1 = ct0.numpy(), ct1.numpy()
    # the small-batch path (tuning ks_small_rows, default 512 rows: every digit in ONE extension launch, the transforms of both rings' digit blocks in one launch pair): same bits
    rq.set_tuning("ks_small_rows", 1024)
    s0, s1 = rh.DevicePoly(rq, npoly, LQ), rh.DevicePoly(rq, npoly, LQ)
    for _ in range(2):
        be.GadgetProduct(levelQ, levelP, pcx, dq.ptr, dp.ptr, beta_key, s0, s1)
    assert np.array_equal(s0.numpy(), g0) and np.array_equal(s1.numpy(), g1)
    for k in range(npoly):
        e0, e1 = oracle_gadget_product(oracle, rh, N, Q, P, levelQ, levelP, cx[k], evkQ, evkP)
        assert np.array_equal(g0[k], e0)
        assert np.array_equal(g1[k], e1)
    be.close(); rq.close(); rp.close()


@pytest.mark.parametrize("N,nq,np_,npoly", [(2048, 4, 2, 34), (16384, 6, 2, 32)])
def test_gadget_product_larger_batch_and_aliased_addends(rh, oracle, N, nq, np_, npoly):
    # a batch large enough for the multi-poly key MAC workgroups and the pipelined digit transforms; repeated calls reuse the scratch;
    # the product with addends that alias the outputs
    Q, P = QI60[:nq], PI60[:np_]
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    be = rh.BasisExtender(rq, rp)
    rng = np.random.default_rng(7 * N + npoly)
    levelQ, levelP = nq - 1, np_ - 1
    beta = (levelQ + levelP + 1) // (levelP + 1)
    cx = np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(npoly)])
    add = np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(npoly)])
    evkQ, evkP = _rand_key(rng, beta, Q, N), _rand_key(rng, beta, P, N)
    dq = rh.DevicePoly.from_numpy(rq, evkQ.reshape(beta * 2, nq, N)); dp = rh.DevicePoly.from_numpy(rp, evkP.reshape(beta * 2, np_, N))
    pcx = rh.DevicePoly.from_numpy(rq, cx)
    ct0, ct1 = rh.DevicePoly(rq, npoly, nq), rh.DevicePoly(rq, npoly, nq)
    for _ in range(2):
        be.GadgetProduct(levelQ, levelP, pcx, dq.ptr, dp.ptr, beta, ct0, ct1)
    a0, a1 = rh.DevicePoly.from_numpy(rq, add), rh.DevicePoly.from_numpy(rq, add)
    be.GadgetProductThenAdd(levelQ, levelP, pcx, dq.ptr, dp.ptr, beta, a0, a1, a0, a1)     # outputs alias the addends
    g0, g1, s0, s1 = ct0.numpy(), ct1.numpy(), a0.numpy(), a1.numpy()
    for k in (0, npoly // 2 - 1, npoly // 2, npoly - 1):
        e0, e1 = oracle_gadget_product(oracle, rh, N, Q, P, levelQ, levelP, cx[k], evkQ, evkP)
        assert np.array_equal(g0[k], e0) and np.array_equal(g1[k], e1)
        for i, q in enumerate(Q):
            assert np.array_equal(s0[k, i], (e0[i] + add[k, i]) % np.uint64(q))
            assert np.array_equal(s1[k, i], (e1[i] + add[k, i]) % np.uint64(q))
    be.close(); rq.close(); rp.close()


def _rand_key(rng, rows, mods, N):
    return np.stack([np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(q) for q in mods]) for _ in range(2)]) for _ in range(rows)])


@pytest.mark.parametrize("N,nq,np_,levelQ", [(64, 6, 2, 5), (4096, 5, 2, 4), (8192, 5, 3, 3)])
def test_gadget_product_coefficient_domain_ciphertext(rh, oracle, N, nq, np_, levelQ):
    # ct.IsNTT == false (core/rlwe/evaluator_gadget_product.go:114-118, :139-143; ModDown INTT -> INTT :62-66), levelP >= 1
    from oracle import compose
    Q, P = QI60[:nq], PI60[:np_]
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    rng = np.random.default_rng(N + nq)
    levelP = np_ - 1
    beta = (nq - 1 + np_) // np_
    evkQ, evkP = _rand_key(rng, beta, Q, N), _rand_key(rng, beta, P, N)
    LQ = levelQ + 1
    cx = np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(q) for q in Q[:LQ]]) for _ in range(2)])
    ev = rh.rlwe.Evaluator(rq, rp)
    gct = rh.rlwe.GadgetCiphertext(rq, rp, evkQ, evkP)
    rl = rq.AtLevel(levelQ)
    ct = rh.Ciphertext([rl.NewPoly(2), rl.NewPoly(2)], is_ntt=False)
    ev.GadgetProduct(levelQ, rh.DevicePoly.from_numpy(rl, cx), gct, ct)
    g0, g1 = ct.Value[0].numpy(), ct.Value[1].numpy()
    for k in range(2):
        e0, e1 = compose.gadget_product_coeff(N, Q, P, levelQ, levelP, cx[k], evkQ, evkP)
        assert np.array_equal(g0[k], e0) and np.array_equal(g1[k], e1)
    ev.close(); rq.close(); rp.close()


@pytest.mark.parametrize("N,nq,levelQ,levelP,pw2,is_ntt", [
    (64, 4, 3, 0, 0, True), (4096, 4, 2, 0, 0, True), (4096, 3, 2, 0, 0, False),          # one P modulus, RNS digits only
    (64, 3, 2, 0, 16, True), (8192, 3, 2, 0, 20, True), (4096, 3, 1, 0, 31, False),          # ... with a power-of-two decomposition
    (64, 3, 2, -1, 16, True), (4096, 3, 2, -1, 24, True), (8192, 2, 1, -1, 20, False)])      # no P modulus at all
def test_gadget_product_single_p_and_bit_decomposition(rh, oracle, N, nq, levelQ, levelP, pw2, is_ntt):
    # gadgetProductSinglePAndBitDecompLazy (core/rlwe/evaluator_gadget_product.go:190-324) + ModDown (:33-98) vs the oracle composition
    from oracle import compose
    Q, P = QI60[:nq], PI60[:1]
    rq = rh.Ring(N, Q)
    rp = rh.Ring(N, P) if levelP == 0 else None
    rng = np.random.default_rng(N + nq * 7 + pw2)
    dpl = [(61 + pw2 - 1) // pw2 for _ in Q] if pw2 else None      # BaseTwoDecompositionVectorSize: ceil(log q_i / pw2) (params.go:615-633)
    rows = sum(dpl) if pw2 else nq
    evkQ = _rand_key(rng, rows, Q, N)
    evkP = _rand_key(rng, rows, P, N) if levelP == 0 else None
    LQ = levelQ + 1
    cx = np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(q) for q in Q[:LQ]]) for _ in range(2)])
    ev = rh.rlwe.Evaluator(rq, rp)
    gct = rh.rlwe.GadgetCiphertext(rq, rp, evkQ, evkP, BaseTwoDecomposition=pw2, digits_per_limb=dpl)
    assert gct.LevelP() == levelP
    rl = rq.AtLevel(levelQ)
    ct = rh.Ciphertext([rl.NewPoly(2), rl.NewPoly(2)], is_ntt=is_ntt)
    pcx = rh.DevicePoly.from_numpy(rl, cx)
    ev.GadgetProduct(levelQ, pcx, gct, ct)
    g0, g1 = ct.Value[0].numpy(), ct.Value[1].numpy()
    for k in range(2):
        e0, e1 = compose.gadget_product_single_p(N, Q, P if levelP == 0 else [], levelQ, levelP, cx[k], is_ntt, pw2, dpl, evkQ, evkP)
        assert np.array_equal(g0[k], e0), (k, 0)
        assert np.array_equal(g1[k], e1), (k, 1)
    assert np.array_equal(pcx.numpy(), cx)
    if pw2 == 0 and levelP == 0:                                    # no P modulus and no power-of-two decomposition: refused (the reference's call degenerates)
        bad = rh.rlwe.GadgetCiphertext(rq, None, evkQ, None)
        with pytest.raises(rh.RingHipError):
            rh.rlwe.Evaluator(rq, None).GadgetProduct(levelQ, pcx, bad, ct)
    ev.close(); rq.close()
    if rp is not None:
        rp.close()


def _generic_gadget_product(oracle, N, Q, P, levelQ, levelP, cx, evkQ, evkP, fq, iq, fp, ip):
    """GadgetProduct over a ring of any type: fq / iq (fp / ip) = forward / inverse transform of limb i of the Q (P) chain.  Every
    intermediate is a canonical residue: sum_d evk_d * c2_d * 2^-64 (what the reference's lazy accumulation and closing Reduce leave),
    ModDown as NTT(ModDownQPtoQ(INTT ...)) -- the transform is linear, so these are the bits of ModDownQPtoQNTT."""
    LQ, LP = levelQ + 1, levelP + 1
    Ql, Pl = Q[:LQ], P[:LP]
    beta = (levelQ + levelP + 1) // (levelP + 1)
    cxinv = np.stack([iq(cx[i], i) for i in range(LQ)])
    accQ = [[np.zeros(N, dtype=object) for _ in range(LQ)] for _ in range(2)]
    accP = [[np.zeros(N, dtype=object) for _ in range(LP)] for _ in range(2)]
    for d in range(beta):
        c2q, c2p = oracle.decompose_and_split(levelQ, levelP, LP, d, cxinv, Q, P)
        st, ed = d * LP, min(d * LP + LP, LQ)
        c2q = [cx[i] if st <= i < ed else fq(c2q[i], i) for i in range(LQ)]
        c2p = [fp(c2p[j], j) for j in range(LP)]
        for c in (0, 1):
            for i, q in enumerate(Ql):
                accQ[c][i] = (accQ[c][i] + evkQ[d, c, i].astype(object) * c2q[i].astype(object)) % q
            for j, p in enumerate(Pl):
                accP[c][j] = (accP[c][j] + evkP[d, c, j].astype(object) * c2p[j].astype(object)) % p
    out = []
    for c in (0, 1):
        aq = np.stack([iq(((accQ[c][i] * pow(1 << 64, -1, q)) % q).astype(np.uint64), i) for i, q in enumerate(Ql)])
        ap = np.stack([ip(((accP[c][j] * pow(1 << 64, -1, p)) % p).astype(np.uint64), j) for j, p in enumerate(Pl)])
        md = oracle.moddown_qp_to_q(aq, ap, Ql, Pl)
        out.append(np.stack([fq(md[i], i) for i in range(LQ)]))
    return out


@pytest.mark.parametrize("kind,N", [("ci", 256), ("ci", 4096), ("ci", 16384), ("3n", 3 << 6), ("3n", 3 << 13)])
def test_gadget_product_on_conjugate_invariant_and_3n_rings(rh, oracle, kind, N):
    # the key switch is generic over the ring type in the reference (rlwe.Evaluator works on whatever ringQ / ringP it is given): same
    # steps through the ring's own transform.  Direct product, the hoisted one, and the product with addends.
    from test_oracle_ntt3n import find_prime_3n, omega_for
    nq, np_, npoly = 5, 2, 2
    if kind == "ci":
        Q, P = QI60[:nq], PI60[:np_]
        rq, rp = rh.Ring(N, Q, kind=rh.ConjugateInvariant), rh.Ring(N, P, kind=rh.ConjugateInvariant)
        srQ = [oracle.SubRingConsts(N, q, nthroot=4 * N) for q in Q]; srP = [oracle.SubRingConsts(N, p, nthroot=4 * N) for p in P]
        fq = lambda x, i: oracle.ntt_ci(x, srQ[i]); iq = lambda x, i: oracle.intt_ci(x, srQ[i])
        fp = lambda x, j: oracle.ntt_ci(x, srP[j]); ip = lambda x, j: oracle.intt_ci(x, srP[j])
    else:
        mods, q = [], find_prime_3n(N, 60)
        while len(mods) < nq + np_:
            if oracle.lib().orc_is_prime(q):
                mods.append(q)
            q += 3 * N
        Q, P = mods[:nq], mods[nq:]
        wQ, wP = [omega_for(q, N) for q in Q], [omega_for(p, N) for p in P]
        rq, rp = rh.Ring(N, Q, kind=rh.Matrix3N, omega3n=wQ), rh.Ring(N, P, kind=rh.Matrix3N, omega3n=wP)
        fq = lambda x, i: oracle.ntt3n_forward(x, Q[i], wQ[i]); iq = lambda x, i: oracle.ntt3n_backward(x, Q[i], wQ[i])
        fp = lambda x, j: oracle.ntt3n_forward(x, P[j], wP[j]); ip = lambda x, j: oracle.ntt3n_backward(x, P[j], wP[j])
    levelQ, levelP = nq - 1, np_ - 1
    beta = (levelQ + levelP + 1) // (levelP + 1)
    rng = np.random.default_rng(N + nq)
    cx = np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(npoly)])
    add = np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(npoly)])
    evkQ, evkP = _rand_key(rng, beta, Q, N), _rand_key(rng, beta, P, N)
    be = rh.BasisExtender(rq, rp)
    dq = rh.DevicePoly.from_numpy(rq, evkQ.reshape(beta * 2, nq, N)); dp = rh.DevicePoly.from_numpy(rp, evkP.reshape(beta * 2, np_, N))
    pcx = rh.DevicePoly.from_numpy(rq, cx)
    ct0, ct1 = rh.DevicePoly(rq, npoly, nq), rh.DevicePoly(rq, npoly, nq)
    be.GadgetProduct(levelQ, levelP, pcx, dq.ptr, dp.ptr, beta, ct0, ct1)
    g0, g1 = ct0.numpy(), ct1.numpy()
    e0, e1 = _generic_gadget_product(oracle, N, Q, P, levelQ, levelP, cx[1], evkQ, evkP, fq, iq, fp, ip)
    assert np.array_equal(g0[1], e0) and np.array_equal(g1[1], e1)
    a0, a1 = rh.DevicePoly.from_numpy(rq, add), rh.DevicePoly.from_numpy(rq, add)
    be.GadgetProductThenAdd(levelQ, levelP, pcx, dq.ptr, dp.ptr, beta, a0, a1, a0, a1)
    for i, q in enumerate(Q):
        assert np.array_equal(a0.numpy()[1, i], (e0[i] + add[1, i]) % np.uint64(q))
        assert np.array_equal(a1.numpy()[1, i], (e1[i] + add[1, i]) % np.uint64(q))
    # hoisted == direct
    ev = rh.rlwe.Evaluator(rq, rp)
    gct = rh.rlwe.GadgetCiphertext(rq, rp, evkQ, evkP)
    dec = ev.DecomposeNTT(levelQ, levelP, pcx, True)
    h = rh.Ciphertext([rq.NewPoly(npoly), rq.NewPoly(npoly)], is_ntt=True)
    ev.GadgetProductHoisted(levelQ, dec, gct, h)
    assert np.array_equal(h.Value[0].numpy(), g0) and np.array_equal(h.Value[1].numpy(), g1)
    ev.close(); be.close(); rq.close(); rp.close()
