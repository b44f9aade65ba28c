"""GPU: rgsw.Evaluator.ExternalProduct (core/rgsw/evaluator.go:42-80, 188-257, LevelP >= 1) against the reference's loop restated over
the oracle pieces (oracle/compose.py: one pair of lazy accumulators through both components and all digits, running Reduce counter,
closing Reduce, ModDownQPtoQNTT).  Uniformly random RGSW values: arithmetic parity needs no encryption (the reference pins the external
product only through decryption noise, core/rgsw/rgsw_test.go:60-129: that end-to-end statement stays parity unpinned)."""
import numpy as np
import pytest

from conftest import QI60, PI60, uniform_mod

pytestmark = pytest.mark.gpu


def _key(rng, digits, mods, N):
    return np.stack([np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(2)]) for _ in range(digits)])


@pytest.mark.parametrize("N,nq,np_,is_ntt,inplace", [(64, 4, 2, True, False), (4096, 5, 2, True, True), (8192, 6, 3, False, False),
                                                     (1 << 14, 7, 2, True, False), (1 << 15, 4, 4, False, True), (1 << 16, 6, 2, True, False)])
def test_external_product_vs_reference_loop(rh, oracle, N, nq, np_, is_ntt, inplace):
    from oracle import compose
    Q, P = QI60[:nq], PI60[:np_]
    rng = np.random.default_rng(N + nq + np_)
    levelQ, levelP = nq - 1, np_ - 1
    beta = (levelQ + levelP + 1) // (levelP + 1)
    B = 2
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    ev = rh.rgsw.Evaluator(rq, rp)
    kq = [_key(rng, beta, Q, N) for _ in (0, 1)]
    kp = [_key(rng, beta, P, N) for _ in (0, 1)]
    rgsw = rh.rgsw.Ciphertext(rh.rlwe.GadgetCiphertext(rq, rp, kq[0], kp[0]), rh.rlwe.GadgetCiphertext(rq, rp, kq[1], kp[1]))
    c = [np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(B)]) for _ in (0, 1)]
    op0 = rh.Ciphertext([rh.DevicePoly.from_numpy(rq, c[0]), rh.DevicePoly.from_numpy(rq, c[1])], is_ntt=is_ntt)
    if inplace:
        out = op0
        if not is_ntt:
            out = rh.Ciphertext(op0.Value, is_ntt=True)               # the same buffers, flagged as the NTT-domain result they will hold
    else:
        out = rh.Ciphertext([rq.NewPoly(B), rq.NewPoly(B)], is_ntt=True)
    ev.ExternalProduct(op0, rgsw, out)
    g0, g1 = out.Value[0].numpy(), out.Value[1].numpy()
    for k in range(B):
        e0, e1 = compose.external_product(N, Q, P, levelQ, levelP, np.stack([c[0][k], c[1][k]]), is_ntt, kq, kp)
        assert np.array_equal(g0[k], e0), "component 0, poly %d" % k
        assert np.array_equal(g1[k], e1), "component 1, poly %d" % k
    if not inplace:
        assert np.array_equal(op0.Value[0].numpy(), c[0]) and np.array_equal(op0.Value[1].numpy(), c[1])
    ev.close(); rq.close(); rp.close()


@pytest.mark.parametrize("N,nq,levelQ,levelP,pw2", [(64, 3, 2, 0, 0), (4096, 4, 3, 0, 0), (8192, 3, 2, 0, 20), (64, 3, 2, -1, 16), (4096, 2, 1, -1, 31),
                                                     (4096, 3, 1, 0, 0), (1 << 14, 2, 1, -1, 0)])
def test_external_product_single_p_and_bit_decomposition(rh, oracle, N, nq, levelQ, levelP, pw2):
    """LevelP <= 0 (core/rgsw/evaluator.go:55-70, 119-186): RNS digits by MaskVec of each limb (all-ones mask without a power-of-two
    decomposition), optionally base-2^pw2 digits on top, with one P modulus (ModDownQPtoQNTT) or none (CopyLvl)"""
    from oracle import compose
    Q, P = QI60[:nq], PI60[:1]
    rng = np.random.default_rng(N + nq + pw2 + levelP)
    B = 2
    rq = rh.Ring(N, Q)
    rp = rh.Ring(N, P) if levelP == 0 else None
    ev = rh.rgsw.Evaluator(rq, rp)
    dpl = [-(-int(q).bit_length() // pw2) for q in Q] if pw2 else None      # ceil(bitlen(q_i) / pw2) digits per limb (core/rlwe/params.go:615-633)
    rows = sum(dpl) if pw2 else nq
    kq = [_key(rng, rows, Q, N) for _ in (0, 1)]
    kp = [_key(rng, rows, P, N) for _ in (0, 1)] if rp is not None else [None, None]
    mk = lambda k: rh.rlwe.GadgetCiphertext(rq, rp, kq[k], kp[k], BaseTwoDecomposition=pw2, digits_per_limb=dpl)
    rgsw = rh.rgsw.Ciphertext(mk(0), mk(1))
    # the product runs at the RGSW ciphertext's own levels (:44): give it levelQ by building the ring view's blocks at that level
    rl = rq.AtLevel(levelQ)
    c = [np.stack([np.stack([uniform_mod(rng, q, N) for q in Q[:levelQ + 1]]) for _ in range(B)]) for _ in (0, 1)]
    op0 = rh.Ciphertext([rh.DevicePoly.from_numpy(rl, c[0]), rh.DevicePoly.from_numpy(rl, c[1])], is_ntt=True)
    out = rh.Ciphertext([rl.NewPoly(B), rl.NewPoly(B)], is_ntt=True)
    for g in rgsw.Value:
        g.levelQ = levelQ                                                  # a gadget ciphertext at a lower level: its leading limbs
    ev.ExternalProduct(op0, rgsw, out)
    g0, g1 = out.Value[0].numpy(), out.Value[1].numpy()
    dd = dpl[:levelQ + 1] if pw2 else None
    for k in range(B):
        e0, e1 = compose.external_product_single_p(N, Q, P, levelQ, levelP, np.stack([c[0][k], c[1][k]]), pw2, dd, kq, kp)
        assert np.array_equal(g0[k], e0), "component 0, poly %d" % k
        assert np.array_equal(g1[k], e1), "component 1, poly %d" % k
    ev.close(); rq.close()
    if rp is not None:
        rp.close()


@pytest.mark.parametrize("N,pw2,levelP", [(4096, 0, 0), (4096, 20, 0), (8192, 16, -1), (4096, 0, -1)])
def test_external_product_single_p_mixed_size_moduli(rh, oracle, N, pw2, levelP):
    """the MaskVec digit of limb i (< q_i, or < 2^pw2) is used under EVERY modulus; with 61-, 41- and 37-bit primes in one chain it
    exceeds the small ones by up to 2^24 x, far outside the [0, 8q) the forward transform's first stage assumes.  The kernel writes the
    canonical residue per target limb; the oracle restates the reference (raw window into NTTLazy): both end in the same canonical bits."""
    from oracle import compose
    Q = [QI60[0], 0x10000140001, 0x10004a0001, QI60[1]]               # 61, 41, 37, 61 bits; all = 1 mod 2^17
    P = [0x100003e0001]                                               # a 41-bit special modulus
    nq, levelQ = len(Q), len(Q) - 1
    rng = np.random.default_rng(N + pw2 + levelP + 5)
    B = 2
    rq = rh.Ring(N, Q)
    rp = rh.Ring(N, P) if levelP == 0 else None
    ev = rh.rgsw.Evaluator(rq, rp)
    dpl = [-(-int(q).bit_length() // pw2) for q in Q] if pw2 else None
    rows = sum(dpl) if pw2 else nq
    kq = [_key(rng, rows, Q, N) for _ in (0, 1)]
    kp = [_key(rng, rows, P, N) for _ in (0, 1)] if rp is not None else [None, None]
    mk = lambda k: rh.rlwe.GadgetCiphertext(rq, rp, kq[k], kp[k], BaseTwoDecomposition=pw2, digits_per_limb=dpl)
    rgsw = rh.rgsw.Ciphertext(mk(0), mk(1))
    c = [np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(B)]) for _ in (0, 1)]
    op0 = rh.Ciphertext([rh.DevicePoly.from_numpy(rq, c[0]), rh.DevicePoly.from_numpy(rq, c[1])], is_ntt=True)
    out = rh.Ciphertext([rq.NewPoly(B), rq.NewPoly(B)], is_ntt=True)
    ev.ExternalProduct(op0, rgsw, out)
    g0, g1 = out.Value[0].numpy(), out.Value[1].numpy()
    for k in range(B):
        e0, e1 = compose.external_product_single_p(N, Q, P, levelQ, levelP, np.stack([c[0][k], c[1][k]]), pw2, dpl, kq, kp)
        assert np.array_equal(g0[k], e0), "component 0, poly %d" % k
        assert np.array_equal(g1[k], e1), "component 1, poly %d" % k
    ev.close(); rq.close()
    if rp is not None:
        rp.close()
