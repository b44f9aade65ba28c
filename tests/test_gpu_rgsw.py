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
    with pytest.raises(rh.RingHipError):                                   # single-P RGSW ciphertexts are refused, not mis-computed
        rp1 = rh.Ring(N, P[:1])
        g = rh.rlwe.GadgetCiphertext(rq, rp1, _key(rng, nq, Q, N), _key(rng, nq, P[:1], N))
        rh.rgsw.Evaluator(rq, rp1).ExternalProduct(op0, rh.rgsw.Ciphertext(g, g), out)
    ev.close(); rq.close(); rp.close()
