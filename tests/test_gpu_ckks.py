"""GPU: the CKKS multiply -> relinearise -> rescale chain (schemes/ckks/evaluator.go:786-881, 500-535) on device batches
against the same sequence on the CPU oracle.  Keys are uniformly random; scheme-level decryption
correctness is the reference's business and is not claimed here."""
import numpy as np
import pytest

from conftest import QI60, PI60, uniform_mod
from test_gpu_keyswitch import oracle_gadget_product

pytestmark = pytest.mark.gpu


def vop(oracle, rh, name, a, b, c, q):
    return oracle.vec_op(rh.OPS[name], a, b, c, 0, 0, q)


def oracle_tensor(oracle, rh, A, B, Q):
    """(:821-834) on (2, L, N) arrays -> c0, c1, c2"""
    L, N = len(Q), A.shape[2]
    z = np.zeros(N, dtype=np.uint64)
    c = [np.zeros((L, N), dtype=np.uint64) for _ in range(3)]
    for i, q in enumerate(Q):
        m0 = vop(oracle, rh, "MFORM", A[0, i], None, z, q)
        m1 = vop(oracle, rh, "MFORM", A[1, i], None, z, q)
        c[0][i] = vop(oracle, rh, "MUL_MONT", m0, B[0, i], z, q)
        c[2][i] = vop(oracle, rh, "MUL_MONT", m1, B[1, i], z, q)
        t = vop(oracle, rh, "MUL_MONT", m0, B[1, i], z, q)
        c[1][i] = vop(oracle, rh, "MUL_MONT_THEN_ADD", m1, B[0, i], t, q)
    return c


@pytest.mark.parametrize("N,nq,np_", [(64, 5, 2), (4096, 6, 3)])
def test_mul_relin_rescale_chain(rh, oracle, N, nq, np_):
    Q, P = QI60[:nq], PI60[:np_]
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    rng = np.random.default_rng(N * 3 + nq)
    beta = (nq - 1 + np_) // np_
    evkQ = np.stack([np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(2)]) for _ in range(beta)])
    evkP = np.stack([np.stack([np.stack([uniform_mod(rng, p, N) for p in P]) for _ in range(2)]) for _ in range(beta)])
    rlk = rh.rlwe.GadgetCiphertext(rq, rp, evkQ, evkP)
    ev = rh.ckks.Evaluator(rq, rp, rlk=rlk)
    B = 2
    mk = lambda: np.stack([np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(B)]) for _ in range(2)])   # (component, poly, limb, N)
    a, b = mk(), mk()
    ct0 = rh.Ciphertext([rh.DevicePoly.from_numpy(rq, a[0]), rh.DevicePoly.from_numpy(rq, a[1])], is_ntt=True)
    ct1 = rh.Ciphertext([rh.DevicePoly.from_numpy(rq, b[0]), rh.DevicePoly.from_numpy(rq, b[1])], is_ntt=True)
    # without relinearisation: degree 2
    out2 = rh.Ciphertext([rq.NewPoly(B) for _ in range(3)], is_ntt=True)
    ev.MulRelin(ct0, ct1, out2, relin=False)
    got2 = [v.numpy() for v in out2.Value]
    exp = [oracle_tensor(oracle, rh, a[:, k], b[:, k], Q) for k in range(B)]
    for k in range(B):
        for c in range(3):
            assert np.array_equal(got2[c][k], exp[k][c])
    ev.fused_tensor = False                                       # the six separate ring calls give the same bits
    out2b = rh.Ciphertext([rq.NewPoly(B) for _ in range(3)], is_ntt=True)
    ev.MulRelin(ct0, ct1, out2b, relin=False)
    for c in range(3):
        assert np.array_equal(out2b.Value[c].numpy(), got2[c])
    ev.fused_tensor = True
    # with relinearisation, then rescale
    out = rh.Ciphertext([rq.NewPoly(B), rq.NewPoly(B)], is_ntt=True)
    ev.MulRelin(ct0, ct1, out, relin=True)
    got = [v.numpy() for v in out.Value]
    res = rh.Ciphertext([rq.NewPoly(B), rq.NewPoly(B)], is_ntt=True)
    ev.Rescale(out, res)
    gres = [v.numpy() for v in res.Value]
    srQ = [oracle.SubRingConsts(N, q) for q in Q]
    for k in range(B):
        g0, g1 = oracle_gadget_product(oracle, rh, N, Q, P, nq - 1, np_ - 1, exp[k][2], evkQ, evkP)
        e = [np.stack([vop(oracle, rh, "ADD", exp[k][c][i], g[i], g[i], Q[i]) for i in range(nq)]) for c, g in ((0, g0), (1, g1))]
        for c in range(2):
            assert np.array_equal(got[c][k], e[c])
            coeff = np.stack([oracle.intt(e[c][i], srQ[i]) for i in range(nq)])
            down = oracle.div_by_last_modulus_many(coeff, Q, 1, True)
            want = np.stack([oracle.ntt(down[i], srQ[i]) for i in range(nq - 1)])
            assert np.array_equal(gres[c][k, :nq - 1], want)
    # squaring (:825-829) and plaintext x ciphertext (:855-878)
    sq = rh.Ciphertext([rq.NewPoly(B) for _ in range(3)], is_ntt=True)
    ev.MulRelin(ct0, ct0, sq, relin=False)
    for k in range(B):
        es = oracle_tensor(oracle, rh, a[:, k], a[:, k], Q)
        for c in range(3):
            assert np.array_equal(sq.Value[c].numpy()[k], es[c])
    pt = rh.Ciphertext([rh.DevicePoly.from_numpy(rq, b[0])], is_ntt=True)
    pc = rh.Ciphertext([rq.NewPoly(B), rq.NewPoly(B)], is_ntt=True)
    ev.MulRelin(pt, ct0, pc)
    z = np.zeros(N, dtype=np.uint64)
    for i, q in enumerate(Q):
        m = vop(oracle, rh, "MFORM", b[0, 1, i], None, z, q)
        assert np.array_equal(pc.Value[1].numpy()[1, i], vop(oracle, rh, "MUL_MONT", m, a[1, 1, i], z, q))
    # errors: missing key (:838-842), level too low for a rescale (:511-513)
    with pytest.raises(rh.RingHipError):
        rh.ckks.Evaluator(rq).MulRelin(ct0, ct1, out, relin=True)
    r1 = rh.Ring(N, Q[:1])
    low = rh.Ciphertext([r1.NewPoly(B), r1.NewPoly(B)], is_ntt=True)
    with pytest.raises(rh.RingHipError):
        rh.ckks.Evaluator(r1).Rescale(low, low)
    ev.close(); rq.close(); rp.close(); r1.close()
