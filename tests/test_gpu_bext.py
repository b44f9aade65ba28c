"""GPU: RNS basis extension through the C ABI, bit-exact (including the reference's non-canonical lazy outputs)
against the oracle that test_oracle_bext.py pins to big-integer ground truth."""
import numpy as np
import pytest

from conftest import QI60, PI60
from test_oracle_bext import centered_randoms, prod, rns

pytestmark = pytest.mark.gpu


def make(rh, N, nq, np_):
    Q, P = QI60[:nq], PI60[:np_]
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    return Q, P, rq, rp, rh.BasisExtender(rq, rp)


def rand_rns(rng, mods, npoly, N, full=True):
    vals = [centered_randoms(rng, prod(mods), N) for _ in range(npoly)]
    return vals, np.stack([rns(v, mods) for v in vals])


@pytest.mark.parametrize("N,nq,np_,levelQ,levelP", [(64, 6, 3, 5, 2), (1024, 14, 14, 12, 13), (4096, 24, 6, 23, 5), (64, 4, 2, 0, 0), (256, 32, 4, 31, 3)])
def test_modup_both_directions(rh, oracle, N, nq, np_, levelQ, levelP):
    Q, P, rq, rp, be = make(rh, N, nq, np_)
    rng = np.random.default_rng(N + nq)
    npoly = 2
    Ql, Pl = Q[:levelQ + 1], P[:levelP + 1]
    vals, a = rand_rns(rng, Ql, npoly, N)
    pa = rh.DevicePoly.from_numpy(rq.AtLevel(levelQ), a)
    pp = rh.DevicePoly(rp, npoly, levelP + 1)
    be.ModUpQtoP(levelQ, levelP, pa, pp)
    got = pp.numpy()
    for k in range(npoly):
        exp = oracle.modup_centered(a[k], Ql, Pl)
        assert np.array_equal(got[k], exp)
        for j, p in enumerate(Pl):
            assert [int(x) % p for x in got[k, j][:16]] == [v % p for v in vals[k][:16]]
    # P -> Q
    vals2, b = rand_rns(rng, Pl, npoly, N)
    pb = rh.DevicePoly.from_numpy(rp.AtLevel(levelP), b)
    pq = rh.DevicePoly(rq, npoly, levelQ + 1)
    be.ModUpPtoQ(levelP, levelQ, pb, pq)
    got = pq.numpy()
    for k in range(npoly):
        assert np.array_equal(got[k], oracle.modup_centered(b[k], Pl, Ql))
    be.close(); rq.close(); rp.close()


@pytest.mark.parametrize("N,nq,np_", [(64, 6, 3), (2048, 14, 14), (8192, 5, 2)])
def test_moddown_variants(rh, oracle, N, nq, np_):
    Q, P, rq, rp, be = make(rh, N, nq, np_)
    rng = np.random.default_rng(N * 3 + np_)
    npoly = 2
    levelQ, levelP = nq - 1, np_ - 1
    QP = Q + P
    vals = [centered_randoms(rng, prod(QP), N) for _ in range(npoly)]
    aq = np.stack([rns(v, Q) for v in vals]); ap = np.stack([rns(v, P) for v in vals])
    pq, pp = rh.DevicePoly.from_numpy(rq, aq), rh.DevicePoly.from_numpy(rp, ap)
    out = rq.NewPoly(npoly)
    be.ModDownQPtoQ(levelQ, levelP, pq, pp, out)
    got = out.numpy()
    Pb = prod(P)
    for k in range(npoly):
        assert np.array_equal(got[k], oracle.moddown_qp_to_q(aq[k], ap[k], Q, P))
        for i, q in enumerate(Q):
            assert [int(x) for x in got[k, i][:8]] == [((2 * v + Pb) // (2 * Pb)) % q for v in vals[k][:8]]
    # QP -> P (floored/rounded division by Q, mirror image)
    outp = rp.NewPoly(npoly)
    be.ModDownQPtoP(levelQ, levelP, pq, pp, outp)
    gp = outp.numpy()
    for k in range(npoly):
        assert np.array_equal(gp[k], oracle.moddown_qp_to_q(ap[k], aq[k], P, Q))
    # NTT-domain variant: NTT inputs -> ModDownQPtoQNTT -> equals NTT of the coefficient-domain result
    srQ = [oracle.SubRingConsts(N, q) for q in Q]; srP = [oracle.SubRingConsts(N, p) for p in P]
    rq.NTT(pq, pq); rp.NTT(pp, pp)
    nq_np, np_np = pq.numpy(), pp.numpy()
    be.ModDownQPtoQNTT(levelQ, levelP, pq, pp, out)
    gn = out.numpy()
    for k in range(npoly):
        assert np.array_equal(gn[k], oracle.moddown_qp_to_q_ntt(nq_np[k], np_np[k], Q, P, srQ, srP))
    rq.INTT(out, out)
    assert np.array_equal(out.numpy(), got)
    be.close(); rq.close(); rp.close()


@pytest.mark.parametrize("N,levelQ,levelP,nbPi,digit", [(64, 5, 1, 2, 0), (64, 5, 1, 2, 2), (128, 6, 2, 3, 2), (64, 4, 1, 2, 2),
                                                           (4096, 23, 5, 6, 3), (4096, 23, 5, 6, 0), (64, 3, 0, 1, 2)])
def test_decompose_and_split(rh, oracle, N, levelQ, levelP, nbPi, digit):
    nq, np_ = levelQ + 1, max(levelP + 1, nbPi)
    Q, P, rq, rp, be = make(rh, N, nq, np_)
    rng = np.random.default_rng(levelQ * 7 + digit)
    npoly = 2
    vals, a = rand_rns(rng, Q, npoly, N)
    p0 = rh.DevicePoly.from_numpy(rq, a)
    oq = rh.DevicePoly.from_numpy(rq, np.zeros((npoly, nq, N), dtype=np.uint64))
    op = rh.DevicePoly.from_numpy(rp.AtLevel(levelP), np.zeros((npoly, levelP + 1, N), dtype=np.uint64))
    be.DecomposeAndSplit(levelQ, levelP, nbPi, digit, p0, oq, op)
    gq, gp = oq.numpy(), op.numpy()
    for k in range(npoly):
        eq, ep = oracle.decompose_and_split(levelQ, levelP, nbPi, digit, a[k], Q, P)
        assert np.array_equal(gq[k], eq)
        assert np.array_equal(gp[k], ep)
    be.close(); rq.close(); rp.close()


def test_config5_shape_keyswitch_digit(rh, oracle):
    # BASELINE config 5 shapes at reduced N: Q = Qi60[0:24], P = Pi60[0:6], alpha = 6, beta = 4: every digit
    N = 1 << 13
    Q, P, rq, rp, be = make(rh, N, 24, 6)
    rng = np.random.default_rng(55)
    vals, a = rand_rns(rng, Q, 1, N)
    p0 = rh.DevicePoly.from_numpy(rq, a)
    for digit in range(4):
        oq = rh.DevicePoly.from_numpy(rq, np.zeros((1, 24, N), dtype=np.uint64))
        op = rh.DevicePoly.from_numpy(rp, np.zeros((1, 6, N), dtype=np.uint64))
        be.DecomposeAndSplit(23, 5, 6, digit, p0, oq, op)
        eq, ep = oracle.decompose_and_split(23, 5, 6, digit, a[0], Q, P)
        assert np.array_equal(oq.numpy()[0], eq) and np.array_equal(op.numpy()[0], ep)
    be.close(); rq.close(); rp.close()


@pytest.mark.parametrize("nq,np_", [(6, 3), (2, 2), (14, 6)])
def test_adversarial_values_for_the_floating_point_v(rh, oracle, nq, np_):
    # reconstructRNS decides v = trunc(sum_i float64(y_i) / float64(q_i)) in IEEE double with sequential adds
    # (ring/basis_extension.go:576-593).  For a centred value X the exact sum is an integer + (X + Q/2)/Q, so values at the ends of
    # the centred range (X = +-(Q-1)/2, +-(Q/2 - small)) and the values whose shifted form is 0, 1, Q-1 ... put the sum within one
    # rounding error of an integer: v then hinges on the exact rounding of every division and addition (at the very top of the range
    # the reference's sum rounds up and the result is X - Q: tests/test_oracle_bext.py).  GPU and oracle (gcc, -ffp-contract=off)
    # must agree bit for bit on all of them, including the lazy (non-canonical) outputs and the reference's off-by-Q band.
    N = 256
    Q, P, rq, rp, be = make(rh, N, nq, np_)
    bigQ = prod(Q)
    half = bigQ // 2
    specials = [0, 1, -1, 2, -2, half, -half, half - 1, -(half - 1), half - 2, half // 2, -(half // 2), 3 * (half // 4)]
    specials += [s * (1 << k) for k in (10, 40, 61, 100) for s in (1, -1) if (1 << k) < half]
    specials += [half - (1 << k) for k in (1, 20, 50) if (1 << k) < half] + [-(half - (1 << k)) for k in (1, 20, 50) if (1 << k) < half]
    rng = np.random.default_rng(nq * 31 + np_)
    vals = [specials[i % len(specials)] if i < 2 * len(specials) else centered_randoms(rng, bigQ, 1)[0] for i in range(N)]
    a = np.stack([rns(vals, Q)])
    pa = rh.DevicePoly.from_numpy(rq, a)
    pp = rh.DevicePoly(rp, 1, np_)
    be.ModUpQtoP(nq - 1, np_ - 1, pa, pp)
    got = pp.numpy()[0]
    assert np.array_equal(got, oracle.modup_centered(a[0], Q, P))
    band = half - (half >> 40)
    for j, p in enumerate(P):                                         # and the values are right (outside the bands), not merely equal
        assert all(int(x) % p == v % p for x, v in zip(got[j], vals) if abs(v) <= band)
    # the digit decomposition of the key switch on the same coefficients (centred reconstruction with the raw add, :504-548)
    if nq >= 4:
        alpha = np_
        oq, op_ = rh.DevicePoly(rq, 1, nq), rh.DevicePoly(rp, 1, np_)
        be.DecomposeAndSplit(nq - 1, np_ - 1, alpha, 0, pa, oq, op_)
        eq, ep = oracle.decompose_and_split(nq - 1, np_ - 1, alpha, 0, a[0], Q, P)
        st, ed = 0, min(alpha, nq)
        keep = [i for i in range(nq) if not (st <= i < ed)]
        assert np.array_equal(oq.numpy()[0][keep], eq[keep]) and np.array_equal(op_.numpy()[0], ep)
    be.close(); rq.close(); rp.close()
