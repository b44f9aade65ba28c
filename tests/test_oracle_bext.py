"""CPU: pins the oracle's basis-extension restatement against big-integer ground truth, the way the reference's own
tests do (ring/ring_test.go:707-884: random centred big integers -> RNS -> ModUp/ModDown -> compare after Reduce)."""
import numpy as np
import pytest

from conftest import QI60, PI60


def prod(xs):
    r = 1
    for x in xs:
        r *= x
    return r


def rns(vals, mods):
    return np.array([[v % m for v in vals] for m in mods], dtype=np.uint64)


def centered_randoms(rng, bound, n):
    # uniform in (-bound/2, bound/2) like the reference (random big ints, centred)
    out = []
    for _ in range(n):
        v = int.from_bytes(rng.bytes(int(bound).bit_length() // 8 + 2), "little") % bound
        out.append(v - bound // 2)
    return out


@pytest.mark.parametrize("nq,np_", [(6, 3), (1, 1), (13, 14), (4, 1), (32, 2)])
def test_modup_q_to_p_vs_bigint(oracle, nq, np_):
    # TestModUpQtoP-style (:707-749): exact centred extension
    Q, P = (QI60 + PI60)[:nq], (PI60[::-1])[:np_]
    rng = np.random.default_rng(nq * 100 + np_)
    n = 64
    vals = centered_randoms(rng, prod(Q), n)
    pq = rns(vals, Q)
    pp = oracle.modup_centered(pq, Q, P)
    for j, p in enumerate(P):
        assert int(pp[j].max()) < 3 * p                       # lazy range (SURVEY a.5: < 3p, not canonical)
        assert [int(x) % p for x in pp[j]] == [v % p for v in vals]


def test_moddown_qp_to_q_vs_bigint(oracle):
    # TestModDownQPtoQ-style (:751-800): round(x / P) mod Q
    Q, P = QI60[:6], PI60[:3]
    rng = np.random.default_rng(9)
    n = 64
    bigQP = prod(Q) * prod(P)
    vals = centered_randoms(rng, bigQP, n)
    pq, pp = rns(vals, Q), rns(vals, P)
    out = oracle.moddown_qp_to_q(pq, pp, Q, P)
    Pb = prod(P)
    for i, q in enumerate(Q):
        exp = [((2 * v + Pb) // (2 * Pb)) % q for v in vals]      # rounded division
        assert [int(x) for x in out[i]] == exp


def test_moddown_ntt_equals_coefficient_moddown(oracle):
    N = 64
    Q, P = QI60[:3], PI60[:2]
    srQ = [oracle.SubRingConsts(N, q) for q in Q]
    srP = [oracle.SubRingConsts(N, p) for p in P]
    rng = np.random.default_rng(3)
    vals = centered_randoms(rng, prod(Q) * prod(P), N)
    pq, pp = rns(vals, Q), rns(vals, P)
    ref = oracle.moddown_qp_to_q(pq, pp, Q, P)
    nq = np.stack([oracle.ntt(pq[i], srQ[i]) for i in range(len(Q))])
    npp = np.stack([oracle.ntt(pp[j], srP[j]) for j in range(len(P))])
    out = oracle.moddown_qp_to_q_ntt(nq, npp, Q, P, srQ, srP)
    back = np.stack([oracle.intt(out[i], srQ[i]) for i in range(len(Q))])
    assert np.array_equal(back, ref)


@pytest.mark.parametrize("levelQ,levelP,nbPi,digit", [(5, 1, 2, 0), (5, 1, 2, 2), (6, 2, 3, 2), (4, 1, 2, 2), (23, 5, 6, 3), (3, 0, 1, 2)])
def test_decompose_and_split_vs_bigint(oracle, levelQ, levelP, nbPi, digit):
    # digit = centred (x mod Q_digit), re-expressed modulo every limb of Q and P
    Qall, Pall = QI60[:levelQ + 1], PI60[:max(levelP + 1, nbPi)]
    rng = np.random.default_rng(levelQ * 10 + digit)
    n = 32
    st = digit * nbPi
    ed = min(st + nbPi, levelQ + 1)
    vals = centered_randoms(rng, prod(Qall), n)
    p0 = rns(vals, Qall)
    oq, op = oracle.decompose_and_split(levelQ, levelP, nbPi, digit, p0, Qall, Pall)
    Qd = prod(Qall[st:ed])
    cent = []
    for v in vals:
        r = v % Qd
        if ed - st == 1:
            cent.append(r - Qd if r >= (Qd >> 1) else r)        # single prime: sign rule of :411-415
        else:
            cent.append(r - Qd if r > Qd // 2 else r)
    for j in range(levelQ + 1):
        if st <= j < ed:
            continue
        got = [int(x) % Qall[j] for x in oq[j]]
        assert got == [c % Qall[j] for c in cent], j
    for j in range(levelP + 1):
        assert [int(x) % Pall[j] for x in op[j]] == [c % Pall[j] for c in cent], j


@pytest.mark.parametrize("nq,np_", [(6, 3), (2, 2), (14, 6), (24, 6)])
def test_modup_exact_at_the_ends_of_the_centred_range(oracle, nq, np_):
    # values for which sum_i y_i/q_i is within a rounding error of an integer (see tests/test_gpu_bext.py).  The reference's
    # extension (+ Q/2 before, - Q/2 after: ring/basis_extension.go:188-217) is exact except in bands of relative width < 2^-40 at
    # the two ENDS of the centred range, where the double sum falls on the wrong side of an integer and v is off by one: there the
    # result is X -+ Q instead of X (the algorithm's approximation, which random test values never meet).  The restatement must show
    # exactly that behaviour: exact everywhere else, off by one multiple of Q (and nothing else) in the bands.
    from conftest import QI60, PI60
    Q, P = QI60[:nq], PI60[:np_]
    bigQ = prod(Q)
    half = bigQ // 2
    vals = [0, 1, -1, 2, -2, half, -half, half - 1, -(half - 1), half - 2, half // 2, -(half // 2)]
    vals += [s * (1 << k) for k in (10, 40, 61, 100) for s in (1, -1) if (1 << k) < half]
    vals += [half - (1 << k) for k in (1, 20, 50) if (1 << k) < half]
    a = rns(vals, Q)
    out = oracle.modup_centered(a, Q, P)
    band = half - (half >> 40)          # |X| beyond it: the sum of up to 32 rounded quotients can fall on the wrong side of the integer
    for j, p in enumerate(P):
        for x, v in zip(out[j], vals):
            if abs(v) <= band:
                assert int(x) % p == v % p, v
            else:
                assert int(x) % p in (v % p, (v - bigQ) % p, (v + bigQ) % p), v
    assert any(int(out[0][i]) % P[0] != vals[i] % P[0] for i in range(len(vals)) if abs(vals[i]) > band)   # the band is real
