"""CPU: pins the oracle's NTT restatement against the reference's own known-answer vectors
(ring/ntt_test.go:10-89 -> tests/golden/ntt_kat.json) and its property tests (ring/ring_test.go:534-705)."""
import json
import os

import numpy as np
import pytest

from conftest import QI60, uniform_mod

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ntt_kat.json")))["vectors"]


@pytest.mark.parametrize("vec", KAT, ids=lambda v: "N=%d" % v["N"])
def test_ntt_kat_forward_and_inverse(oracle, vec):
    # TestNTT (ring/ntt_test.go:91-121): NTT(poly) == polyNTT, INTT(NTT(poly)) == poly, per limb
    for q, a, b in zip(vec["Qis"], vec["poly"], vec["polyNTT"]):
        sr = oracle.SubRingConsts(vec["N"], q)
        y = oracle.ntt(a, sr)
        assert np.array_equal(y, np.array(b, dtype=np.uint64))
        assert np.array_equal(oracle.intt(y, sr), np.array(a, dtype=np.uint64))
        # lazy forward: congruent and inside the documented range [0, 6q-2] (ring/ntt.go:64)
        yl = oracle.ntt(a, sr, lazy=True)
        assert int(yl.max()) <= 6 * q - 2
        assert np.array_equal(yl % np.uint64(q), y)


def test_scalar_primitives_vs_bigint(oracle):
    # TestMRed / TestBRed style (ring/ring_test.go:534-670): edge operands and random ones against Python ints
    L = oracle.lib()
    rng = np.random.default_rng(1)
    for q in QI60[:4] + [576460752303439873, 0xffffffffffc0001 >> 4 | 1]:
        if not L.orc_is_prime(q):
            continue
        qinv = L.orc_gen_mred_constant(q)
        assert (qinv * q) % (1 << 64) == 1
        b = np.zeros(2, dtype=np.uint64)
        L.orc_gen_bred_constant(q, b.ctypes.data_as(oracle.ring_oracle.U64P))
        assert (int(b[0]) << 64) + int(b[1]) == (1 << 128) // q
        bp = b.ctypes.data_as(oracle.ring_oracle.U64P)
        rinv = pow(1 << 64, -1, q)
        xs = [1, q - 1, 0, (1 << 64) - 1, 2, q // 2] + [int(v) for v in rng.integers(0, 1 << 63, size=50, dtype=np.uint64)]
        for x in xs:
            for y in xs[:8]:
                xr, yr = x % q, y % q
                assert L.orc_mred(xr, yr, q, qinv) == (xr * yr * rinv) % q
                assert L.orc_mred_lazy(xr, yr, q, qinv) % q == (xr * yr * rinv) % q
                assert L.orc_mred_lazy(xr, yr, q, qinv) < 2 * q
                assert L.orc_bred(x, y, q, bp) == (x * y) % q
                assert L.orc_bred_lazy(x, y, q, bp) % q == (x * y) % q
            assert L.orc_bred_add(x, q, bp) == x % q
            assert L.orc_mform(x % q, q, bp) == ((x % q) << 64) % q
            assert L.orc_imform(x % q, q, qinv) == ((x % q) * rinv) % q
            assert L.orc_imform(L.orc_mform(x % q, q, bp), q, qinv) == x % q      # TestMForm round trip (:672-690)
            assert L.orc_modexp(x % q, 65537, q) == pow(x % q, 65537, q)


def test_tables_follow_reference_rules(oracle):
    # generateNTTConstants (ring/subring.go:129-214): psi = g^((q-1)/2N), tables[bitrev(j)] = psi^j in Montgomery form
    N, q = 64, QI60[0]
    sr = oracle.SubRingConsts(N, q)
    g = sr.primitive_root
    assert g >= 3
    for f in (2, 3, 5, 7):
        if (q - 1) % f == 0:
            assert pow(g, (q - 1) // f, q) != 1
    psi = pow(g, (q - 1) // (2 * N), q)
    R = 1 << 64
    for j in range(N):
        idx = int(format(j, "06b")[::-1], 2)
        assert int(sr.roots_fwd[idx]) == (pow(psi, j, q) * R) % q
        assert int(sr.roots_bwd[idx]) == (pow(psi, -j, q) * R) % q
    assert sr.ninv == (pow(N, -1, q) * R) % q


@pytest.mark.parametrize("logN", [4, 5, 8, 11, 12, 13])
def test_ntt_is_negacyclic_evaluation(oracle, logN):
    # the forward output at bit-reversed slot i equals f(psi^(2*bitrev(i)+1)): checks a few slots against Horner
    N, q = 1 << logN, QI60[1]
    sr = oracle.SubRingConsts(N, q)
    rng = np.random.default_rng(logN)
    a = uniform_mod(rng, q, N)
    y = oracle.ntt(a, sr)
    psi = pow(sr.primitive_root, (q - 1) // (2 * N), q)
    for i in [0, 1, N // 2, N - 1, 5 % N]:
        e = 2 * int(format(i, "0%db" % logN)[::-1], 2) + 1
        x = pow(psi, e, q)
        acc = 0
        for c in reversed(a.tolist()):
            acc = (acc * x + int(c)) % q
        assert int(y[i]) == acc
    assert np.array_equal(oracle.intt(y, sr), a)
    assert np.array_equal(oracle.intt(y, sr, lazy=True), a)   # N >= 16: INTTStandardLazy is fully reduced (ntt.go:203-205)
