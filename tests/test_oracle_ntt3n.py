"""CPU: pins the oracle's 3N-cyclotomic restatement.
(1) definition (Horner at omega^E[k], ring/ntt_3n.go:82-109) == fast factorisation, for the reference's test sizes
    N in {6,12,18,24,36,48} (ring/ntt_3n_test.go) and more;
(2) fast factorisation == references/integer_dft.py outputs (tests/golden/ntt3n_intdft.json), through the tree -> ascending
    totative permutation;
(3) Backward definition (Vandermonde solve, :118-151) == fast inverse; round trips;
(4) multiplication through the transform == naive reduction with X^N = X^(N/2) - 1 (ntt_3n_test.go:312-364)."""
import json
import os

import numpy as np
import pytest

VEC = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ntt3n_intdft.json")))["vectors"]


def rank(e):
    return 2 * (e // 6) + (1 if e % 6 == 5 else 0)


def find_prime_3n(N, bits):
    import oracle
    step = 3 * N
    c = ((1 << bits) // step + 1) * step + 1          # Find3NRNSPrimes stepping rule (ring/primes_3n.go:11-43)
    while not oracle.lib().orc_is_prime(c):
        c += step
    return c


def omega_for(q, N):
    import oracle
    g = oracle.lib().orc_primitive_root(q)
    return pow(g, (q - 1) // (3 * N), q)               # Find3NPrimitiveRoot (ring/subring.go:255-290)


@pytest.mark.parametrize("vec", VEC, ids=lambda v: "N=%d" % v["N"])
def test_fast_matches_reference_python_notes(oracle, vec):
    N, p, w = vec["N"], vec["p"], vec["w"]
    x = np.array(vec["input"], dtype=np.uint64)
    y = oracle.ntt3n_forward(x, p, w, fast=True)
    exp = np.zeros(N, dtype=np.uint64)
    for s, e in enumerate(vec["tree_last"]):
        exp[rank(e)] = vec["dft_tree_order"][s]
    assert np.array_equal(y, exp)
    # and both equal the Go definition: out[k] = f(w^E[k]), E ascending totatives
    assert np.array_equal(y, oracle.ntt3n_forward(x, p, w, fast=False))
    E = oracle.ntt3n_exponents(3 * N)
    assert len(E) == N and E == sorted(E)
    for k in (0, 1, N // 2, N - 1):
        acc = 0
        for c in reversed(vec["input"]):
            acc = (acc * pow(w, E[k], p) + c) % p
        assert int(y[k]) == acc
    assert np.array_equal(oracle.ntt3n_backward(y, p, w, fast=True), x)


@pytest.mark.parametrize("N,bits", [(6, 12), (12, 31), (18, 31), (24, 31), (36, 45), (48, 60), (96, 60), (24, 60)])
def test_definition_vs_fast_large_primes(oracle, N, bits):
    q = find_prime_3n(N, bits)
    om = omega_for(q, N)
    rng = np.random.default_rng(N * 100 + bits)
    x = (rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(q)).astype(np.uint64)
    yf = oracle.ntt3n_forward(x, q, om, fast=True)
    yd = oracle.ntt3n_forward(x, q, om, fast=False)
    assert np.array_equal(yf, yd)
    assert np.array_equal(oracle.ntt3n_backward(yd, q, om, fast=False), x)     # Gaussian elimination path
    assert np.array_equal(oracle.ntt3n_backward(yd, q, om, fast=True), x)
    # a different primitive root gives a different (permuted) spectrum but the same round trip
    om2 = pow(om, 5, q)
    assert np.array_equal(oracle.ntt3n_backward(oracle.ntt3n_forward(x, q, om2), q, om2), x)


def naive_mul_3n(a, b, q, N):
    # naiveCyclotomicMultiply3N (ring/ntt_3n_test.go:312-364): schoolbook product reduced with X^N = X^(N/2) - 1
    full = [0] * (2 * N - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            full[i + j] = (full[i + j] + x * y) % q
    for d in range(2 * N - 2, N - 1, -1):
        c = full[d]
        if c:
            full[d] = 0
            full[d - N // 2] = (full[d - N // 2] + c) % q
            full[d - N] = (full[d - N] - c) % q
    return full[:N]


@pytest.mark.parametrize("N", [12, 24, 48, 96, 768])
def test_multiplication_vs_naive(oracle, N):
    q = find_prime_3n(N, 60)
    om = omega_for(q, N)
    rng = np.random.default_rng(N)
    a = [int(v) for v in rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(q)]
    b = [int(v) for v in rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(q)]
    A, B = oracle.ntt3n_forward(a, q, om), oracle.ntt3n_forward(b, q, om)
    C = np.array([(int(x) * int(y)) % q for x, y in zip(A, B)], dtype=np.uint64)
    c = oracle.ntt3n_backward(C, q, om)
    assert [int(v) for v in c] == naive_mul_3n(a, b, q, N)
