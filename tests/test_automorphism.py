"""Automorphisms X -> X^gen (ring/automorphism.go).  CPU: the oracle against the algebraic definition
(sigma(f)(X) = f(X^gen) mod X^N+1, and NTT(sigma(f)) = permuted NTT(f)); GPU: bit-exact against the oracle."""
import numpy as np
import pytest

from conftest import QI60, uniform_mod


def sigma_naive(a, gen, q, N):
    out = [0] * N
    for i, c in enumerate(a):
        e = (i * gen) % (2 * N)
        if e < N:
            out[e] = (out[e] + c) % q
        else:
            out[e - N] = (out[e - N] - c) % q
    return out


@pytest.mark.parametrize("N,gen", [(16, 5), (64, 25), (64, 127), (256, 3), (1024, 5 ** 7 % 2048)])
def test_oracle_automorphism_definition(oracle, N, gen):
    q = QI60[0]
    rng = np.random.default_rng(N + gen)
    a = uniform_mod(rng, q, N)
    s = oracle.automorphism(a, gen, q)
    assert [int(v) for v in s] == sigma_naive([int(v) for v in a], gen, q, N)
    # NTT-domain form: permuting NTT(a) equals NTT(sigma(a))
    sr = oracle.SubRingConsts(N, q)
    assert np.array_equal(oracle.automorphism_ntt(oracle.ntt(a, sr), gen), oracle.ntt(s, sr))


@pytest.mark.gpu
@pytest.mark.parametrize("logN,gen", [(4, 5), (12, 5 ** 3), (16, 5 ** 11 % (1 << 17)), (13, (1 << 14) - 1)])
def test_gpu_automorphisms_vs_oracle(rh, oracle, logN, gen):
    N, mods = 1 << logN, QI60[:3]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(logN)
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(2)])
    acc = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(2)])
    p, o = rh.DevicePoly.from_numpy(ring, a), ring.NewPoly(2)
    ring.AutomorphismNTT(p, gen, o)
    got = o.numpy()
    for k in range(2):
        for i in range(3):
            assert np.array_equal(got[k, i], oracle.automorphism_ntt(a[k, i], gen))
    pa = rh.DevicePoly.from_numpy(ring, acc)
    ring.AutomorphismNTTThenAddLazy(p, gen, pa)
    assert np.array_equal(pa.numpy()[1, 2], oracle.automorphism_ntt(a[1, 2], gen, acc=acc[1, 2]))
    ring.Automorphism(p, gen, o)
    got = o.numpy()
    for i, q in enumerate(mods):
        assert np.array_equal(got[0, i], oracle.automorphism(a[0, i], gen, q))
    with pytest.raises(rh.RingHipError):
        ring.AutomorphismNTT(p, gen, p)          # "the result cannot be in-place"
    ring.close()
