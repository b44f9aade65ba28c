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


def _ci_extension(a, q, N):
    """coefficients of the element of Z[X]/(X^2N+1) a conjugate-invariant poly stands for (ring/ring_test.go:85-126)"""
    ext = np.zeros(2 * N, dtype=np.uint64)
    ext[:N] = a
    for j in range(1, N):
        ext[2 * N - j] = (q - int(a[j])) % q
    return ext


@pytest.mark.parametrize("N,gen", [(16, 5), (64, 25), (256, 5 ** 9 % 1024), (64, 4 * 64 - 3)])
def test_oracle_ci_automorphism_definition(oracle, N, gen):
    # conjugate-invariant branch of Ring.Automorphism (ring/automorphism.go:131-156): sigma acts on the symmetric extension in the
    # standard ring of degree 2N; the result is again symmetric and its first N coefficients are the CI representation.
    # NTT-domain form over NthRoot = 4N: permuting NTT_ci(a) equals NTT_ci(sigma(a))  (gen = 1 mod 4)
    q = QI60[0]
    rng = np.random.default_rng(N * 3 + gen)
    a = uniform_mod(rng, q, N)
    a[3] = 0
    s = oracle.automorphism_ci(a, gen, q)
    ext = sigma_naive([int(v) for v in _ci_extension(a, q, N)], gen, q, 2 * N)
    assert [int(v) % q for v in s] == ext[:N]
    assert [int(v) for v in _ci_extension(np.array(ext[:N], dtype=np.uint64), q, N)] == ext      # still symmetric
    sr = oracle.SubRingConsts(N, q, nthroot=4 * N)
    red = np.array([int(v) % q for v in s], dtype=np.uint64)
    assert np.array_equal(oracle.automorphism_ntt_ci(oracle.ntt_ci(a, sr), gen), oracle.ntt_ci(red, sr))


@pytest.mark.gpu
@pytest.mark.parametrize("logN,gen", [(4, 5), (12, 5 ** 5), (14, 5 ** 13 % (1 << 16))])
def test_gpu_ci_automorphisms_vs_oracle(rh, oracle, logN, gen):
    N, mods = 1 << logN, QI60[:3]
    ring = rh.Ring(N, mods, kind=rh.ConjugateInvariant)
    rng = np.random.default_rng(logN + 40)
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(2)])
    a[0, :, 5] = 0                                                # -0 comes out as q, like the reference's select formula
    acc = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(2)])
    p, o = rh.DevicePoly.from_numpy(ring, a), ring.NewPoly(2)
    ring.Automorphism(p, gen, o)
    got = o.numpy()
    for k in range(2):
        for i, q in enumerate(mods):
            assert np.array_equal(got[k, i], oracle.automorphism_ci(a[k, i], gen, q))
    ring.AutomorphismNTT(p, gen, o)
    assert np.array_equal(o.numpy()[1, 2], oracle.automorphism_ntt_ci(a[1, 2], gen))
    pa = rh.DevicePoly.from_numpy(ring, acc)
    ring.AutomorphismNTTThenAddLazy(p, gen, pa)
    assert np.array_equal(pa.numpy()[0, 1], oracle.automorphism_ntt_ci(a[0, 1], gen, acc=acc[0, 1]))
    with pytest.raises(rh.RingHipError):
        ring.AutomorphismNTT(p, 4 * N - 1, o)                    # gen = 3 mod 4: the reference's table look-up runs out of range
    ring.close()


@pytest.mark.gpu
def test_automorphism_with_index_table(rh, oracle):
    # AutomorphismNTTIndex (ring/automorphism.go:12-34) + AutomorphismNTTWithIndex / ...ThenAddLazy (:50-117) == AutomorphismNTT by generator
    N, mods = 256, QI60[:2]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(5)
    a = np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(q) for q in mods]) for _ in range(2)])
    pa, o1, o2 = rh.DevicePoly.from_numpy(ring, a), ring.NewPoly(2), ring.NewPoly(2)
    for gal in (5, 25, 2 * N - 1):
        idx = rh.AutomorphismNTTIndex(N, 2 * N, gal)
        ring.AutomorphismNTT(pa, gal, o1)
        ring.AutomorphismNTTWithIndex(pa, idx, o2)
        assert np.array_equal(o1.numpy(), o2.numpy())
        assert np.array_equal(o2.numpy()[1, 1], a[1, 1][idx.astype(np.int64)])
        assert np.array_equal(o2.numpy()[1, 0], oracle.automorphism_ntt(a[1, 0], gal))
        acc = rh.DevicePoly.from_numpy(ring, a)
        ring.AutomorphismNTTWithIndexThenAddLazy(pa, idx, acc)
        assert np.array_equal(acc.numpy(), a + o1.numpy())
    be = rh.BasisExtender(ring, rh.Ring(N, QI60[2:4]))
    be2 = be.ShallowCopy()
    assert be2._h != be._h
    be2.close(); be.close(); ring.close()
