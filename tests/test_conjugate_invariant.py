"""Conjugate-invariant NTT (ring/ntt.go:716-1311).  CPU: the oracle against the reference's own cross-check
(ring/ring_test.go:85-126: squaring in Z[X+X^-1]/(X^2N+1) == squaring the symmetric extension in the standard 2N ring);
GPU: bit-exact against the oracle."""
import numpy as np
import pytest

from conftest import QI60, uniform_mod


def ci_square_via_standard(oracle, a, N, q):
    sr2 = oracle.SubRingConsts(2 * N, q)
    ext = np.zeros(2 * N, dtype=np.uint64)
    ext[:N] = a
    for j in range(1, N):
        ext[2 * N - j] = np.uint64(q) - a[j] if a[j] else 0
    y = oracle.ntt(ext, sr2)
    y = oracle.vec_op(33, y, None, y, 0, 0, q)            # MForm
    y = oracle.vec_op(13, y, y, y, 0, 0, q)               # MulCoeffsMontgomery
    y = oracle.vec_op(35, y, None, y, 0, 0, q)            # IMForm
    return oracle.intt(y, sr2)[:N]


@pytest.mark.parametrize("logN", [4, 5, 8, 10])
def test_oracle_ci_matches_standard_ring(oracle, logN):
    N, q = 1 << logN, QI60[0]
    sr = oracle.SubRingConsts(N, q, nthroot=4 * N)
    rng = np.random.default_rng(logN)
    a = uniform_mod(rng, q, N)
    y = oracle.ntt_ci(a, sr)
    assert np.array_equal(oracle.intt_ci(y, sr), a)
    z = oracle.vec_op(33, y, None, y, 0, 0, q)
    z = oracle.vec_op(13, z, z, z, 0, 0, q)
    z = oracle.vec_op(35, z, None, z, 0, 0, q)
    assert np.array_equal(oracle.intt_ci(z, sr), ci_square_via_standard(oracle, a, N, q))


@pytest.mark.gpu
# logN >= 14: the fold runs inside the hand-scheduled column stages (a thread owns the column pair the fold couples, column 0 apart)
@pytest.mark.parametrize("logN", [4, 8, 12, 13, 14, 15, 16])
def test_gpu_ci_vs_oracle(rh, oracle, logN):
    N, mods = 1 << logN, QI60[:3]
    ring = rh.Ring(N, mods, kind=rh.ConjugateInvariant)
    srs = [oracle.SubRingConsts(N, q, nthroot=4 * N) for q in mods]
    c = ring.constants()
    for i, s in enumerate(srs):                                   # engine-generated 4N-th-root tables == oracle's
        assert np.array_equal(c["roots_fwd"][i], s.roots_fwd) and int(c["ninv"][i]) == s.ninv
    rng = np.random.default_rng(logN)
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(2)])
    p, o = rh.DevicePoly.from_numpy(ring, a), ring.NewPoly(2)
    ring.NTT(p, o)
    exp = np.stack([np.stack([oracle.ntt_ci(a[k, i], srs[i]) for i in range(3)]) for k in range(2)])
    assert np.array_equal(o.numpy(), exp)
    ring.NTTLazy(p, o)                                            # lazy contract: congruent, inside [0, 6q-2]
    assert np.array_equal(o.numpy() % np.array(mods, dtype=np.uint64)[None, :, None], exp)
    ring.INTT(o, o)
    assert np.array_equal(o.numpy(), a)
    ring.NTT(p, p)                                                # in place, then back out of place
    assert np.array_equal(p.numpy(), exp)
    ring.INTT(p, o)
    assert np.array_equal(o.numpy(), a) and np.array_equal(p.numpy(), exp)
    ring.set_tuning("fuse_ci", 0)                                 # the fold as a pass of its own: same bits
    ring.INTT(p, p)
    assert np.array_equal(p.numpy(), a)
    ring.NTT(p, p)
    assert np.array_equal(p.numpy(), exp)
    ring.set_tuning("fuse_ci", 1)
    assert np.array_equal(ring.SubRings[1].NTT(a[0, 1]), exp[0, 1])
    assert np.array_equal(ring.SubRings[2].INTT(exp[1, 2]), a[1, 2])
    ring.close()


@pytest.mark.gpu
@pytest.mark.parametrize("logN,L,B", [(14, 3, 700), (15, 2, 2100), (16, 16, 140)])
def test_gpu_ci_large_batches_run_the_fused_pipeline(rh, oracle, logN, L, B):
    # batches of more than ~2048 rows: forward launch j = fold + column stages of span j with the tile stages of span j-1, inverse launch j = tile
    # stages of span j with the column stages + fold of span j-1 (round 3) -- bit-identical to the two-launch path (chunk_polys = 0) and, on
    # spot rows of the first, a middle and the last span, to the oracle; in place and out of place
    import torch
    N = 1 << logN
    mods = [q for q in QI60 if (q - 1) % (4 * N) == 0][:L]
    ring = rh.Ring(N, mods, kind=rh.ConjugateInvariant)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(logN * 7 + B)
    qs = torch.tensor(mods, dtype=torch.int64, device=dev).view(1, L, 1)
    x = torch.randint(0, 1 << 62, (B, L, N), dtype=torch.int64, device=dev, generator=g) % qs
    dp = lambda t: rh.DevicePoly.from_torch(ring, t)
    y, y0 = torch.empty_like(x), torch.empty_like(x)
    ring.NTT(dp(x), dp(y))                                         # pipelined, out of place
    ring.set_tuning("chunk_polys", 0)
    ring.NTT(dp(x), dp(y0))                                        # two launches over the whole batch
    ring.set_tuning("chunk_polys", -1)
    torch.cuda.synchronize()
    assert torch.equal(y, y0)
    srs = [oracle.SubRingConsts(N, q, nthroot=4 * N) for q in mods]
    h = lambda t, k, i: t[k, i].cpu().numpy().view(np.uint64)
    for k in (0, B // 2 + 1, B - 1):
        i = k % L
        assert np.array_equal(h(y, k, i), oracle.ntt_ci(h(x, k, i), srs[i])), (k, i)
    z = y.clone()
    ring.INTT(dp(z), dp(z))                                        # pipelined, in place
    torch.cuda.synchronize()
    assert torch.equal(z, x)
    z2 = torch.empty_like(x)
    ring.set_tuning("chunk_polys", 0)
    ring.INTT(dp(y), dp(z2))
    torch.cuda.synchronize()
    assert torch.equal(z2, x)
    ring.close()
