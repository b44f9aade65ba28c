"""GPU: one ring handle used by several host threads at once (SURVEY 8(b) "Threading"; the reference's transformers are
immutable and Ring.AtLevel views are concurrency-safe, ring/ring.go:192-194; goroutines migrate between OS threads).
ctypes releases the GIL around every call, so the C entry points really overlap."""
import threading

import numpy as np
import pytest

from conftest import QI60, uniform_mod

pytestmark = pytest.mark.gpu


def _run_threads(fns):
    errs = []

    def wrap(f):
        def g():
            try:
                f()
            except BaseException as e:          # noqa: BLE001 -- reported below
                errs.append(e)
        return g
    ts = [threading.Thread(target=wrap(f)) for f in fns]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errs, errs
    assert all(not t.is_alive() for t in ts)


@pytest.mark.parametrize("kind,N", [("std", 1 << 14), ("std", 256), ("ci", 4096), ("3n", 3 << 11)])
def test_four_threads_hammer_one_handle(rh, oracle, kind, N):
    # 4 host threads on ONE handle: three call the per-limb transformer interface (rh_ntt_forward / backward, host slices) on
    # different limbs -- and two of them on the SAME limb -- while the fourth runs the device-batched rh_ring_ntt / intt.
    # Every result is compared with the oracle (or the round trip for the conjugate-invariant ring, pinned elsewhere).
    from test_oracle_ntt3n import find_prime_3n, omega_for
    rng = np.random.default_rng(N)
    if kind == "3n":
        q = find_prime_3n(N, 60); mods = []
        while len(mods) < 3:
            if oracle.lib().orc_is_prime(q):
                mods.append(q)
            q += 3 * N
        om = [omega_for(m, N) for m in mods]
        ring = rh.Ring(N, mods, kind=rh.Matrix3N, omega3n=om)
        fwd = lambda x, i: oracle.ntt3n_forward(x, mods[i], om[i])
    elif kind == "ci":
        mods = QI60[:3]
        ring = rh.Ring(N, mods, kind=rh.ConjugateInvariant)
        srs = [oracle.SubRingConsts(N, m, nthroot=4 * N) for m in mods]
        fwd = lambda x, i: oracle.ntt_ci(x, srs[i])
    else:
        mods = QI60[:3]
        ring = rh.Ring(N, mods)
        srs = [oracle.SubRingConsts(N, m) for m in mods]
        fwd = lambda x, i: oracle.ntt(x, srs[i])
    reps = 12
    xs = {t: [uniform_mod(rng, mods[l], N) for _ in range(reps)] for t, l in ((0, 0), (1, 1), (2, 1))}
    got = {t: [] for t in xs}
    B = 5
    batch = np.stack([np.stack([uniform_mod(rng, m, N) for m in mods]) for _ in range(B)])
    dev = rh.DevicePoly.from_numpy(ring, batch)
    out = ring.NewPoly(B)
    batched = []

    def limb_worker(t, limb):
        def f():
            sr = ring.SubRings[limb]
            for x in xs[t]:
                y = sr.NTT(x)
                got[t].append((y, sr.INTT(y)))
        return f

    def batch_worker():
        for _ in range(reps):
            ring.NTT(dev, out)
            ring.sync()
            batched.append(out.numpy())
            ring.INTT(out, out)
            ring.sync()
            batched.append(out.numpy())

    _run_threads([limb_worker(0, 0), limb_worker(1, 1), limb_worker(2, 1), batch_worker])
    for t, limb in ((0, 0), (1, 1), (2, 1)):
        assert len(got[t]) == reps
        for x, (y, back) in zip(xs[t], got[t]):
            assert np.array_equal(y, fwd(x, limb)), (kind, t)
            assert np.array_equal(back, x)
    exp = np.stack([np.stack([fwd(batch[k, i], i) for i in range(len(mods))]) for k in range(B)])
    for j in range(reps):
        assert np.array_equal(batched[2 * j], exp)
        assert np.array_equal(batched[2 * j + 1], batch)
    ring.close()


def test_two_basis_extenders_share_rings_across_threads(rh, oracle):
    # the reference's pattern: one BasisExtender (ShallowCopy) per goroutine over the SAME rings (ring/basis_extension.go:166-183)
    N, Q, P = 4096, QI60[:5], QI60[8:10]
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    rng = np.random.default_rng(5)
    cases = []
    for _ in range(2):
        a = np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(3)])
        cases.append((a, rh.BasisExtender(rq, rp)))
    res = [None, None]

    def worker(k):
        def f():
            a, be = cases[k]
            pq = rh.DevicePoly.from_numpy(rq, a)
            pp = rh.DevicePoly(rp, 3, len(P))
            outs = []
            for _ in range(6):
                be.ModUpQtoP(len(Q) - 1, len(P) - 1, pq, pp)
                rq.sync()
                outs.append(pp.numpy())
            res[k] = outs
        return f
    _run_threads([worker(0), worker(1)])
    for k in range(2):
        a = cases[k][0]
        exp = np.stack([oracle.modup_centered(a[j], Q, P) for j in range(3)])
        for o in res[k]:
            assert np.array_equal(o, exp)
    for _, be in cases:
        be.close()
    rq.close(); rp.close()


def test_n8_ring_matches_reference_small_degree_rules(rh, oracle):
    # N = 8 is a valid ring degree (MinimumRingDegreeForLoopUnrolledOperations, ring/ring.go:21-23, :318).  For N < 16 every
    # forward stage reduces (ring/ntt.go:223-257) and BackwardLazy ends in MRedLazy: values in [0, 2q), NOT canonical (:197-202)
    N, mods = 8, QI60[:2]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(8)
    srs = [oracle.SubRingConsts(N, q) for q in mods]
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(3)])
    a[0, :, 0] = 0; a[0, :, 1] = np.array(mods, dtype=np.uint64) - np.uint64(1)
    p, o = rh.DevicePoly.from_numpy(ring, a), ring.NewPoly(3)
    ring.NTT(p, o)
    f = o.numpy()
    for k in range(3):
        for i in range(2):
            assert np.array_equal(f[k, i], oracle.ntt(a[k, i], srs[i]))
    ring.NTTLazy(p, o)
    for k in range(3):
        for i in range(2):
            assert np.array_equal(o.numpy()[k, i], oracle.ntt(a[k, i], srs[i], lazy=True))
    pf = rh.DevicePoly.from_numpy(ring, f)
    ring.INTT(pf, o)
    assert np.array_equal(o.numpy(), a)
    ring.INTTLazy(pf, o)
    lz = o.numpy()
    noncanonical = False
    for k in range(3):
        for i in range(2):
            e = oracle.intt(f[k, i], srs[i], lazy=True)
            assert np.array_equal(lz[k, i], e)
            noncanonical |= bool((e >= np.uint64(mods[i])).any())
            assert np.array_equal(e % np.uint64(mods[i]), a[k, i])
    for i in range(2):                                              # and through the per-limb interface
        assert np.array_equal(ring.SubRings[i].INTTLazy(f[1, i]), oracle.intt(f[1, i], srs[i], lazy=True))
        assert np.array_equal(ring.SubRings[i].NTT(a[1, i]), f[1, i])
    ring.close()
    with pytest.raises(rh.RingHipError):
        rh.Ring(4, mods)


def test_reserve_then_no_allocation_paths_still_exact(rh, oracle):
    # rh_ring_reserve / rh_bext_reserve pre-size the scratch; results are unchanged
    N, Q, P = 8192, QI60[:4], QI60[6:8]
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    rq.reserve(3); rp.reserve(3)
    be = rh.BasisExtender(rq, rp)
    be.reserve(3)
    rng = np.random.default_rng(3)
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(3)])
    p = rh.DevicePoly.from_numpy(rq, a)
    o = rh.DevicePoly(rq, 3, len(Q) - 1)
    rq.DivRoundByLastModulusNTT(p, o)
    srs = [oracle.SubRingConsts(N, q) for q in Q]
    coeff = np.stack([oracle.intt(a[1, i], srs[i]) for i in range(len(Q))])
    down = oracle.div_by_last_modulus_many(coeff, Q, 1, True)
    assert np.array_equal(o.numpy()[1], np.stack([oracle.ntt(down[i], srs[i]) for i in range(len(Q) - 1)]))
    be.close(); rq.close(); rp.close()
