"""GPU: the whole-Poly host entry rh_ntt_poly_forward / _backward -- Ring.NTT(p1, p2 Poly) as the reference's callers issue it
(ring/ntt.go:127-152: a loop over Poly.Coeffs [][]uint64) in ONE call through the C ABI: L host limb pointers, pipelined upload /
transform / download, one synchronisation.  Bit-exact against the reference's KATs (ring/ntt_test.go:10-89), the oracle, and the
per-limb NumberTheoreticTransformer seam (rh_ntt_forward ...), for pageable and page-locked limbs, all three ring types."""
import json
import os
import threading

import numpy as np
import pytest

from conftest import QI60, uniform_mod

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ntt_kat.json")))["vectors"]


def limbs(arr):
    """(L, N) array -> Poly.Coeffs: one separately allocated slice per limb, as Go's [][]uint64 is"""
    return [np.array(row, dtype=np.uint64).copy() for row in arr]


@pytest.mark.parametrize("vec", KAT, ids=lambda v: "N=%d" % v["N"])
def test_kat_whole_poly_entry(rh, vec):
    ring = rh.Ring(vec["N"], vec["Qis"])
    a = limbs(vec["poly"])
    want = np.array(vec["polyNTT"], dtype=np.uint64)
    out = [np.zeros(vec["N"], dtype=np.uint64) for _ in a]
    ring.NTTHost(a, out)
    assert np.array_equal(np.stack(out), want)
    assert np.array_equal(np.stack(a), np.array(vec["poly"], dtype=np.uint64))      # input untouched out of place
    ring.INTTHost(out, out)                                                          # in place, as ring/ntt_benchmark_test.go:42 does
    assert np.array_equal(np.stack(out), np.array(vec["poly"], dtype=np.uint64))
    ring.close()


def test_config1_shape_host_pointers(rh, oracle):
    # BASELINE config 1: ring/ntt_test.go power-of-two NTT, N = 2^12, single 60-bit prime, host slices through the seam
    N, q = 4096, QI60[0]
    ring = rh.Ring(N, [q])
    sr = oracle.SubRingConsts(N, q)
    rng = np.random.default_rng(12)
    a = uniform_mod(rng, q, N)
    a[:3] = [0, q - 1, 1]
    want, want_lazy = oracle.ntt(a, sr), oracle.ntt(a, sr, lazy=True)
    p1, p2 = [a.copy()], [np.zeros(N, dtype=np.uint64)]
    ring.NTTHost(p1, p2)
    assert np.array_equal(p2[0], want)
    assert np.array_equal(ring.SubRings[0].NTT(a), want)                             # the per-limb seam: same bits
    ring.NTTLazyHost(p1, p2)
    assert np.array_equal(p2[0], want_lazy)                                          # the reference's lazy representatives, exactly
    ring.INTTHost([want.copy()], p2)
    assert np.array_equal(p2[0], a)
    ring.INTTLazyHost([want.copy()], p2)
    assert np.array_equal(p2[0], a)
    with pytest.raises(rh.RingHipError):
        ring.NTTHost([a[:N - 1].copy()], p2)                                         # short slice: the reference panics (ring/ntt.go:212-214)
    with pytest.raises(rh.RingHipError):
        ring.NTTHost([], p2)
    # a device pointer handed to the HOST entry is refused, not dereferenced on the CPU
    import ctypes as C
    dp = rh.DevicePoly.from_numpy(ring, a[None, None])
    ins, outs = (C.c_void_p * 1)(dp.ptr), (C.c_void_p * 1)(p2[0].ctypes.data)
    assert rh.lib().rh_ntt_poly_forward(ring._h, 0, ins, outs, 0) == -1 and b"device pointer" in rh.lib().rh_last_error()
    ring.close()


@pytest.mark.parametrize("logN,L", [(8, 3), (12, 5), (13, 4), (14, 16), (16, 16), (16, 3), (17, 2)])
def test_whole_poly_vs_oracle_pageable_and_pinned(rh, oracle, logN, L):
    N, mods = 1 << logN, QI60[:L]
    ring = rh.Ring(N, mods)
    srs = [oracle.SubRingConsts(N, q) for q in mods]
    rng = np.random.default_rng(logN * 100 + L)
    a = np.stack([uniform_mod(rng, q, N) for q in mods])
    want = np.stack([oracle.ntt(a[i], srs[i]) for i in range(L)])
    # pageable limbs, out of place
    out = [np.zeros(N, dtype=np.uint64) for _ in range(L)]
    ring.NTTHost(limbs(a), out)
    assert np.array_equal(np.stack(out), want)
    # page-locked limbs (rh_host_alloc): DMA'd where they lie, in place
    pin = rh.PinnedBuffer((L, N))
    pin.array[:] = a
    rows = [pin.array[i] for i in range(L)]
    ring.NTTHost(rows, rows)
    assert np.array_equal(pin.array, want)
    ring.INTTHost(rows, rows)
    assert np.array_equal(pin.array, a)
    # mixed: pinned input, pageable output; and a view at a lower level touches only limbs 0..level
    lvl = L // 2
    out2 = [np.full(N, 7, dtype=np.uint64) for _ in range(L)]
    ring.AtLevel(lvl).NTTHost(rows, out2)
    assert np.array_equal(np.stack(out2[:lvl + 1]), want[:lvl + 1])
    assert all(np.all(o == 7) for o in out2[lvl + 1:])
    # lazy forward: the same representatives as the per-limb seam
    ring.NTTLazyHost(limbs(a), out)
    for i in (0, L - 1):
        assert np.array_equal(out[i], ring.SubRings[i].NTTLazy(a[i]))
    # a registered (rh_host_register) ordinary allocation: whole pages of its own (an anonymous mapping, as a Go runtime hands out large slices) --
    # registering part of a page pins bytes of unrelated heap objects with it (ringhip.h)
    import mmap
    nbytes = -(-(L * N * 8) // 4096) * 4096
    mm = mmap.mmap(-1, nbytes)
    own = np.frombuffer(mm, dtype=np.uint64, count=L * N).reshape(L, N)
    own[:] = a
    assert rh.lib().rh_host_register(own.ctypes.data + 8, nbytes // 8) == -1 and b"whole 4 KiB pages" in rh.lib().rh_last_error()   # not in the middle of a page
    assert rh.lib().rh_host_register(own.ctypes.data, nbytes // 8) == 0
    try:
        r2 = [own[i] for i in range(L)]
        ring.NTTHost(r2, r2)
        assert np.array_equal(own, want)
    finally:
        assert rh.lib().rh_host_unregister(own.ctypes.data) == 0
    del r2, own
    mm.close()
    pin.free()
    ring.close()


def test_whole_poly_conjugate_invariant_and_3n(rh, oracle):
    from test_oracle_ntt3n import find_prime_3n

    def primes_3n(N3, count):
        out = [find_prime_3n(N3, 60)]
        while len(out) < count:
            q = out[-1] + 3 * N3
            while not oracle.lib().orc_is_prime(q):
                q += 3 * N3
            out.append(q)
        return out
    N = 1 << 14
    ci = rh.Ring(N, QI60[:3], kind=rh.ConjugateInvariant)
    rng = np.random.default_rng(5)
    a = np.stack([uniform_mod(rng, q, N) for q in QI60[:3]])
    out = [np.zeros(N, dtype=np.uint64) for _ in range(3)]
    ci.NTTHost(limbs(a), out)
    for i in range(3):
        assert np.array_equal(out[i], ci.SubRings[i].NTT(a[i]))
    ci.INTTHost(out, out)
    assert np.array_equal(np.stack(out), a)
    ci.close()
    for N3 in (3 << 6, 3 << 13):
        mods = primes_3n(N3, 5)
        r3 = rh.Ring(N3, mods, kind=rh.Matrix3N)
        a = np.stack([uniform_mod(rng, q, N3) for q in mods])
        out = [np.zeros(N3, dtype=np.uint64) for _ in mods]
        r3.NTTHost(limbs(a), out)                                       # five limbs in groups over two streams, each with its own workspace
        for i in range(len(mods)):
            assert np.array_equal(out[i], r3.SubRings[i].NTT(a[i])), (N3, i)
        r3.INTTHost(out, out)
        assert np.array_equal(np.stack(out), a)
        r3.close()


def test_whole_poly_concurrent_callers(rh, oracle):
    # goroutines share one Ring (ring/ring.go:192-194): four host threads, each with its own polys, one handle
    N, L = 1 << 13, 4
    mods = QI60[:L]
    ring = rh.Ring(N, mods)
    srs = [oracle.SubRingConsts(N, q) for q in mods]
    errs = []

    def worker(seed):
        try:
            rng = np.random.default_rng(seed)
            for _ in range(6):
                a = np.stack([uniform_mod(rng, q, N) for q in mods])
                out = [np.zeros(N, dtype=np.uint64) for _ in range(L)]
                ring.NTTHost(limbs(a), out)
                i = int(rng.integers(0, L))
                if not np.array_equal(out[i], oracle.ntt(a[i], srs[i])):
                    errs.append("mismatch seed %d" % seed)
                ring.INTTHost(out, out)
                if not np.array_equal(np.stack(out), a):
                    errs.append("round trip seed %d" % seed)
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    ts = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    ring.close()
