"""GPU: edge cases of the C ABI -- extreme ring sizes, odd limb counts, empty batches, aliasing, levels."""
import numpy as np
import pytest

from conftest import QI60, uniform_mod

pytestmark = pytest.mark.gpu


def test_largest_ring_2_17(rh, oracle):
    # N = 2^17 (Qi60 are valid up to 2^17: ring/test_params.go:14): five column stages per thread
    N, mods = 1 << 17, QI60[:2]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(17)
    a = np.stack([uniform_mod(rng, q, N) for q in mods])[None]
    p = rh.DevicePoly.from_numpy(ring, a)
    o = ring.NewPoly(1)
    ring.NTT(p, o)
    srs = [oracle.SubRingConsts(N, q) for q in mods]
    assert np.array_equal(o.numpy()[0], np.stack([oracle.ntt(a[0, i], srs[i]) for i in range(2)]))
    ring.NTTLazy(p, o)
    assert np.array_equal(o.numpy()[0, 1], oracle.ntt(a[0, 1], srs[1], lazy=True))
    ring.NTT(p, p); ring.INTT(p, p)
    assert np.array_equal(p.numpy(), a)
    ring.close()


@pytest.mark.parametrize("L", [1, 3, 7, 13])
def test_limb_counts_not_multiple_of_eight(rh, oracle, L):
    N, mods = 8192, QI60[:L]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(L)
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(3)])
    p = rh.DevicePoly.from_numpy(ring, a)
    ring.NTT(p, p)
    got = p.numpy()
    for i in (0, L - 1):
        assert np.array_equal(got[2, i], oracle.ntt(a[2, i], oracle.SubRingConsts(N, mods[i])))
    ring.INTT(p, p)
    assert np.array_equal(p.numpy(), a)
    ring.close()


def test_empty_batch_and_bad_arguments(rh):
    ring = rh.Ring(4096, QI60[:2])
    p = ring.NewPoly(1)
    lib = rh.lib()
    assert lib.rh_ring_ntt(ring._h, p.ptr, p.ptr, 0, 1, 0) == 0            # npoly = 0 is a no-op
    assert lib.rh_ring_vec_op(ring._h, 0, p.ptr, p.ptr, p.ptr, 0, 1, None, None) == 0
    assert lib.rh_ring_ntt(ring._h, p.ptr, p.ptr, 1, 2, 0) == -1           # level out of range
    assert lib.rh_ring_ntt(ring._h, p.ptr, p.ptr, -1, 1, 0) == -1
    assert lib.rh_ring_vec_op(ring._h, 99, p.ptr, p.ptr, p.ptr, 1, 1, None, None) == -1
    assert lib.rh_ring_vec_op(ring._h, rh.OPS["ADD"], p.ptr, None, p.ptr, 1, 1, None, None) == -1     # missing operand
    assert lib.rh_ring_set_tuning(ring._h, b"no_such_knob", 1) == -1
    ring.close()


def test_modulus_range_is_enforced(rh):
    with pytest.raises(rh.RingHipError):                       # q must be < 2^61 (lazy ranges reach 8q: ring/ntt.go:169)
        rh.Ring(64, [(1 << 62) + 1], constants=dict(mred=[1], bred=[[0, 0]], ninv=[1],
                                                     roots_fwd=np.zeros((1, 64), dtype=np.uint64), roots_bwd=np.zeros((1, 64), dtype=np.uint64)))


def test_small_moduli_and_mixed_sizes(rh, oracle):
    # 31-bit and 45-bit primes next to a 61-bit one: every kernel must be generic in q
    N = 4096
    mods = []
    for bits in (31, 45, 58):
        q = (1 << bits) + 1
        while not (oracle.lib().orc_is_prime(q) and q % (2 * N) == 1):
            q += 2 * N if q % (2 * N) == 1 else 1
        mods.append(q)
    mods.append(QI60[3])
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(3)
    a = np.stack([uniform_mod(rng, q, N) for q in mods])[None]
    p = rh.DevicePoly.from_numpy(ring, a)
    o = ring.NewPoly(1)
    ring.NTT(p, o)
    srs = [oracle.SubRingConsts(N, q) for q in mods]
    assert np.array_equal(o.numpy()[0], np.stack([oracle.ntt(a[0, i], srs[i]) for i in range(4)]))
    ring.NTTLazy(p, o)
    assert np.array_equal(o.numpy()[0], np.stack([oracle.ntt(a[0, i], srs[i], lazy=True) for i in range(4)]))
    ring.INTT(o, o)            # lazy representatives (< 6q) are accepted by INTT only up to 4q: reduce first
    ring.NTT(p, o); ring.INTT(o, o)
    assert np.array_equal(o.numpy(), a)
    ring.close()


def test_thirty_two_limbs_and_widest_basis_extension(rh, oracle):
    # the reference's largest basis: 32 source limbs (stack arrays of reconstructRNS, ring/basis_extension.go:285); all 32 Qi60
    # primes in one ring: NTT round trip + one limb vs the oracle, ModUpQtoP from 32 limbs (the bounded-unroll kernel variant)
    # and ModUpPtoQ to 32 limbs against the oracle, unreduced values bit for bit
    from conftest import QI60, PI60
    N, B = 1 << 13, 2
    Q, P = QI60[:32], PI60[:3]
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    rng = np.random.default_rng(32)
    a = np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(q) for q in Q]) for _ in range(B)])
    p = rh.DevicePoly.from_numpy(rq, a)
    f = rq.NewPoly(B)
    rq.NTT(p, f)
    sr = oracle.SubRingConsts(N, Q[31])
    assert np.array_equal(f.numpy()[1, 31], oracle.ntt(a[1, 31], sr))
    rq.INTT(f, f)
    assert np.array_equal(f.numpy(), a)
    be = rh.BasisExtender(rq, rp)
    outp = rp.NewPoly(B)
    be.ModUpQtoP(31, 2, p, outp)
    for k in range(B):
        assert np.array_equal(outp.numpy()[k], oracle.modup_centered(a[k], Q, P))
    b = np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(m) for m in P]) for _ in range(B)])
    outq = rq.NewPoly(B)
    be.ModUpPtoQ(2, 31, rh.DevicePoly.from_numpy(rp, b), outq)
    for k in range(B):
        assert np.array_equal(outq.numpy()[k], oracle.modup_centered(b[k], P, Q))
    # 12 source limbs: the 16-bounded kernel variant (levels below the top, polys allocated at that level)
    p12 = rh.DevicePoly.from_numpy(rq.AtLevel(11), np.ascontiguousarray(a[:, :12]))
    be.ModUpQtoP(11, 2, p12, outp)
    for k in range(B):
        assert np.array_equal(outp.numpy()[k], oracle.modup_centered(a[k, :12], Q[:12], P))
    be.close(); rq.close(); rp.close()


def test_bigint_converters_and_equal(rh):
    """ring/ring.go:433-558 (SetCoefficientsBigint, PolyToBigint, PolyToBigintCentered, Equal): host-side big-integer converters around
    device blocks; the method of ring/ring_test.go:186-240 -- set, transform there and back, reconstruct, compare with the integers"""
    N, mods = 64, QI60[:3]
    ring = rh.Ring(N, mods)
    Q = 1
    for q in mods:
        Q *= int(q)
    rng = np.random.default_rng(3)
    vals = [int.from_bytes(rng.bytes(24), "little") % Q - Q // 2 for _ in range(N)]             # signed, up to |Q|/2
    p = ring.NewPoly(2)
    ring.SetCoefficientsBigint(vals, p, poly=1)
    host = p.numpy()
    for i, q in enumerate(mods):
        assert [int(x) for x in host[1, i]] == [v % int(q) for v in vals]
    ring.NTT(p, p); ring.INTT(p, p)
    assert ring.PolyToBigint(p, poly=1) == [v % Q for v in vals]
    assert ring.PolyToBigint(p, gap=4, poly=1) == [v % Q for v in vals[::4]]
    cen = ring.PolyToBigintCentered(p, poly=1)
    assert cen == [(v % Q) - Q if (v % Q) >= Q >> 1 else v % Q for v in vals]
    # at a lower level the reconstruction is modulo the shorter chain
    v1 = ring.AtLevel(1)
    q01 = int(mods[0]) * int(mods[1])
    p1 = rh.DevicePoly.from_numpy(v1, host[1:2, :2].copy())
    assert v1.PolyToBigint(p1) == [v % q01 for v in vals]
    # Equal reduces in place first: a lazy representative equals its canonical one
    host[0] = host[1][:, ::-1]                                       # poly 0 was never written: give it canonical content
    a = rh.DevicePoly.from_numpy(ring, host)
    lazy = host.copy(); lazy[:, 0] += np.uint64(mods[0])
    b = rh.DevicePoly.from_numpy(ring, lazy)
    assert ring.Equal(a, b) and np.array_equal(b.numpy(), host)
    lazy[0, 2, 5] ^= np.uint64(1)
    assert not ring.Equal(a, rh.DevicePoly.from_numpy(ring, lazy))
    ring.close()
