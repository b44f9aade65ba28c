"""GPU: 3N-cyclotomic transform parity through the C ABI (NumberTheoreticTransformer3N semantics, ring/ntt_3n.go)."""
import json
import os

import numpy as np
import pytest

from test_oracle_ntt3n import VEC, find_prime_3n, omega_for, naive_mul_3n, rank

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("vec", VEC, ids=lambda v: "N=%d" % v["N"])
def test_reference_python_vectors(rh, vec):
    # omega handed over explicitly (the Go transformer draws it at random: SURVEY F4), 17-bit primes of the notes
    N, p, w = vec["N"], vec["p"], vec["w"]
    ring = rh.Ring(N, [p], kind=rh.Matrix3N, omega3n=[w])
    exp = np.zeros(N, dtype=np.uint64)
    for s, e in enumerate(vec["tree_last"]):
        exp[rank(e)] = vec["dft_tree_order"][s]
    y = ring.SubRings[0].NTT(vec["input"])
    assert np.array_equal(y, exp)
    assert np.array_equal(ring.SubRings[0].NTTLazy(vec["input"]), exp)          # ForwardLazy = Forward (ntt_3n.go:112-114)
    assert np.array_equal(ring.SubRings[0].INTT(y), np.array(vec["input"], dtype=np.uint64))
    assert np.array_equal(ring.SubRings[0].INTTLazy(y), np.array(vec["input"], dtype=np.uint64))
    ring.close()


@pytest.mark.parametrize("N", [6, 12, 18, 24, 36, 48, 54, 96, 192, 768, 3 * 1024, 3 * 4096, 3 * 8192, 9 * 4096])
def test_forward_backward_vs_oracle_60bit(rh, oracle, N):
    mods = []
    q = find_prime_3n(N, 60)
    mods.append(q)
    q2 = q + 3 * N
    while not oracle.lib().orc_is_prime(q2):
        q2 += 3 * N
    mods.append(q2)
    oms = [omega_for(m, N) for m in mods]
    ring = rh.Ring(N, mods, kind=rh.Matrix3N)          # engine derives omega = g^((q-1)/3N) like Find3NPrimitiveRoot
    assert [int(v) for v in ring.constants()["omega3n"]] == oms
    rng = np.random.default_rng(N)
    npoly = 2
    a = np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(m) for m in mods]) for _ in range(npoly)])
    a[0, :, 0] = 0
    a[0, :, 1] = np.array(mods, dtype=np.uint64) - np.uint64(1)
    p = rh.DevicePoly.from_numpy(ring, a)
    o = ring.NewPoly(npoly)
    ring.NTT(p, o)
    got = o.numpy()
    exp = np.stack([np.stack([oracle.ntt3n_forward(a[k, i], mods[i], oms[i]) for i in range(2)]) for k in range(npoly)])
    assert np.array_equal(got, exp)
    if N <= 768:
        assert np.array_equal(got[1, 0], oracle.ntt3n_forward(a[1, 0], mods[0], oms[0], fast=False))     # Horner definition
    else:
        E = oracle.ntt3n_exponents(3 * N)
        for k in (0, 1, N // 3, N - 1):
            x = pow(oms[1], E[k], mods[1]); acc = 0
            for cft in reversed(a[1, 1].tolist()):
                acc = (acc * x + int(cft)) % mods[1]
            assert int(got[1, 1, k]) == acc
    ring.INTT(o, o)                                      # in place
    assert np.array_equal(o.numpy(), a)
    ring.close()


def test_config2_size_multiplication(rh, oracle):
    # BASELINE config 2: N = 3*2^13, one 60-bit modulus.  Sparse product checked against the reduction rule
    # X^N = X^(N/2) - 1 (ring/ntt_3n_test.go:103-107), dense round trip.
    N = 3 * 8192
    q = find_prime_3n(N, 60)
    ring = rh.Ring(N, [q], kind=rh.Matrix3N)
    x = np.zeros((1, 1, N), dtype=np.uint64); y = np.zeros((1, 1, N), dtype=np.uint64)
    x[0, 0, N - 2] = 3; x[0, 0, 5] = 7
    y[0, 0, N // 2 + 4] = 11
    px, py = rh.DevicePoly.from_numpy(ring, x), rh.DevicePoly.from_numpy(ring, y)
    ring.NTT(px, px); ring.NTT(py, py)
    ring.MulCoeffsBarrett(px, py, px)
    ring.INTT(px, px)
    # 3*11*X^(3N/2+2) + 7*11*X^(N/2+9);  X^(3N/2+2) = X^(N/2+2) * X^N = X^(N/2+2)(X^(N/2) - 1) = X^(N+2) - X^(N/2+2)
    #   = X^2 (X^(N/2) - 1) - X^(N/2+2) = -X^2   (since X^(N/2+2) cancels)
    exp = np.zeros(N, dtype=np.uint64)
    exp[2] = (q - 33) % q
    exp[N // 2 + 9] = 77
    assert np.array_equal(px.numpy()[0, 0], exp)
    ring.close()


def test_config4_size_properties(rh, oracle):
    # config 4 ring: N = 3*2^16, 3 limbs here (24 in the config), batch 2: round trip + linearity + one limb vs oracle
    N = 3 << 16
    mods = []
    q = find_prime_3n(N, 60)
    for _ in range(3):
        mods.append(q)
        q += 3 * N
        while not oracle.lib().orc_is_prime(q):
            q += 3 * N
    ring = rh.Ring(N, mods, kind=rh.Matrix3N)
    rng = np.random.default_rng(4)
    a = np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(m) for m in mods]) for _ in range(2)])
    b = np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(m) for m in mods]) for _ in range(2)])
    pa, pb = rh.DevicePoly.from_numpy(ring, a), rh.DevicePoly.from_numpy(ring, b)
    na, nb_, ns = ring.NewPoly(2), ring.NewPoly(2), ring.NewPoly(2)
    ring.NTT(pa, na); ring.NTT(pb, nb_)
    ring.Add(pa, pb, ns); ring.NTT(ns, ns)
    ring.Add(na, nb_, nb_)
    assert np.array_equal(ns.numpy(), nb_.numpy())
    om = omega_for(mods[1], N)
    assert np.array_equal(na.numpy()[1, 1], oracle.ntt3n_forward(a[1, 1], mods[1], om))
    ring.INTT(na, na)
    assert np.array_equal(na.numpy(), a)
    ring.close()


def test_3n_errors(rh):
    with pytest.raises(rh.RingHipError):       # modulus without a primitive 3N-th root -> the Go ctor panics (ntt_3n.go:41)
        rh.Ring(24, [0x1fffffffffe00001 - 0], kind=rh.Matrix3N) if (0x1fffffffffe00001 - 1) % 72 else (_ for _ in ()).throw(rh.RingHipError("skip"))
    with pytest.raises(rh.RingHipError):
        rh.Ring(20, [65537], kind=rh.Matrix3N)   # N not of the form 2^a 3^b


@pytest.mark.parametrize("logn2", [13, 14, 15])
def test_fused_pre_and_column_stages_identical(rh, oracle, logn2):
    # b = 1 rings with n2 = 2^13..2^15: split + radix-3 layer fused with the sub-transforms' column stages (default) vs the
    # separate passes; forward and inverse, batch of 3 with 2 limbs, one limb against the oracle
    N = 6 << logn2
    mods = []
    q = find_prime_3n(N, 60)
    while len(mods) < 2:
        if oracle.lib().orc_is_prime(q):
            mods.append(q)
        q += 3 * N
    ring = rh.Ring(N, mods, kind=rh.Matrix3N, omega3n=[omega_for(m, N) for m in mods])
    rng = np.random.default_rng(logn2)
    a = np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(m) for m in mods]) for _ in range(3)])
    p = rh.DevicePoly.from_numpy(ring, a)
    f0, f1, b0, b1 = (ring.NewPoly(3) for _ in range(4))
    ring.set_tuning("fuse3n", 0); ring.NTT(p, f0); ring.INTT(f0, b0)
    ring.set_tuning("fuse3n", 1); ring.NTT(p, f1); ring.INTT(f1, b1)
    assert np.array_equal(f1.numpy(), f0.numpy())
    assert np.array_equal(b1.numpy(), a) and np.array_equal(b0.numpy(), a)
    if logn2 == 13:
        assert np.array_equal(f1.numpy()[2, 1], oracle.ntt3n_forward(a[2, 1], mods[1], omega_for(mods[1], N)))
    ring.NTT(p, p); ring.INTT(p, p)                     # in place
    assert np.array_equal(p.numpy(), a)
    ring.close()


@pytest.mark.parametrize("logn2,L,B", [(12, 2, 3), (13, 2, 2), (15, 3, 2)])
def test_block_order_is_a_relayout_of_the_reference_order(rh, oracle, logn2, L, B):
    # tuning ntt3n_block_order: the device NTT domain in the sub-transforms' (block, slot) layout, no permutation pass.
    # (1) reorder(NTT_block(x)) == NTT_reference(x) == oracle; (2) INTT_block inverts NTT_block, in and out of place;
    # (3) reorder is an involution pair; (4) the per-limb host interface keeps the reference order; (5) pointwise product chain
    #     NTT -> MulCoeffsMontgomery -> INTT gives the same coefficient-domain bits in either layout
    N = 6 << logn2
    mods = []
    q = find_prime_3n(N, 60)
    while len(mods) < L:
        if oracle.lib().orc_is_prime(q):
            mods.append(q)
        q += 3 * N
    om = [omega_for(m, N) for m in mods]
    ring = rh.Ring(N, mods, kind=rh.Matrix3N, omega3n=om)
    rng = np.random.default_rng(logn2 * 5 + L)
    mk = lambda: np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(m) for m in mods]) for _ in range(B)])
    a, b = mk(), mk()
    pa, pb = rh.DevicePoly.from_numpy(ring, a), rh.DevicePoly.from_numpy(ring, b)
    ref, blk, conv, back = (ring.NewPoly(B) for _ in range(4))
    ring.NTT(pa, ref)
    refn = ref.numpy()
    assert np.array_equal(refn[B - 1, L - 1], oracle.ntt3n_forward(a[B - 1, L - 1], mods[L - 1], om[L - 1]))
    prod_ref = ring.NewPoly(B)
    nb_ref = ring.NewPoly(B)
    ring.NTT(pb, nb_ref); ring.MulCoeffsMontgomery(ref, nb_ref, prod_ref); ring.INTT(prod_ref, prod_ref)
    ring.set_tuning("ntt3n_block_order", 1)
    ring.NTT(pa, blk)
    assert not np.array_equal(blk.numpy(), refn)
    ring.NTT3NReorder(blk, conv, to_reference=True)
    assert np.array_equal(conv.numpy(), refn)
    ring.NTT3NReorder(conv, back, to_reference=False)
    assert np.array_equal(back.numpy(), blk.numpy())
    ring.INTT(blk, back)
    assert np.array_equal(back.numpy(), a)
    ring.NTT(pa, pa); ring.INTT(pa, pa)                                  # in place, both directions
    assert np.array_equal(pa.numpy(), a)
    assert np.array_equal(ring.SubRings[0].NTT(a[0, 0]), refn[0, 0])     # host-limb interface: reference order regardless
    nb_blk, prod_blk = ring.NewPoly(B), ring.NewPoly(B)
    ring.NTT(pb, nb_blk); ring.MulCoeffsMontgomery(blk, nb_blk, prod_blk); ring.INTT(prod_blk, prod_blk)
    assert np.array_equal(prod_blk.numpy(), prod_ref.numpy())
    ring.set_tuning("ntt3n_block_order", 0)
    ring.NTT(pa, conv)
    assert np.array_equal(conv.numpy(), refn)
    ring.close()
    small = rh.Ring(96, [find_prime_3n(96, 60)], kind=rh.Matrix3N)
    small.set_tuning("ntt3n_block_order", 1)
    with pytest.raises(rh.RingHipError):                                 # block order needs N = 3 * 2^k >= 24576
        small.NTT(small.NewPoly(1), small.NewPoly(1))
    small.close()


from test_oracle_ntt3n import LARGE, large_input, check_against_large_fixture   # noqa: E402


@pytest.mark.parametrize("vec", LARGE, ids=lambda v: "N=%d" % v["N"])
def test_reference_python_vectors_at_config_sizes(rh, vec):
    # the 3N kernels the configs really run (hand-scheduled layer + tile bodies, N = 3*2^13 .. 3*2^16; two radix-3 layers at 9*2^10) against
    # outputs of the reference's own Python notes at those sizes (18-22-bit primes, omega handed over), through every route: the per-limb seam,
    # the batched reference-order transform, and block order (tagged; numpy() converts at the host boundary)
    N, p, w = vec["N"], vec["p"], vec["w"]
    x = large_input(N, p)
    ring = rh.Ring(N, [p], kind=rh.Matrix3N, omega3n=[w])
    y = ring.SubRings[0].NTT(x)
    check_against_large_fixture(vec, y)
    assert np.array_equal(ring.SubRings[0].INTT(y), x)
    B = 3
    blk = rh.DevicePoly.from_numpy(ring, np.stack([x[None]] * B))
    out = ring.NewPoly(B)
    ring.NTT(blk, out)
    got = out.numpy()
    for k in range(B):
        assert np.array_equal(got[k, 0], y)
    if rh.lib().rh_ring_ntt3n_block_order_supported(ring._h):
        ring.ntt3n_layout = "block"
        ring.NTT(blk, out)
        assert out.layout == "block"
        assert np.array_equal(out.numpy()[B - 1, 0], y)
        ring.INTT(out, out)
        assert np.array_equal(out.numpy()[1, 0], x)
    ring.close()
