"""Scheme-level callers (BASELINE configs 3 and 4) through the C ABI: the evaluators' call sequences on device batches.
matrix_ckks.Evaluator.Mul has no reference test: end-to-end parity is UNPINNED by the reference; it is
checked here against (a) the same sequence on the CPU oracle, whose pieces are pinned, and (b) the ring's product rule
X^N = X^(N/2) - 1 with the 2^-64 factor the reference's missing MForm leaves in (ring/ntt_3n_test.go:312-364)."""
import numpy as np
import pytest

from conftest import QI60, uniform_mod
from test_oracle_ntt3n import find_prime_3n, omega_for, naive_mul_3n

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rh():
    import matrix_fhe_lattigo_amd as m
    return m


def primes_3n(oracle, N, n, bits=60):
    q = find_prime_3n(N, bits)
    out = []
    while len(out) < n:
        out.append(q)
        q += 3 * N
        while not oracle.lib().orc_is_prime(q):
            q += 3 * N
    return out


def oracle_mul_3n(oracle, a0, a1, b0, b1, mods, N):
    """evaluator.go:114-192 on the oracle: per limb 3N NTT, MulCoeffsMontgomery x3 + ThenAdd, inverse 3N NTT"""
    OPS = __import__("matrix_fhe_lattigo_amd").OPS
    out = [np.zeros_like(a0) for _ in range(3)]
    for i, q in enumerate(mods):
        om = omega_for(q, N)
        f = lambda x: oracle.ntt3n_forward(x, q, om)
        A0, A1, B0, B1 = f(a0[i]), f(a1[i]), f(b0[i]), f(b1[i])
        z = np.zeros(N, dtype=np.uint64)
        c0 = oracle.vec_op(OPS["MUL_MONT"], A0, B0, z, 0, 0, q)
        c1 = oracle.vec_op(OPS["MUL_MONT"], A0, B1, z, 0, 0, q)
        c1 = oracle.vec_op(OPS["MUL_MONT_THEN_ADD"], A1, B0, c1, 0, 0, q)
        c2 = oracle.vec_op(OPS["MUL_MONT"], A1, B1, z, 0, 0, q)
        for o, c in zip(out, (c0, c1, c2)):
            o[i] = oracle.ntt3n_backward(c, q, om)
    return out


@pytest.mark.parametrize("N,B", [(48, 3), (3 * 1024, 2)])
def test_matrix_ckks_mul_degree1(rh, oracle, N, B):
    mods = primes_3n(oracle, N, 2)
    ring = rh.Ring(N, mods, kind=rh.Matrix3N, omega3n=[omega_for(q, N) for q in mods])
    rng = np.random.default_rng(N + B)
    mk = lambda: np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    a0, a1, b0, b1 = mk(), mk(), mk(), mk()
    ct0 = rh.Ciphertext([rh.DevicePoly.from_numpy(ring, a0), rh.DevicePoly.from_numpy(ring, a1)])
    ct1 = rh.Ciphertext([rh.DevicePoly.from_numpy(ring, b0), rh.DevicePoly.from_numpy(ring, b1)])
    out = rh.Ciphertext([ring.NewPoly(B) for _ in range(3)])
    ev = rh.MatrixCKKSEvaluator(ring)
    ev.Mul(ct0, ct1, out)
    assert ct0.IsNTT and ct1.IsNTT and not out.IsNTT            # the reference's side effects (:136-149, :189)
    got = [v.numpy() for v in out.Value]
    for k in range(B):
        exp = oracle_mul_3n(oracle, a0[k], a1[k], b0[k], b1[k], mods, N)
        for c in range(3):
            assert np.array_equal(got[c][k], exp[c]), (k, c)
    # inputs were transformed in place
    assert np.array_equal(ct0.Value[1].numpy()[0, 1], oracle.ntt3n_forward(a1[0, 1], mods[1], omega_for(mods[1], N)))
    if N <= 96:                                                  # product rule with the 2^-64 factor
        for i, q in enumerate(mods):
            rinv = pow(1 << 64, -1, q)
            A0, A1, B0, B1 = ([int(v) for v in x[0, i]] for x in (a0, a1, b0, b1))
            e0 = naive_mul_3n(A0, B0, q, N)
            e1 = [(x + y) % q for x, y in zip(naive_mul_3n(A0, B1, q, N), naive_mul_3n(A1, B0, q, N))]
            e2 = naive_mul_3n(A1, B1, q, N)
            for c, e in enumerate((e0, e1, e2)):
                assert [int(v) for v in got[c][0, i]] == [(v * rinv) % q for v in e]
    ev.fused_tensor = False                                      # the four separate ring calls: same bits
    out1 = rh.Ciphertext([ring.NewPoly(B) for _ in range(3)])
    ev.Mul(ct0, ct1, out1)
    for c in range(3):
        assert np.array_equal(out1.Value[c].numpy(), got[c])
    ev.fused_tensor = True
    # second call with inputs already in the NTT domain: same result, inputs untouched
    keep = ct0.Value[0].numpy().copy()
    out2 = rh.Ciphertext([ring.NewPoly(B) for _ in range(3)])
    ev.Mul(ct0, ct1, out2)
    assert np.array_equal(ct0.Value[0].numpy(), keep)
    for c in range(3):
        assert np.array_equal(out2.Value[c].numpy(), got[c])
    ring.close()


def test_matrix_ckks_mul_mixed_degrees_and_errors(rh, oracle):
    N, B = 96, 2
    mods = primes_3n(oracle, N, 2)
    ring = rh.Ring(N, mods, kind=rh.Matrix3N, omega3n=[omega_for(q, N) for q in mods])
    rng = np.random.default_rng(9)
    mk = lambda: np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    a0, b0, b1 = mk(), mk(), mk()
    ev = rh.MatrixCKKSEvaluator(ring)
    ct0 = rh.Ciphertext([rh.DevicePoly.from_numpy(ring, a0)])
    ct1 = rh.Ciphertext([rh.DevicePoly.from_numpy(ring, b0), rh.DevicePoly.from_numpy(ring, b1)])
    out = rh.Ciphertext([ring.NewPoly(B), ring.NewPoly(B)])
    ev.Mul(ct0, ct1, out)                                       # degree 0 x degree 1 (:155-158)
    for i, q in enumerate(mods):
        rinv = pow(1 << 64, -1, q)
        A0, B0, B1 = ([int(v) for v in x[1, i]] for x in (a0, b0, b1))
        assert [int(v) for v in out.Value[0].numpy()[1, i]] == [(v * rinv) % q for v in naive_mul_3n(A0, B0, q, N)]
        assert [int(v) for v in out.Value[1].numpy()[1, i]] == [(v * rinv) % q for v in naive_mul_3n(A0, B1, q, N)]
    with pytest.raises(rh.RingHipError):                        # level mismatch (:116-118)
        ev.Mul(ct0, rh.Ciphertext([ring.AtLevel(0).NewPoly(B)]), out)
    with pytest.raises(rh.RingHipError):                        # degree 2 input (:174-176)
        ev.Mul(rh.Ciphertext([ring.NewPoly(B) for _ in range(3)]), ct1, rh.Ciphertext([ring.NewPoly(B) for _ in range(4)]))
    ring.close()


def test_config4_size_mul_bilinear(rh, oracle):
    # config 4 ring N = 3*2^16 (2 of the 24 limbs, batch 2): Mul is bilinear -- Mul(x + x', y) = Mul(x, y) + Mul(x', y)
    N, B = 3 << 16, 2
    mods = primes_3n(oracle, N, 2)
    ring = rh.Ring(N, mods, kind=rh.Matrix3N)
    rng = np.random.default_rng(44)
    mk = lambda: np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    x0, x1, xp0, xp1, y0, y1 = mk(), mk(), mk(), mk(), mk(), mk()
    dp = lambda a: rh.DevicePoly.from_numpy(ring, a)
    ev = rh.MatrixCKKSEvaluator(ring)

    def mul(u0, u1):
        o = rh.Ciphertext([ring.NewPoly(B) for _ in range(3)])
        ev.Mul(rh.Ciphertext([dp(u0), dp(u1)]), rh.Ciphertext([dp(y0), dp(y1)]), o)
        return o
    r1, r2 = mul(x0, x1), mul(xp0, xp1)
    s0, s1 = dp(x0), dp(x1)
    ring.Add(s0, dp(xp0), s0); ring.Add(s1, dp(xp1), s1)
    rs = rh.Ciphertext([ring.NewPoly(B) for _ in range(3)])
    ev.Mul(rh.Ciphertext([s0, s1]), rh.Ciphertext([dp(y0), dp(y1)]), rs)
    for c in range(3):
        ring.Add(r1.Value[c], r2.Value[c], r1.Value[c])
        assert np.array_equal(rs.Value[c].numpy(), r1.Value[c].numpy())
    ring.close()


def test_config3_polymul(rh, oracle):
    # config 3: N = 2^15, 16 limbs, c = INTT(NTT(a) . NTT(b)) as mulRelin sequences it; X^i * X^j = -X^(i+j-N) wraps
    N, L, B = 1 << 15, 16, 3
    mods = QI60[:L]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(3)
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    b = np.zeros_like(a)
    b[:, :, 1] = 1                                               # multiply by X: negacyclic shift
    b[1, :, 1] = 0; b[1, :, N - 1] = 5                           # poly 1: multiply by 5 X^(N-1)
    pa, pb, pc, tmp = (rh.DevicePoly.from_numpy(ring, a), rh.DevicePoly.from_numpy(ring, b), ring.NewPoly(B), ring.NewPoly(B))
    rh.ckks_polymul(ring, pa, pb, pc, tmp)
    got = pc.numpy()
    for i, q in enumerate(mods):
        q = int(q)
        e0 = np.roll(a[0, i], 1); e0[0] = (q - int(e0[0])) % q
        assert np.array_equal(got[0, i], e0)
        e1 = np.roll(a[1, i], -1).astype(object) * 5 % q        # X^(N-1) * X^j = -X^(j-1) for j >= 1, X^(N-1) for j = 0
        e1 = np.array([(q - int(v)) % q for v in e1], dtype=np.uint64)
        e1[N - 1] = (5 * int(a[1, i, 0])) % q
        assert np.array_equal(got[1, i], e1)
    # one limb of the dense product against the oracle sequence
    sr = oracle.SubRingConsts(N, mods[2])
    assert np.array_equal(pa.numpy()[2, 2], oracle.ntt(a[2, 2], sr))
    ring.close()


@pytest.mark.parametrize("logN,L,B", [(6, 2, 2), (12, 3, 3), (13, 2, 3), (15, 16, 3), (16, 3, 2)])
def test_intt_mul_equals_the_three_ring_calls(rh, oracle, logN, L, B):
    # Ring.INTTMul (rh_ring_intt_mul): INTT(a . b) with the product formed on load == MForm; MulCoeffsMontgomery; INTT
    # (schemes/ckks/evaluator.go:821-834 + INTT), and one limb against the oracle sequence; lazy inputs (< 2q) accepted
    N, mods = 1 << logN, QI60[:L]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(logN * 7 + L)
    mk = lambda: np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    a, b = mk(), mk()
    a[0, :, 0] = 0; a[0, :, 1] = np.array(mods, dtype=np.uint64) - np.uint64(1); b[0, :, 1] = np.array(mods, dtype=np.uint64) - np.uint64(1)
    pa, pb = rh.DevicePoly.from_numpy(ring, a), rh.DevicePoly.from_numpy(ring, b)
    ref, tmp, got = ring.NewPoly(B), ring.NewPoly(B), ring.NewPoly(B)
    ring.MForm(pa, tmp); ring.MulCoeffsMontgomery(tmp, pb, ref); ring.INTT(ref, ref)
    ring.INTTMul(pa, pb, got)
    assert np.array_equal(got.numpy(), ref.numpy())
    assert np.array_equal(pa.numpy(), a) and np.array_equal(pb.numpy(), b)            # inputs untouched
    i = L - 1
    sr = oracle.SubRingConsts(N, mods[i])
    z = np.zeros(N, dtype=np.uint64)
    m = oracle.vec_op(rh.OPS["MFORM"], a[B - 1, i], None, z, 0, 0, mods[i])
    e = oracle.intt(oracle.vec_op(rh.OPS["MUL_MONT"], m, b[B - 1, i], z, 0, 0, mods[i]), sr)
    assert np.array_equal(got.numpy()[B - 1, i], e)
    lazy = a.copy(); lazy[:, :, ::3] += np.array(mods, dtype=np.uint64)[None, :, None]   # representatives in [q, 2q)
    ring.INTTMul(rh.DevicePoly.from_numpy(ring, lazy), pb, got)
    assert np.array_equal(got.numpy(), ref.numpy())
    if logN > 12:                                                                      # the pipelined launches (spans of 2 polys), and the C++ bodies
        for key, val in (("chunk_polys", 2), ("asm_tile", 0)):
            ring.set_tuning(key, val)
            ring.INTTMul(rh.DevicePoly.from_numpy(ring, a), pb, got)
            assert np.array_equal(got.numpy(), ref.numpy()), key
        ring.set_tuning("asm_tile", 1)
        ring.INTTMul(pa, pb, pa)                                                       # pipelined, output aliases an input
        assert np.array_equal(pa.numpy(), ref.numpy())
        ring.set_tuning("chunk_polys", -1)
        pa = rh.DevicePoly.from_numpy(ring, a)
    ring.INTTMul(pa, pb, pa)                                                           # output aliases an input
    assert np.array_equal(pa.numpy(), ref.numpy())
    ring.close()


def test_matrix_ckks_rescale_and_add(rh, oracle):
    # matrix_ckks.Evaluator.Rescale (evaluator.go:208-243: DivRoundByLastModulusManyNTT on the 3N ring, one level) and Add (:60-102,
    # unequal degrees: the longer ciphertext's components are copied)
    N, L, B = 3 << 13, 3, 2
    mods = primes_3n(oracle, N, L)
    ring = rh.Ring(N, mods, kind=rh.Matrix3N)
    ev = rh.MatrixCKKSEvaluator(ring)
    rng = np.random.default_rng(8)
    mk = lambda: np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    c0, c1, d0 = mk(), mk(), mk()
    ct = rh.Ciphertext([rh.DevicePoly.from_numpy(ring, c0), rh.DevicePoly.from_numpy(ring, c1)], is_ntt=True)
    for v in ct.Value:
        ring.NTT(v, v)
    out = rh.Ciphertext([rh.DevicePoly(ring, B, L - 1), rh.DevicePoly(ring, B, L - 1)])
    ev.Rescale(ct, out)
    assert out.IsNTT and out.Level() == L - 2
    sub = ring.AtLevel(L - 2)
    for v, src in zip(out.Value, (c0, c1)):
        sub.INTT(v, v)
        exp = np.stack([oracle.div_by_last_modulus_many(src[k], mods, 1, 1) for k in range(B)])
        assert np.array_equal(v.numpy(), exp)
    with pytest.raises(rh.RingHipError):                        # level too low (:217-219)
        ev.Rescale(rh.Ciphertext([ring.AtLevel(0).NewPoly(B)]), rh.Ciphertext([ring.AtLevel(0).NewPoly(B)]))
    # Add: degree 1 + degree 0
    x = rh.Ciphertext([rh.DevicePoly.from_numpy(ring, c0), rh.DevicePoly.from_numpy(ring, c1)])
    y = rh.Ciphertext([rh.DevicePoly.from_numpy(ring, d0)])
    s = rh.Ciphertext([ring.NewPoly(B), ring.NewPoly(B)])
    ev.Add(x, y, s)
    for i, q in enumerate(mods):
        assert np.array_equal(s.Value[0].numpy()[:, i], (c0[:, i] + d0[:, i]) % np.uint64(q))
    assert np.array_equal(s.Value[1].numpy(), c1)
    ring.close()


@pytest.mark.parametrize("kind", ["ci", "3n"])
def test_intt_mul_and_polymul_on_other_ring_types(rh, oracle, kind):
    # Ring.INTTMul / ckks_polymul on the conjugate-invariant and the 3N ring: the three ring calls through the ring's own transform
    B, L = 2, 2
    if kind == "ci":
        N, mods = 4096, QI60[:L]
        ring = rh.Ring(N, mods, kind=rh.ConjugateInvariant)
    else:
        N = 3 << 10
        mods = primes_3n(oracle, N, L)
        ring = rh.Ring(N, mods, kind=rh.Matrix3N)
    rng = np.random.default_rng(3)
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    b = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    pa, pb = rh.DevicePoly.from_numpy(ring, a), rh.DevicePoly.from_numpy(ring, b)
    ring.NTT(pa, pa); ring.NTT(pb, pb)
    t, ref, got = ring.NewPoly(B), ring.NewPoly(B), ring.NewPoly(B)
    ring.MForm(pa, t); ring.MulCoeffsMontgomery(t, pb, ref); ring.INTT(ref, ref)
    ring.INTTMul(pa, pb, got)
    assert np.array_equal(got.numpy(), ref.numpy())
    qa, qb, c = rh.DevicePoly.from_numpy(ring, a), rh.DevicePoly.from_numpy(ring, b), ring.NewPoly(B)
    rh.schemes.ckks_polymul(ring, qa, qb, c)
    assert np.array_equal(c.numpy(), ref.numpy())
    ring.close()


def test_matrix_ckks_new_forms_and_level_drop(rh, oracle):
    # MulNew / AddNew / RescaleNew / ModDownNew (evaluator.go:104-111, 195-200, 246-251, 259-297)
    N, L, B = 3 << 10, 3, 2
    mods = primes_3n(oracle, N, L)
    ring = rh.Ring(N, mods, kind=rh.Matrix3N)
    ev = rh.MatrixCKKSEvaluator(ring)
    rng = np.random.default_rng(9)
    mk = lambda: np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    a0, a1, b0, b1 = mk(), mk(), mk(), mk()
    dp = lambda x: rh.DevicePoly.from_numpy(ring, x)
    x, y = rh.Ciphertext([dp(a0), dp(a1)]), rh.Ciphertext([dp(b0), dp(b1)])
    s = ev.AddNew(x, y)
    for i, q in enumerate(mods):
        assert np.array_equal(s.Value[1].numpy()[:, i], (a1[:, i] + b1[:, i]) % np.uint64(q))
    prod = ev.MulNew(rh.Ciphertext([dp(a0), dp(a1)]), rh.Ciphertext([dp(b0), dp(b1)]))
    ref = rh.Ciphertext([ring.NewPoly(B) for _ in range(3)])
    ev.Mul(rh.Ciphertext([dp(a0), dp(a1)]), rh.Ciphertext([dp(b0), dp(b1)]), ref)
    assert prod.Degree() == 2 and all(np.array_equal(u.numpy(), v.numpy()) for u, v in zip(prod.Value, ref.Value))
    low = ev.ModDownNew(x, 1)
    assert low.Level() == L - 2 and np.array_equal(low.Value[0].numpy(), a0[:, :L - 1]) and np.array_equal(low.Value[1].numpy(), a1[:, :L - 1])
    xn = rh.Ciphertext([dp(a0), dp(a1)], is_ntt=True)
    for v in xn.Value:
        ring.NTT(v, v)
    r = ev.RescaleNew(xn)
    sub = ring.AtLevel(L - 2)
    sub.INTT(r.Value[0], r.Value[0])
    assert np.array_equal(r.Value[0].numpy(), np.stack([oracle.div_by_last_modulus_many(a0[k], mods, 1, 1) for k in range(B)]))
    ring.close()


def test_block_order_tags_follow_the_data(rh, oracle):
    # block order is the DEFAULT for device-resident 3N chains, carried as a tag on each device block; the host boundary
    # (numpy()) and any meeting with reference-order NTT-domain data convert lazily; key-switch callers get the reference's order
    N, L, B = 3 << 13, 3, 2
    mods = primes_3n(oracle, N, L)
    om = [omega_for(q, N) for q in mods]
    ring = rh.Ring(N, mods, kind=rh.Matrix3N, omega3n=om)
    assert ring.ntt3n_layout is None
    ev = rh.MatrixCKKSEvaluator(ring)
    assert ring.ntt3n_layout == "block" and ring.AtLevel(1).ntt3n_layout == "block"            # views share the handle's setting
    rng = np.random.default_rng(31)
    mk = lambda: np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    a, b = mk(), mk()
    fwd = lambda x: np.stack([np.stack([oracle.ntt3n_forward(x[k, i], mods[i], om[i]) for i in range(L)]) for k in range(B)])
    fa, fb = fwd(a), fwd(b)
    pa = rh.DevicePoly.from_numpy(ring, a)
    assert pa.layout is None
    ring.NTT(pa, pa)
    assert pa.layout == "block"
    assert np.array_equal(pa.numpy(), fa)                                  # the host sees the reference's order ...
    raw = np.empty((B, L, N), dtype=np.uint64)
    rh.ringhip._check(rh.lib().rh_dev_download(ring._h, raw.ctypes.data_as(rh.ringhip.U64P), pa.ptr, raw.size))
    assert not np.array_equal(raw, fa) and np.array_equal(np.sort(raw, axis=2), np.sort(fa, axis=2))   # ... of data that really lies in another order
    # a reference-order NTT-domain operand from the host meets a block-order one: converted first, result in the reference's order
    pb_ref = rh.DevicePoly.from_numpy(ring, fb)
    s = ring.NewPoly(B)
    ring.Add(pa, pb_ref, s)
    assert s.layout is None and pa.layout is None
    exp = np.stack([(fa[:, i] + fb[:, i]) % np.uint64(q) for i, q in enumerate(mods)], axis=1)
    assert np.array_equal(s.numpy(), exp)
    # two block-order operands stay in block order; z-reading opcodes count their output as an operand
    pa2, pb2 = rh.DevicePoly.from_numpy(ring, a), rh.DevicePoly.from_numpy(ring, b)
    ring.NTT(pa2, pa2); ring.NTT(pb2, pb2)
    t = ring.NewPoly(B)
    ring.MulCoeffsMontgomery(pa2, pb2, t)
    assert t.layout == "block"
    ring.MulCoeffsMontgomeryThenAdd(pa2, pb2, s)                           # s is reference-order: everything comes back to it
    assert s.layout is None and pa2.layout is None
    MUL = rh.OPS["MUL_MONT"]
    z = np.zeros(N, dtype=np.uint64)
    prod = np.stack([np.stack([oracle.vec_op(MUL, fa[k, i], fb[k, i], z, 0, 0, mods[i]) for i in range(L)]) for k in range(B)])
    assert np.array_equal(t.numpy(), prod)
    assert np.array_equal(s.numpy(), np.stack([(exp[:, i] + prod[:, i]) % np.uint64(q) for i, q in enumerate(mods)], axis=1))
    # INTT reads the operand's tag whatever the ring's setting; CopyLvl and a lower-level view carry it
    low = ring.AtLevel(1)
    c = rh.DevicePoly(low, B, 2)
    low.CopyLvl(t, c)
    assert c.layout == "block"
    low.INTT(c, c)
    assert c.layout is None
    back = np.stack([np.stack([oracle.ntt3n_backward(prod[k, i], mods[i], om[i]) for i in range(2)]) for k in range(B)])
    assert np.array_equal(c.numpy(), back)
    ring.ntt3n_layout = None
    ring.INTT(t, t)                                                        # still tagged: converted by the kernel choice, not by the setting
    assert np.array_equal(t.numpy()[:, :2], back)
    with pytest.raises(rh.RingHipError):
        rh.Ring(3 << 6, primes_3n(oracle, 3 << 6, 1), kind=rh.Matrix3N).ntt3n_layout = "block"     # too small for block order
    ring.close()


def test_matrix_ckks_mul_by_const(rh, oracle):
    # matrix_ckks.Evaluator.MulByConst (evaluator.go:322-380): integer constants as they are, float constants scaled by the level's modulus and
    # rounded half away from zero; both halves of the double RNS scalar equal the constant when it is real.  Against Python integers.
    N, L, B = 3 << 6, 3, 2
    mods = primes_3n(oracle, N, L)
    ring = rh.Ring(N, mods, kind=rh.Matrix3N)
    ev = rh.MatrixCKKSEvaluator(ring)
    rng = np.random.default_rng(77)
    mk = lambda: np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    c0, c1 = mk(), mk()
    ct = rh.Ciphertext([rh.DevicePoly.from_numpy(ring, c0), rh.DevicePoly.from_numpy(ring, c1)])
    out = rh.Ciphertext([ring.NewPoly(B), ring.NewPoly(B)])
    from fractions import Fraction
    for const in (3, -7, 0, 2.5, -0.1, 1e-3, 12345.678):
        scale = ev.MulByConst(ct, const, out)
        if float(const).is_integer():
            assert scale == 1
            k = int(const)
        else:
            assert scale == mods[L - 1]
            v = Fraction(const) * scale
            k = int(v + Fraction(1, 2)) if const > 0 else int(v - Fraction(1, 2))
        for src, got in ((c0, out.Value[0].numpy()), (c1, out.Value[1].numpy())):
            for i, q in enumerate(mods):
                want = (src[1, i].astype(object) * (k % q)) % q
                assert [int(x) for x in got[1, i][:40]] == [int(x) for x in want[:40]], (const, i)
                assert int(got[1, i][N - 1]) == int(want[N - 1])
    # a complex constant: the two halves differ through RootsForward[1]; with the root w the scalars are real +- MRed(imag, w)
    w = [int(rng.integers(1, int(q))) for q in mods]
    scale = ev.MulByConst(ct, complex(2, 3), out, roots_forward_1=w)
    assert scale == 1
    got = out.Value[0].numpy()
    for i, q in enumerate(mods):
        im = (3 * w[i] * pow(1 << 64, -1, q)) % q
        lo, hi = (2 + im) % q, (2 - im) % q
        want = np.concatenate([(c0[0, i, :N // 2].astype(object) * lo) % q, (c0[0, i, N // 2:].astype(object) * hi) % q])
        assert [int(x) for x in got[0, i]] == [int(x) for x in want]
    with pytest.raises(rh.RingHipError):
        ev.MulByConst(ct, complex(1, 1), out)                            # no RootsForward[1] handed over
    with pytest.raises(rh.RingHipError):
        ev.MulByConst(ct, 2, rh.Ciphertext([ring.AtLevel(0).NewPoly(B), ring.AtLevel(0).NewPoly(B)]))
    ring.close()


from test_oracle_ntt3n import PRODUCT, product_inputs, check_mul_against_reference_product   # noqa: E402


@pytest.mark.parametrize("vec", PRODUCT, ids=lambda v: "N=%d" % v["N"])
def test_matrix_ckks_mul_against_reference_python_ring_product(rh, vec):
    # the reference holds no vector for matrix_ckks.Evaluator.Mul.  Its Python notes do compute the ring product in
    # Z_p[X]/(X^N - X^(N/2) + 1) (references/integer.py); the Go Mul is that product times 2^-64.  Degree 1 x degree 1, default evaluator
    # (block-order device NTT domain where the ring allows it), sizes up to config 4's ring; the engine picks its OWN omega -- the product
    # does not depend on which primitive 3N-th root the transform evaluates at.
    N, p = vec["N"], vec["p"]
    ring = rh.Ring(N, [p], kind=rh.Matrix3N)
    a0, a1, b0, b1 = product_inputs(N, p)
    dp = lambda v: rh.DevicePoly.from_numpy(ring, v[None, None])
    ev = rh.MatrixCKKSEvaluator(ring)
    out = rh.Ciphertext([ring.NewPoly(1) for _ in range(3)])
    ev.Mul(rh.Ciphertext([dp(a0), dp(a1)]), rh.Ciphertext([dp(b0), dp(b1)]), out)
    check_mul_against_reference_product(vec, [v.numpy()[0, 0] for v in out.Value])
    ring.close()


@pytest.mark.parametrize("logN,L,B", [(13, 2, 3), (14, 5, 2), (15, 16, 3), (16, 3, 5), (17, 2, 1), (15, 16, 130), (14, 8, 600), (13, 2, 3000), (17, 2, 1100), (16, 16, 300)])
def test_polymul_one_tile_kernel_equals_the_five_ring_calls(rh, oracle, logN, L, B):
    # Ring.PolyMul (rh_ring_polymul): forward tile stages of both operands + product + inverse tile stages as ONE kernel per 4096-tile (config 3
    # without NTT(a), NTT(b) or the product in memory) -- the canonical values of NTT, NTT, MForm, MulCoeffsMontgomery, INTT, bit for bit.  Batches
    # of more than ~2048 rows run the software pipeline (column stages of span j + tile middle of span j-1 + inverse column stages of span j-2 in one
    # launch): two spans with a short tail, three spans, and the compiled column stages of N = 2^13 / 2^17 inside the fused launch
    N, mods = 1 << logN, QI60[:L]
    ring = rh.Ring(N, mods)
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(logN * 1000 + L * 10 + B)
    qs = torch.tensor(mods, dtype=torch.int64, device=dev).view(1, L, 1)
    ta = torch.randint(0, 1 << 62, (B, L, N), dtype=torch.int64, device=dev, generator=g) % qs
    tb = torch.randint(0, 1 << 62, (B, L, N), dtype=torch.int64, device=dev, generator=g) % qs
    ta[0, :, :3] = torch.tensor([0, 1, 2], device=dev)
    tb[B - 1, :, N - 1] = qs.view(L) - 1
    keep_a, keep_b = ta.clone(), tb.clone()
    dp = lambda t: rh.DevicePoly.from_torch(ring, t)
    ref = torch.empty_like(ta)
    ra, rb = keep_a.clone(), keep_b.clone()
    rh.ckks_polymul(ring, dp(ra), dp(rb), dp(ref), dp(torch.empty_like(ta)), fused=False)     # the five ring calls as written
    out = torch.empty_like(ta)
    rh.ckks_polymul(ring, dp(ta), dp(tb), dp(out), fused="tile")
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    # output aliasing an operand
    ta2, tb2 = keep_a.clone(), keep_b.clone()
    ring.PolyMul(dp(ta2), dp(tb2), dp(tb2))
    torch.cuda.synchronize()
    assert torch.equal(tb2, ref)
    # one row against the oracle's own sequence
    k, i = B - 1, L - 1
    sr = oracle.SubRingConsts(N, mods[i])
    h = lambda t: t[k, i].cpu().numpy().view(np.uint64)
    A, Bn = oracle.ntt(h(keep_a), sr), oracle.ntt(h(keep_b), sr)
    prod = oracle.vec_op(rh.OPS["MUL_MONT"], oracle.vec_op(rh.OPS["MFORM"], A, A, A, 0, 0, mods[i]), Bn, Bn, 0, 0, mods[i])
    assert np.array_equal(h(out), oracle.intt(prod, sr))
    with pytest.raises(rh.RingHipError):
        small = rh.Ring(4096, mods)
        small.PolyMul(small.NewPoly(1), small.NewPoly(1), small.NewPoly(1))                     # N = 4096: not covered by the fused kernel
    ring.close()
