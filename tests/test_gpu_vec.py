"""GPU: every element-wise kernel of ring/vec_ops.go, bit-exact against the oracle (formulas incl. the lazy forms)."""
import numpy as np
import pytest

from conftest import QI60, uniform_mod

pytestmark = pytest.mark.gpu


def ops_table(rh):
    return sorted((v, k) for k, v in rh.OPS.items() if k != "COUNT")


@pytest.mark.parametrize("N", [16, 4096, 1 << 15])
def test_all_vec_ops_vs_oracle(rh, oracle, N):
    mods = QI60[:3]
    ring = rh.Ring(N, mods) if N >= 16 else None
    rng = np.random.default_rng(N)
    npoly = 2
    shape = (npoly, len(mods), N)
    qs = np.array(mods, dtype=np.uint64)[None, :, None]
    x = rng.integers(0, 1 << 62, size=shape, dtype=np.uint64) % qs
    y = rng.integers(0, 1 << 62, size=shape, dtype=np.uint64) % qs
    z = rng.integers(0, 1 << 62, size=shape, dtype=np.uint64) % qs
    # edge operands (ring/ring_test.go:534-670 style): 0, 1, q-1
    x[0, :, 0] = 0; x[0, :, 1] = 1; x[0, :, 2] = qs[0, :, 0] - np.uint64(1)
    y[0, :, 0] = qs[0, :, 0] - np.uint64(1); y[0, :, 1] = qs[0, :, 0] - np.uint64(1); y[0, :, 2] = qs[0, :, 0] - np.uint64(1)
    s0 = np.array([int(rng.integers(1, q)) for q in mods], dtype=np.uint64)
    s1 = np.array([int(rng.integers(1, q)) for q in mods], dtype=np.uint64)
    px, py = rh.DevicePoly.from_numpy(ring, x), rh.DevicePoly.from_numpy(ring, y)
    for code, name in ops_table(rh):
        a0, a1 = s0, s1
        if name == "MASK":
            a0 = np.array([7, 13, 0], dtype=np.uint64); a1 = np.array([(1 << 20) - 1, 0xffff, (1 << 61) - 1], dtype=np.uint64)
        pz = rh.DevicePoly.from_numpy(ring, z)
        ring.vec_op(name, px, py, pz, s0=a0, s1=a1)
        got = pz.numpy()
        for k in range(npoly):
            for i, q in enumerate(mods):
                exp = oracle.vec_op(code, x[k, i], y[k, i], z[k, i], a0[i], a1[i], q)
                assert np.array_equal(got[k, i], exp), (name, k, i)
        pz.free()
    # in-place use (p1 is p3), as the reference's callers do
    pz = rh.DevicePoly.from_numpy(ring, x)
    ring.MulCoeffsMontgomery(pz, py, pz)
    exp = np.stack([np.stack([oracle.vec_op(rh.OPS["MUL_MONT"], x[k, i], y[k, i], x[k, i], 0, 0, mods[i]) for i in range(3)]) for k in range(npoly)])
    assert np.array_equal(pz.numpy(), exp)
    ring.close()


def test_lazy_inputs_full_range(rh, oracle):
    # lazy kernels take any 64-bit operand; check wrap-around behaviour is identical
    N, mods = 64, QI60[:2]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(5)
    x = rng.integers(0, 1 << 64, size=(1, 2, N), dtype=np.uint64)
    y = rng.integers(0, 1 << 64, size=(1, 2, N), dtype=np.uint64)
    z = rng.integers(0, 1 << 64, size=(1, 2, N), dtype=np.uint64)
    px, py = rh.DevicePoly.from_numpy(ring, x), rh.DevicePoly.from_numpy(ring, y)
    for name in ["ADD_LAZY", "SUB_LAZY", "MUL_LAZY", "MUL_LAZY_THEN_ADD_LAZY", "REDUCE", "REDUCE_LAZY", "MUL_BARRETT",
                 "MUL_BARRETT_LAZY", "MUL_MONT_LAZY", "MUL_MONT_LAZY_THEN_ADD_LAZY", "MFORM_LAZY", "MUL_MONT_LAZY_THEN_NEG"]:
        pz = rh.DevicePoly.from_numpy(ring, z)
        ring.vec_op(name, px, py, pz)
        got = pz.numpy()
        for i, q in enumerate(mods):
            assert np.array_equal(got[0, i], oracle.vec_op(rh.OPS[name], x[0, i], y[0, i], z[0, i], 0, 0, q)), name
    ring.close()


def test_scalar_forms_of_ring_operations(rh):
    # ring/operations.go:151-275: AddScalar(Bigint), SubScalar(Bigint), MulScalar(ThenAdd / ThenSub), MulScalarBigint(ThenAdd), EvalPolyScalar
    # against plain integer arithmetic
    N, mods = 256, QI60[:3]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(12)
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(2)])
    b = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(2)])
    big = (1 << 200) + 12345678901234567890
    sm = 0xFFFFFFFFFFFFFFF1                                       # a uint64 scalar above every modulus
    A = [[int(x) for x in a[1, i]] for i in range(3)]
    B = [[int(x) for x in b[1, i]] for i in range(3)]

    def run(fn, *args, acc=None):
        pa = rh.DevicePoly.from_numpy(ring, a)
        po = rh.DevicePoly.from_numpy(ring, b if acc is None else acc)
        fn(pa, *args, po)
        return [[int(x) for x in po.numpy()[1, i]] for i in range(3)]
    for i, q in enumerate(mods):
        q = int(q)
        assert run(ring.AddScalar, 77)[i] == [(x + 77) % q for x in A[i]]
        assert run(ring.SubScalar, 77)[i] == [(x - 77) % q for x in A[i]]
        assert run(ring.AddScalarBigint, big)[i] == [(x + big) % q for x in A[i]]
        assert run(ring.SubScalarBigint, big)[i] == [(x - big) % q for x in A[i]]
        assert run(ring.MulScalar, sm)[i] == [(x * sm) % q for x in A[i]]
        assert run(ring.MulScalarBigint, big)[i] == [(x * big) % q for x in A[i]]
        assert run(ring.MulScalarThenAdd, sm)[i] == [(y + x * sm) % q for x, y in zip(A[i], B[i])]
        assert run(ring.MulScalarThenSub, sm)[i] == [(y - x * sm) % q for x, y in zip(A[i], B[i])]
        assert run(ring.MulScalarBigintThenAdd, big)[i] == [(y + x * big) % q for x, y in zip(A[i], B[i])]
    pa, pb, po = rh.DevicePoly.from_numpy(ring, a), rh.DevicePoly.from_numpy(ring, b), ring.NewPoly(2)
    ring.EvalPolyScalar([pa, pb, pa], 5, po)                    # a + 5 b + 25 a
    for i, q in enumerate(mods):
        assert [int(x) for x in po.numpy()[1, i]] == [(26 * x + 5 * y) % int(q) for x, y in zip(A[i], B[i])]
    ring.close()


def test_double_rns_scalars_vector_shift_and_monomial(rh):
    # ring/operations.go:167-184, 250-266 (one RNS scalar per half of the coefficients), :366-377 (one vector for every limb), :278-282
    # (Shift, with the reference's known answer ring/ring_test.go:904-916), :306-363 (MultByMonomial)
    N, mods = 64, QI60[:2]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(21)
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(2)])
    b = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(2)])
    s0, s1 = [11, 1 << 40], [(1 << 61) - 5, 3]
    A = lambda k, i: [int(x) for x in a[k, i]]
    pa, pb, po = rh.DevicePoly.from_numpy(ring, a), rh.DevicePoly.from_numpy(ring, b), ring.NewPoly(2)
    for name, f in (("Add", lambda x, s, q: (x + s) % q), ("Sub", lambda x, s, q: (x - s) % q), ("Mul", lambda x, s, q: (x * s) % q)):
        getattr(ring, name + "DoubleRNSScalar")(pa, s0, s1, po)
        g = po.numpy()
        for i, q in enumerate(mods):
            q = int(q)
            exp = [f(x, s0[i] % q, q) for x in A(1, i)[:N // 2]] + [f(x, s1[i] % q, q) for x in A(1, i)[N // 2:]]
            assert [int(x) for x in g[1, i]] == exp, name
    acc = rh.DevicePoly.from_numpy(ring, b)
    ring.MulDoubleRNSScalarThenAdd(pa, s0, s1, acc)
    for i, q in enumerate(mods):
        q = int(q)
        exp = [(y + x * (s0[i] if j < N // 2 else s1[i])) % q for j, (x, y) in enumerate(zip(A(0, i), [int(v) for v in b[0, i]]))]
        assert [int(x) for x in acc.numpy()[0, i]] == exp
    # one vector for every limb: MulCoeffsMontgomery(p1[i], vector) = x * v * 2^-64
    vec = np.array([[rng.integers(0, 1 << 60, size=N, dtype=np.uint64)]])
    pv = rh.DevicePoly.from_numpy(ring.AtLevel(0), vec)
    ring.MulByVectorMontgomery(pa, pv, po)
    for i, q in enumerate(mods):
        q = int(q); rinv = pow(1 << 64, -1, q)
        assert [int(x) for x in po.numpy()[1, i]] == [(x * int(v) * rinv) % q for x, v in zip(A(1, i), vec[0, 0])]
    acc = rh.DevicePoly.from_numpy(ring, b)
    ring.MulByVectorMontgomeryThenAddLazy(pa, pv, acc)
    for i, q in enumerate(mods):
        q = int(q); rinv = pow(1 << 64, -1, q)
        assert [int(x) % q for x in acc.numpy()[1, i]] == [(int(y) + x * int(v) * rinv) % q for x, v, y in zip(A(1, i), vec[0, 0], b[1, i])]
    # Shift: the reference's known answer (N = 16, q = 97, k = 3)
    r16 = rh.Ring(16, [97])
    p1 = rh.DevicePoly.from_numpy(r16, np.arange(16, dtype=np.uint64).reshape(1, 1, 16)); p2 = r16.NewPoly(1)
    r16.Shift(p1, 3, p2)
    assert [int(x) for x in p2.numpy()[0, 0]] == [3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2]
    r16.Shift(p2, -3, p2)                                         # in place, back
    assert [int(x) for x in p2.numpy()[0, 0]] == list(range(16))
    r16.close()
    # MultByMonomial: the reference's loops restated, and X^1 then X^8 == X^9 (ring/ring_test.go:880-899)
    def monomial(x, k, q):
        sh = (k + 2 * N) % (2 * N)
        if sh == 0:
            return list(x)
        t = list(x) if sh < N else [q - v for v in x]
        sh %= N
        return [q - t[N - sh + j] for j in range(sh)] + [t[j - sh] for j in range(sh, N)]
    for k in (1, 9, N, N + 5, -7, 2 * N):
        ring.MultByMonomial(pa, k, po)
        for i, q in enumerate(mods):
            assert [int(x) for x in po.numpy()[1, i]] == monomial(A(1, i), k, int(q)), k
    t1 = ring.NewPoly(2)
    ring.MultByMonomial(pa, 1, t1); ring.MultByMonomial(t1, 8, t1); ring.MultByMonomial(pa, 9, po)
    assert np.array_equal(t1.numpy(), po.numpy())
    ring.close()
