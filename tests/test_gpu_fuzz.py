"""GPU: seeded random differential test of the ring entry points against the oracle.

The hand-picked cases of the other files cover the shapes the reference's tests and the BASELINE configs use; this file draws
the shapes the callers of ring.Ring produce in between -- any degree 2^3 .. 2^17, any limb count, batches on either side of the
launch-shape thresholds of the engine (one launch pair / pipelined spans of ~2048 rows), AtLevel views over polys that carry more
limbs than the view (ring/ring.go:192-213: different row strides on the two sides), in place and out of place, every ring type
-- and checks drawn (poly, limb) rows bit for bit, and that rows above the view's level are not touched.  Every case is derived
from the case number, so a failure names a reproducible shape."""
import os

import numpy as np
import pytest

from conftest import QI60, PI60, uniform_mod
from test_oracle_ntt3n import find_prime_3n, omega_for

pytestmark = pytest.mark.gpu

SENTINEL = np.uint64(0xDEADBEEFCAFEF00D)
SCALE = int(os.environ.get("RH_FUZZ_SCALE", "1"))              # RH_FUZZ_SCALE=10: ten times the cases (a longer hunt on the GPU box)


def _draw_shape(rng, max_rows):
    """(logN, L, level, B): rows = B * (level + 1) <= max_rows words-rows of N"""
    logN = int(rng.choice([3, 4, 5, 7, 10, 11, 12, 13, 14, 15, 16, 17], p=[.04, .04, .04, .06, .08, .06, .1, .12, .12, .12, .16, .06]))
    L = int(rng.choice([1, 2, 3, 5, 8, 16, 17]))
    level = int(rng.integers(0, L))
    cap = max(1, min(max_rows // (level + 1), (1 << 27) // ((level + 1) << logN)))   # <= 1 GiB per block
    B = int(rng.choice([1, 2, 3, 7, 33, 130, 300]))
    return logN, L, level, max(1, min(B, cap))


def _block(rng, mods, B, limbs, N, level):
    """B polys of `limbs` limbs: limbs 0..level uniform residues of their modulus, the others a sentinel that must survive"""
    a = np.full((B, limbs, N), SENTINEL, dtype=np.uint64)
    for i in range(level + 1):
        a[:, i] = uniform_mod(rng, mods[i], (B, N))
    return a


def _spots(rng, B, level, n=4):
    s = {(0, 0), (B - 1, level)}
    for _ in range(n):
        s.add((int(rng.integers(0, B)), int(rng.integers(0, level + 1))))
    return sorted(s)


@pytest.mark.parametrize("case", range(96 * SCALE))
def test_fuzz_standard_ntt_family(rh, oracle, case):
    """Ring.NTT / NTTLazy / INTT / INTTLazy (ring/ntt.go:127-152) through AtLevel views"""
    rng = np.random.default_rng(7000 + case)
    logN, L, level, B = _draw_shape(rng, 600)
    if case % 8 in (5, 6):                                     # just past the pipelined-span threshold of the engine (~2048 rows)
        logN = int(rng.choice([12, 13, 14, 15, 16]))
        B = 2048 // (level + 1) + int(rng.choice([1, 5, 40]))
    N = 1 << logN
    mods = (QI60 + PI60)[:L]
    which = ["NTT", "INTT", "NTTLazy", "INTTLazy"][(case + case // 8) % 4]
    inplace = bool(rng.integers(0, 2))
    limbs_in = level + 1 + int(rng.choice([0, 0, 1, 3]))
    limbs_out = limbs_in if inplace else level + 1 + int(rng.choice([0, 0, 2]))
    ring = rh.Ring(N, mods)
    view = ring.AtLevel(level)
    a = _block(rng, mods, B, limbs_in, N, level)
    pin = rh.DevicePoly.from_numpy(ring, a)
    if inplace:
        pout = pin
    else:
        pout = rh.DevicePoly.from_numpy(ring, np.full((B, limbs_out, N), SENTINEL, dtype=np.uint64))
    getattr(view, which)(pin, pout)
    got = pout.numpy()
    ctx = "case %d: %s N=2^%d L=%d level=%d B=%d limbs %d->%d inplace=%s" % (case, which, logN, L, level, B, limbs_in, limbs_out, inplace)
    # every AtLevel shape is ONE batched transform (strides inside the kernels, or one strided copy in / out): never a loop over the polys
    strided = limbs_in != level + 1 or limbs_out != level + 1
    assert ring.stats("rows_direct") + ring.stats("rows_compacted") == (1 if strided else 0) and ring.stats("rows_poly_by_poly") == 0, ctx
    for (k, i) in _spots(rng, B, level):
        sr = oracle.SubRingConsts(N, mods[i])
        f = oracle.ntt if which.startswith("NTT") else oracle.intt
        exp = f(a[k, i], sr, lazy=which.endswith("Lazy"))
        assert np.array_equal(got[k, i], exp), ctx + " row (%d, %d)" % (k, i)
    if limbs_out > level + 1:
        assert (got[:, level + 1:] == SENTINEL).all(), ctx + ": rows above the level were written"
    if not inplace:
        assert np.array_equal(pin.numpy(), a), ctx + ": input modified"
    ring.close()


@pytest.mark.parametrize("case", range(30 * SCALE))
def test_fuzz_vec_ops_at_level(rh, oracle, case):
    """one drawn element-wise kernel of ring/vec_ops.go through an AtLevel view with per-operand row strides (ring/operations.go)"""
    rng = np.random.default_rng(8000 + case)
    logN, L, level, B = _draw_shape(rng, 400)
    logN = max(logN, 4)
    N = 1 << logN
    mods = (QI60 + PI60)[:L]
    names = sorted(k for k in rh.OPS if k not in ("COUNT", "MASK", "ZERO"))
    name = names[int(rng.integers(0, len(names)))]
    code = rh.OPS[name]
    ring = rh.Ring(N, mods)
    view = ring.AtLevel(level)
    lx, ly, lz = (level + 1 + int(rng.choice([0, 1, 2])) for _ in range(3))
    lazy_in = name in ("ADD_LAZY", "SUB_LAZY", "MUL_LAZY", "MUL_LAZY_THEN_ADD_LAZY", "REDUCE", "REDUCE_LAZY", "MUL_BARRETT", "MUL_BARRETT_LAZY",
                       "MUL_MONT_LAZY", "MUL_MONT_LAZY_THEN_ADD_LAZY", "MFORM_LAZY", "MUL_MONT_LAZY_THEN_NEG")
    x, y, z = _block(rng, mods, B, lx, N, level), _block(rng, mods, B, ly, N, level), _block(rng, mods, B, lz, N, level)
    if lazy_in and case % 2:                                   # the lazy forms take any 64-bit operand
        x[:, :level + 1] = rng.integers(0, 1 << 64, size=(B, level + 1, N), dtype=np.uint64)
    s0 = np.array([int(rng.integers(1, int(q))) for q in mods[:level + 1]], dtype=np.uint64)
    s1 = np.array([int(rng.integers(1, int(q))) for q in mods[:level + 1]], dtype=np.uint64)
    px, py, pz = (rh.DevicePoly.from_numpy(ring, t) for t in (x, y, z))
    view.vec_op(name, px, py, pz, s0=s0, s1=s1)
    got = pz.numpy()
    ctx = "case %d: %s N=2^%d level=%d B=%d limbs (%d, %d, %d)" % (case, name, logN, level, B, lx, ly, lz)
    for (k, i) in _spots(rng, B, level):
        exp = oracle.vec_op(code, x[k, i], y[k, i], z[k, i], s0[i], s1[i], mods[i])
        assert np.array_equal(got[k, i], exp), ctx + " row (%d, %d)" % (k, i)
    if lz > level + 1:
        assert (got[:, level + 1:] == SENTINEL).all(), ctx + ": rows above the level were written"
    assert np.array_equal(px.numpy(), x) and np.array_equal(py.numpy(), y), ctx + ": operand modified"
    ring.close()


@pytest.mark.parametrize("case", range(24 * SCALE))
def test_fuzz_conjugate_invariant_and_3n(rh, oracle, case):
    """the other two ring types through the same batched entry points (ring/ntt.go:80-124, ring/ntt_3n.go:82-156)"""
    rng = np.random.default_rng(9000 + case)
    L = int(rng.choice([1, 2, 3]))
    level = int(rng.integers(0, L))
    B = int(rng.choice([1, 2, 5, 19]))
    inplace = bool(rng.integers(0, 2))
    if case % 2 == 0:
        logN = int(rng.choice([4, 8, 11, 12, 13, 14, 15, 16]))
        N = 1 << logN
        mods = [q for q in QI60 + PI60 if (q - 1) % (4 * N) == 0][:L]
        ring = rh.Ring(N, mods, kind=rh.ConjugateInvariant)
        srs = [oracle.SubRingConsts(N, q, nthroot=4 * N) for q in mods]
        fwd = lambda v, i: oracle.ntt_ci(v, srs[i])
        bwd = lambda v, i: oracle.intt_ci(v, srs[i])
        name = "CI N=2^%d" % logN
    else:
        N = int(rng.choice([6, 18, 48, 96, 3 * 256, 3 * 1024, 3 * 2048, 3 * 8192, 9 * 1024, 3 << 14]))
        q = find_prime_3n(N, 60)
        mods = [q]
        while len(mods) < L:
            q += 3 * N
            while not oracle.lib().orc_is_prime(q):
                q += 3 * N
            mods.append(q)
        oms = [omega_for(m, N) for m in mods]
        ring = rh.Ring(N, mods, kind=rh.Matrix3N)
        fwd = lambda v, i: oracle.ntt3n_forward(v, mods[i], oms[i])
        bwd = lambda v, i: oracle.ntt3n_backward(v, mods[i], oms[i])
        name = "3N N=%d" % N
    view = ring.AtLevel(level)
    # AtLevel views over polys with more limbs than the view (different row strides on the two sides), batched since round 3
    limbs_in = level + 1 + int(rng.choice([0, 1, 2]))
    limbs_out = limbs_in if inplace else level + 1 + int(rng.choice([0, 0, 3]))
    a = _block(rng, mods, B, limbs_in, N, level)
    pin = rh.DevicePoly.from_numpy(ring, a)
    pout = pin if inplace else rh.DevicePoly.from_numpy(ring, np.full((B, limbs_out, N), SENTINEL, dtype=np.uint64))
    inverse = bool((case // 2) % 2)
    (view.INTT if inverse else view.NTT)(pin, pout)
    got = pout.numpy()
    ctx = "case %d: %s %s L=%d level=%d B=%d limbs %d->%d inplace=%s" % (case, name, "INTT" if inverse else "NTT", L, level, B, limbs_in, limbs_out, inplace)
    for (k, i) in _spots(rng, B, level, n=2):
        assert np.array_equal(got[k, i], (bwd if inverse else fwd)(a[k, i], i)), ctx + " row (%d, %d)" % (k, i)
    if limbs_out > level + 1:
        assert (got[:, level + 1:] == SENTINEL).all(), ctx + ": rows above the level were written"
    if not inplace:
        assert np.array_equal(pin.numpy(), a), ctx + ": input modified"
    strided = limbs_in != level + 1 or limbs_out != level + 1
    assert ring.stats("rows_direct") + ring.stats("rows_compacted") == (1 if strided else 0) and ring.stats("rows_poly_by_poly") == 0, ctx
    ring.close()


@pytest.mark.parametrize("case", range(10 * SCALE))
def test_fuzz_basis_extension_levels(rh, oracle, case):
    """ModUpQtoP / ModUpPtoQ / ModDownQPtoQ at drawn (levelQ, levelP) of a larger extender (ring/basis_extension.go:188-234): the
    constants of EVERY source level are the extender's own tables; unreduced outputs compared word for word"""
    rng = np.random.default_rng(9500 + case)
    logN = int(rng.choice([4, 9, 12, 13, 14]))
    N = 1 << logN
    nq, npm = int(rng.integers(2, 9)), int(rng.integers(1, 5))
    Q, P = QI60[:nq], PI60[:npm]
    levelQ, levelP = int(rng.integers(0, nq)), int(rng.integers(0, npm))
    B = int(rng.choice([1, 2, 6]))
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    be = rh.BasisExtender(rq, rp)
    vq, vp = rq.AtLevel(levelQ), rp.AtLevel(levelP)
    xq = _block(rng, Q, B, levelQ + 1, N, levelQ)
    xp = _block(rng, P, B, levelP + 1, N, levelP)
    pq, pp = rh.DevicePoly.from_numpy(rq, xq), rh.DevicePoly.from_numpy(rp, xp)
    ctx = "case %d: N=2^%d Q=%d P=%d levelQ=%d levelP=%d B=%d" % (case, logN, nq, npm, levelQ, levelP, B)
    kind = case % 3
    if kind == 0:
        out = vp.NewPoly(B)
        be.ModUpQtoP(levelQ, levelP, pq, out)
        got = out.numpy()
        for k in {0, B - 1}:
            exp = oracle.modup_centered(xq[k], Q[:levelQ + 1], P[:levelP + 1])
            assert np.array_equal(got[k], exp), ctx + " ModUpQtoP poly %d" % k
    elif kind == 1:
        out = vq.NewPoly(B)
        be.ModUpPtoQ(levelP, levelQ, pp, out)
        got = out.numpy()
        for k in {0, B - 1}:
            exp = oracle.modup_centered(xp[k], P[:levelP + 1], Q[:levelQ + 1])
            assert np.array_equal(got[k], exp), ctx + " ModUpPtoQ poly %d" % k
    else:
        out = vq.NewPoly(B)
        be.ModDownQPtoQ(levelQ, levelP, pq, pp, out)
        got = out.numpy()
        for k in {0, B - 1}:
            exp = oracle.moddown_qp_to_q(xq[k], xp[k], Q[:levelQ + 1], P[:levelP + 1])
            assert np.array_equal(got[k], exp), ctx + " ModDownQPtoQ poly %d" % k
    be.close(); rq.close(); rp.close()


@pytest.mark.parametrize("case", range(16 * SCALE))
def test_fuzz_gadget_product_shapes(rh, oracle, case):
    """rlwe.Evaluator.GadgetProduct, hybrid branch (core/rlwe/evaluator_gadget_product.go:16-188): drawn chain lengths, levels (incl.
    last digits of a single prime and levelP below the extender's), degrees either side of the kernel-shape thresholds, small batches"""
    from oracle import compose
    rng = np.random.default_rng(9700 + case)
    logN = int(rng.choice([6, 11, 12, 13, 14, 15]))
    N = 1 << logN
    nq, npm = int(rng.integers(2, 11)), int(rng.integers(2, 5))
    levelQ, levelP = int(rng.integers(0, nq)), int(rng.integers(1, npm))
    if case % 4 == 0:
        levelQ, levelP = nq - 1, npm - 1
    Q, P = QI60[:nq], PI60[:npm]
    beta = -(-(levelQ + 1) // (levelP + 1))                                  # ceil((levelQ + 1) / (levelP + 1)) digits in play
    beta_key = max(beta, (nq - 1 + npm) // npm)
    B = int(rng.choice([1, 2, 3]))
    LQ = levelQ + 1
    cx = np.stack([np.stack([uniform_mod(rng, q, N) for q in Q[:LQ]]) for _ in range(B)])
    key = lambda mods: np.stack([np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(2)]) for _ in range(beta_key)])
    evkQ, evkP = key(Q), key(P)
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    if case % 3 == 0:
        rq.set_tuning("ks_small_rows", 0)            # the large-batch launch sequence (digit by digit + pipelined transforms) on these small batches too
    be = rh.BasisExtender(rq, rp)
    pcx = rh.DevicePoly.from_numpy(rq.AtLevel(levelQ), cx)
    dq = rh.DevicePoly.from_numpy(rq, evkQ.reshape(beta_key * 2, nq, N))
    dp = rh.DevicePoly.from_numpy(rp, evkP.reshape(beta_key * 2, npm, N))
    ct0, ct1 = rh.DevicePoly(rq, B, LQ), rh.DevicePoly(rq, B, LQ)
    be.GadgetProduct(levelQ, levelP, pcx, dq.ptr, dp.ptr, beta_key, ct0, ct1)
    g0, g1 = ct0.numpy(), ct1.numpy()
    ctx = "case %d: N=2^%d Q=%d P=%d levelQ=%d levelP=%d beta_key=%d B=%d" % (case, logN, nq, npm, levelQ, levelP, beta_key, B)
    for k in {0, B - 1}:
        e0, e1 = compose.gadget_product(N, Q, P, levelQ, levelP, cx[k], evkQ, evkP)
        assert np.array_equal(g0[k], e0) and np.array_equal(g1[k], e1), ctx + " poly %d" % k
    assert np.array_equal(pcx.numpy(), cx), ctx + ": input modified"
    be.close(); rq.close(); rp.close()


@pytest.mark.parametrize("case", range(16 * SCALE))
def test_fuzz_rescale_and_automorphism(rh, oracle, case):
    """DivFloor / DivRoundByLastModulusMany(NTT) (ring/scaling.go) with a drawn number of rescales, and AutomorphismNTT / Automorphism
    (ring/automorphism.go) with a drawn Galois element, on drawn degrees / limb counts / batches"""
    rng = np.random.default_rng(9800 + case)
    logN = int(rng.choice([4, 8, 12, 13, 14, 15, 16]))
    N = 1 << logN
    L = int(rng.integers(2, 9))
    B = int(rng.choice([1, 2, 5, 40])) if logN <= 14 else int(rng.choice([1, 3]))
    Q = QI60[:L]
    ring = rh.Ring(N, Q)
    a = _block(rng, Q, B, L, N, L - 1)
    ctx = "case %d: N=2^%d L=%d B=%d" % (case, logN, L, B)
    if case % 2 == 0:
        nb, round_ = int(rng.integers(1, min(L, 4))), int(rng.integers(0, 2))
        p0, p1 = rh.DevicePoly.from_numpy(ring, a), rh.DevicePoly(ring, B, L - nb)
        (ring.DivRoundByLastModulusMany if round_ else ring.DivFloorByLastModulusMany)(nb, p0, p1)
        got = p1.numpy()
        for k in {0, B - 1}:
            assert np.array_equal(got[k], oracle.div_by_last_modulus_many(a[k], Q, nb, round_)), ctx + " nb=%d round=%d poly %d" % (nb, round_, k)
        # NTT domain: the same division seen through the transform
        pn = rh.DevicePoly.from_numpy(ring, a)
        ring.NTT(pn, pn)
        po = rh.DevicePoly.from_numpy(ring, np.full((B, L, N), SENTINEL, dtype=np.uint64))
        (ring.DivRoundByLastModulusManyNTT if round_ else ring.DivFloorByLastModulusManyNTT)(nb, pn, po)
        out = po.numpy()
        assert (out[:, L - nb:] == SENTINEL).all(), ctx + ": limbs above the new level were written"
        sub = ring.AtLevel(L - nb - 1)
        chk = rh.DevicePoly.from_numpy(sub, out[:, :L - nb].copy())
        sub.INTT(chk, chk)
        assert np.array_equal(chk.numpy(), got), ctx + " nb=%d round=%d NTT domain" % (nb, round_)
    else:
        gen = int(rng.integers(0, N)) * 2 + 1
        pin, pout = rh.DevicePoly.from_numpy(ring, a), ring.NewPoly(B)
        ring.AutomorphismNTT(pin, gen, pout)
        got = pout.numpy()
        for (k, i) in _spots(rng, B, L - 1, n=2):
            assert np.array_equal(got[k, i], oracle.automorphism_ntt(a[k, i], gen)), ctx + " AutomorphismNTT gen=%d row (%d, %d)" % (gen, k, i)
        ring.Automorphism(pin, gen, pout)
        got = pout.numpy()
        for (k, i) in _spots(rng, B, L - 1, n=2):
            assert np.array_equal(got[k, i], oracle.automorphism(a[k, i], gen, Q[i])), ctx + " Automorphism gen=%d row (%d, %d)" % (gen, k, i)
    ring.close()


@pytest.mark.parametrize("case", range(10 * SCALE))
def test_fuzz_ckks_mul_relin_rescale_at_levels(rh, oracle, case):
    """ckks.Evaluator.MulRelin + Rescale (schemes/ckks/evaluator.go:786-881, 500-535) on ciphertexts BELOW the key's level: the tensoring, the
    relinearisation key switch at the ciphertext's level with a key of the full chain, the two Adds, the rescale -- against the oracle
    composition of tests/test_gpu_ckks.py, for drawn degrees, chain lengths, levels and batches"""
    from oracle import compose
    from test_gpu_ckks import oracle_tensor, vop
    rng = np.random.default_rng(9900 + case)
    logN = int(rng.choice([6, 12, 13, 14, 15]))
    N = 1 << logN
    nq, npm = int(rng.integers(3, 9)), int(rng.integers(2, 4))
    level = int(rng.integers(1, nq)) if case % 3 else nq - 1
    B = int(rng.choice([1, 2, 3]))
    Q, P = QI60[:nq], PI60[:npm]
    LQ = level + 1
    beta = (nq - 1 + npm) // npm
    key = lambda mods: np.stack([np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(2)]) for _ in range(beta)])
    evkQ, evkP = key(Q), key(P)
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    ev = rh.ckks.Evaluator(rq, rp, rlk=rh.rlwe.GadgetCiphertext(rq, rp, evkQ, evkP))
    rl = rq.AtLevel(level)
    mk = lambda: np.stack([np.stack([np.stack([uniform_mod(rng, q, N) for q in Q[:LQ]]) for _ in range(B)]) for _ in range(2)])
    a, b = mk(), mk()
    ct0 = rh.Ciphertext([rh.DevicePoly.from_numpy(rl, a[0]), rh.DevicePoly.from_numpy(rl, a[1])], is_ntt=True)
    ct1 = rh.Ciphertext([rh.DevicePoly.from_numpy(rl, b[0]), rh.DevicePoly.from_numpy(rl, b[1])], is_ntt=True)
    out = rh.Ciphertext([rl.NewPoly(B), rl.NewPoly(B)], is_ntt=True)
    ev.MulRelin(ct0, ct1, out, relin=True)
    got = [v.numpy() for v in out.Value]
    res = rh.Ciphertext([rl.NewPoly(B), rl.NewPoly(B)], is_ntt=True)
    ev.Rescale(out, res)
    gres = [v.numpy() for v in res.Value]
    srQ = [oracle.SubRingConsts(N, q) for q in Q[:LQ]]
    ctx = "case %d: N=2^%d Q=%d P=%d level=%d B=%d" % (case, logN, nq, npm, level, B)
    for k in {0, B - 1}:
        t = oracle_tensor(oracle, rh, a[:, k], b[:, k], Q[:LQ])
        g0, g1 = compose.gadget_product(N, Q, P, level, npm - 1, t[2], evkQ, evkP)
        for c, g in ((0, g0), (1, g1)):
            e = np.stack([vop(oracle, rh, "ADD", t[c][i], g[i], g[i], Q[i]) for i in range(LQ)])
            assert np.array_equal(got[c][k], e), ctx + " MulRelin component %d poly %d" % (c, k)
            coeff = np.stack([oracle.intt(e[i], srQ[i]) for i in range(LQ)])
            down = oracle.div_by_last_modulus_many(coeff, Q[:LQ], 1, True)
            want = np.stack([oracle.ntt(down[i], srQ[i]) for i in range(LQ - 1)])
            assert np.array_equal(gres[c][k, :LQ - 1], want), ctx + " Rescale component %d poly %d" % (c, k)
    ev.close(); rq.close(); rp.close()


@pytest.mark.parametrize("case", range(10 * SCALE))
def test_fuzz_rotation_key_switch(rh, oracle, case):
    """rlwe.Evaluator.Automorphism / AutomorphismHoisted / AutomorphismHoistedLazy + ModDown (core/rlwe/evaluator_automorphism.go:14-160) at drawn
    degrees, chains, batches and Galois elements: the direct form against the oracle composition, the hoisted and the lazy-hoisted forms
    against the direct one, bit for bit"""
    from oracle import compose
    rng = np.random.default_rng(9950 + case)
    logN = int(rng.choice([6, 11, 12, 13, 14, 15]))
    N = 1 << logN
    nq, npm = int(rng.integers(2, 9)), int(rng.integers(2, 4))
    B = int(rng.choice([1, 2, 3]))
    gal = int(rng.integers(1, N)) * 2 + 1
    Q, P = QI60[:nq], PI60[:npm]
    levelQ, levelP = nq - 1, npm - 1
    beta = (nq - 1 + npm) // npm
    key = lambda mods: np.stack([np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(2)]) for _ in range(beta)])
    evkQ, evkP = key(Q), key(P)
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    ev = rh.rlwe.Evaluator(rq, rp, galois_keys={gal: rh.rlwe.GadgetCiphertext(rq, rp, evkQ, evkP)})
    c0, c1 = _block(rng, Q, B, nq, N, levelQ), _block(rng, Q, B, nq, N, levelQ)
    ct = rh.Ciphertext([rh.DevicePoly.from_numpy(rq, c0), rh.DevicePoly.from_numpy(rq, c1)], is_ntt=True)
    out = rh.Ciphertext([rq.NewPoly(B), rq.NewPoly(B)], is_ntt=True)
    ev.Automorphism(ct, gal, out)
    g0, g1 = out.Value[0].numpy(), out.Value[1].numpy()
    ctx = "case %d: N=2^%d Q=%d P=%d B=%d gal=%d" % (case, logN, nq, npm, B, gal)
    for k in {0, B - 1}:
        e0, e1 = compose.gadget_product(N, Q, P, levelQ, levelP, c1[k], evkQ, evkP)
        for i, q in enumerate(Q):
            s = oracle.vec_op(rh.OPS["ADD"], e0[i], c0[k, i], e0[i], 0, 0, q)
            assert np.array_equal(g0[k, i], oracle.automorphism_ntt(s, gal)), ctx + " component 0 (%d, %d)" % (k, i)
            assert np.array_equal(g1[k, i], oracle.automorphism_ntt(e1[i], gal)), ctx + " component 1 (%d, %d)" % (k, i)
    dec = ev.DecomposeNTT(levelQ, levelP, ct.Value[1], True)
    out2 = rh.Ciphertext([rq.NewPoly(B), rq.NewPoly(B)], is_ntt=True)
    ev.AutomorphismHoisted(levelQ, ct, dec, gal, out2)
    assert np.array_equal(out2.Value[0].numpy(), g0) and np.array_equal(out2.Value[1].numpy(), g1), ctx + ": hoisted != direct"
    qp = rh.rlwe.ElementQP.alloc(rq, rp, B, levelQ, levelP)
    ev.AutomorphismHoistedLazy(levelQ, ct, dec, gal, qp)
    out3 = rh.Ciphertext([rq.NewPoly(B), rq.NewPoly(B)], is_ntt=True)
    ev.ModDown(levelQ, levelP, qp, out3)
    assert np.array_equal(out3.Value[0].numpy(), g0) and np.array_equal(out3.Value[1].numpy(), g1), ctx + ": lazy hoisted + ModDown != direct"
    assert np.array_equal(ct.Value[0].numpy(), c0) and np.array_equal(ct.Value[1].numpy(), c1), ctx + ": input modified"
    ev.close(); rq.close(); rp.close()
