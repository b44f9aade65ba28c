"""GPU: negacyclic NTT parity through the C ABI -- bit-exact against the reference's KATs and the pinned oracle."""
import json
import os

import numpy as np
import pytest

from conftest import QI60, uniform_mod

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ntt_kat.json")))["vectors"]


@pytest.mark.parametrize("vec", KAT, ids=lambda v: "N=%d" % v["N"])
def test_kat_through_transformer_interface(rh, vec):
    # TestNTT (ring/ntt_test.go:91-121) through the per-limb NumberTheoreticTransformer seam, engine-generated tables
    ring = rh.Ring(vec["N"], vec["Qis"])
    for i, (a, b) in enumerate(zip(vec["poly"], vec["polyNTT"])):
        y = ring.SubRings[i].NTT(a)
        assert np.array_equal(y, np.array(b, dtype=np.uint64))
        assert np.array_equal(ring.SubRings[i].INTT(y), np.array(a, dtype=np.uint64))
        assert np.array_equal(ring.SubRings[i].INTTLazy(y), np.array(a, dtype=np.uint64))
    ring.close()


@pytest.mark.parametrize("vec", KAT, ids=lambda v: "N=%d" % v["N"])
def test_kat_batched_ring_ntt(rh, vec):
    # Ring.NTT on a device-resident 2-limb poly, batch of 3 copies
    ring = rh.Ring(vec["N"], vec["Qis"])
    a = np.array(vec["poly"], dtype=np.uint64)
    p = rh.DevicePoly.from_numpy(ring, np.stack([a, a, a]))
    ring.NTT(p, p)                       # in place, like ring/ntt_benchmark_test.go:42
    out = p.numpy()
    for k in range(3):
        assert np.array_equal(out[k], np.array(vec["polyNTT"], dtype=np.uint64))
    ring.INTT(p, p)
    assert np.array_equal(p.numpy()[1], a)
    ring.close()


@pytest.mark.parametrize("logN", [4, 6, 9, 11, 12, 13, 14, 15, 16])
def test_forward_inverse_lazy_vs_oracle(rh, oracle, logN):
    N = 1 << logN
    mods = QI60[:3]
    rng = np.random.default_rng(100 + logN)
    srs = [oracle.SubRingConsts(N, q) for q in mods]
    # hand the engine the oracle's (= reference-rule) constants, the way the Go shim hands over SubRing fields
    consts = dict(mred=[s.mred for s in srs], bred=np.stack([s.bred for s in srs]), ninv=[s.ninv for s in srs],
                  roots_fwd=np.stack([s.roots_fwd for s in srs]), roots_bwd=np.stack([s.roots_bwd for s in srs]))
    ring = rh.Ring(N, mods, constants=consts)
    npoly = 2
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(npoly)])
    # edge values in the first poly
    a[0, :, 0] = 0; a[0, :, 1] = np.array(mods, dtype=np.uint64) - np.uint64(1); a[0, :, 2] = 1
    p = rh.DevicePoly.from_numpy(ring, a)
    o = ring.NewPoly(npoly)
    ring.NTT(p, o)
    got = o.numpy()
    exp = np.stack([np.stack([oracle.ntt(a[k, i], srs[i]) for i in range(len(mods))]) for k in range(npoly)])
    assert np.array_equal(got, exp)
    # NTTLazy: exactly the reference's representatives (reduce schedule of nttUnrolled16Lazy)
    ring.NTTLazy(p, o)
    expl = np.stack([np.stack([oracle.ntt(a[k, i], srs[i], lazy=True) for i in range(len(mods))]) for k in range(npoly)])
    assert np.array_equal(o.numpy(), expl)
    # INTT / INTTLazy from the canonical NTT values
    pn = rh.DevicePoly.from_numpy(ring, exp)
    ring.INTT(pn, o)
    assert np.array_equal(o.numpy(), a)
    ring.INTTLazy(pn, o)
    assert np.array_equal(o.numpy(), a)
    # INTT accepts lazy inputs (< 2q): NTTLazy output reduced once is < 2q? use exp + q on some entries
    lazy_in = exp.copy()
    lazy_in[:, :, ::3] += np.array(mods, dtype=np.uint64)[None, :, None]
    pl = rh.DevicePoly.from_numpy(ring, lazy_in)
    ring.INTT(pl, o)
    assert np.array_equal(o.numpy(), a)
    # engine-generated constants are the same as the oracle's
    c = ring.constants()
    ring2 = rh.Ring(N, mods)
    c2 = ring2.constants()
    for k in ("mred", "bred", "ninv", "roots_fwd", "roots_bwd"):
        assert np.array_equal(np.asarray(c[k], dtype=np.uint64).reshape(-1), np.asarray(c2[k], dtype=np.uint64).reshape(-1)), k
    ring.close(); ring2.close()


def test_levels_use_leading_limbs(rh, oracle):
    # Ring.AtLevel(level).NTT touches limbs 0..level only (ring/ntt.go:127-131)
    N, mods = 4096, QI60[:4]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(7)
    a = np.stack([uniform_mod(rng, q, N) for q in mods[:2]])[None]
    p = rh.DevicePoly.from_numpy(ring, a)
    ring.AtLevel(1).NTT(p, p)
    srs = [oracle.SubRingConsts(N, q) for q in mods[:2]]
    assert np.array_equal(p.numpy()[0], np.stack([oracle.ntt(a[0, i], srs[i]) for i in range(2)]))
    ring.close()


@pytest.mark.parametrize("N", [256, 4096, 8192, 16384, 65536])
def test_at_level_views_are_batched_and_exact(rh, oracle, N):
    # rh_ring_*_rows on a batch: forward with one row stride on both sides and the inverse (N = 2^14..2^16) run as ONE batched launch
    # pair, the element-wise family as one launch with per-operand strides; other shapes fall back to poly-by-poly.  All against the oracle.
    mods = QI60[:4]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(N)
    B = 5
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    srs = [oracle.SubRingConsts(N, q) for q in mods]
    v = ring.AtLevel(2)                                            # three of the four limbs
    want = np.stack([np.stack([oracle.ntt(a[k, i], srs[i]) for i in range(3)]) for k in range(B)])
    pa = rh.DevicePoly.from_numpy(ring, a)
    po = rh.DevicePoly.from_numpy(ring, np.zeros_like(a))
    v.NTT(pa, po)                                                  # out of place, both blocks with 4 limbs per poly
    assert np.array_equal(po.numpy()[:, :3], want) and not po.numpy()[:, 3].any()
    assert np.array_equal(pa.numpy(), a)
    v.NTT(pa, pa)                                                  # in place
    assert np.array_equal(pa.numpy()[:, :3], want) and np.array_equal(pa.numpy()[:, 3], a[:, 3])
    dense = v.NewPoly(B)
    v.INTT(pa, dense)                                              # 4-limb block -> dense 3-limb block
    assert np.array_equal(dense.numpy(), a[:, :3])
    v.INTT(pa, pa)                                                 # in place on the 4-limb block
    assert np.array_equal(pa.numpy(), a)
    v.NTT(dense, po)                                               # dense -> 4-limb block (different strides: poly by poly)
    assert np.array_equal(po.numpy()[:, :3], want)
    v.NTTLazy(pa, po)                                              # exact lazy representatives, strided
    lz = po.numpy()[:, :3]
    for k in (0, B - 1):
        for i in range(3):
            assert np.array_equal(lz[k, i], oracle.ntt(a[k, i], srs[i], lazy=True))
    # element-wise: 4-limb x 3-limb -> 4-limb, then a scalar op into the dense block
    v.MulCoeffsMontgomery(pa, dense, po)
    g = po.numpy()
    for k in (0, 2, B - 1):
        for i in range(3):
            assert np.array_equal(g[k, i], oracle.vec_op(rh.OPS["MUL_MONT"], a[k, i], a[k, i], a[k, i], 0, 0, mods[i]))
    v.MulRNSScalarMontgomery(pa, [3, 5, 9], dense)
    for i in range(3):
        assert np.array_equal(dense.numpy()[B - 1, i], oracle.vec_op(rh.OPS["MUL_SCALAR_MONT"], a[B - 1, i], None, a[B - 1, i], [3, 5, 9][i], 0, mods[i]))
    ring.close()


def test_at_level_on_a_batch_with_more_limbs(rh, oracle):
    # ring.AtLevel(level) on max-level polys (ring/ring.go:192-213), the idiomatic use inside the reference's evaluators: NTT, INTT and
    # the element-wise family take blocks with more limbs per poly than the view's level (rh_ring_*_rows: limbs 0..level of every
    # poly processed, the others untouched); calls without a rows form refuse a BATCH instead of striding wrongly
    N, mods = 4096, QI60[:4]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(11)
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(3)])
    b = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(3)])
    srs = [oracle.SubRingConsts(N, q) for q in mods]
    v = ring.AtLevel(1)
    pa = rh.DevicePoly.from_numpy(ring, a)
    v.NTT(pa, pa)                                                  # in place on the max-level batch
    got = pa.numpy()
    for k in range(3):
        assert np.array_equal(got[k, :2], np.stack([oracle.ntt(a[k, i], srs[i]) for i in range(2)]))
        assert np.array_equal(got[k, 2:], a[k, 2:])               # limbs above the level untouched
    out2 = v.NewPoly(3)                                            # a block AT the level as the other operand
    v.INTT(pa, out2)
    assert np.array_equal(out2.numpy(), a[:, :2])
    pb = rh.DevicePoly.from_numpy(ring, b)
    v.Add(rh.DevicePoly.from_numpy(ring, a), pb, pb)               # element-wise: max-level blocks on all three operands
    gb = pb.numpy()
    for k in range(3):
        for i in range(2):
            assert np.array_equal(gb[k, i], oracle.vec_op(rh.OPS["ADD"], a[k, i], b[k, i], b[k, i], 0, 0, mods[i]))
        assert np.array_equal(gb[k, 2:], b[k, 2:])
    v.MulRNSScalarMontgomery(rh.DevicePoly.from_numpy(ring, a), [5, 7], out2)   # mixed: 4-limb input, 2-limb output
    for i in range(2):
        assert np.array_equal(out2.numpy()[2, i], oracle.vec_op(rh.OPS["MUL_SCALAR_MONT"], a[2, i], None, a[2, i], [5, 7][i], 0, mods[i]))
    two = rh.DevicePoly.from_numpy(ring, a)
    with pytest.raises(rh.RingHipError):                           # no rows form for the automorphisms: a batch is refused
        v.AutomorphismNTT(two, 5, rh.DevicePoly.from_numpy(ring, a))
    with pytest.raises(rh.RingHipError):
        ring.AtLevel(3).NTT(v.NewPoly(1), v.NewPoly(1))            # fewer limbs than the level: always an error
    ring.close()


def test_metric_size_properties(rh, oracle):
    # BASELINE metric size: N = 2^16, 16 limbs, batch of 4.  Full-size checks through size-independent properties:
    # (1) INTT(NTT(a)) == a, (2) linearity NTT(a+b) == NTT(a)+NTT(b) mod q, (3) one limb of one poly against the oracle,
    # (4) negacyclic convolution theorem on a sparse pair.
    N, mods = 1 << 16, QI60[:16]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(2026)
    B = 4
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    b = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    pa, pb = rh.DevicePoly.from_numpy(ring, a), rh.DevicePoly.from_numpy(ring, b)
    na, nb, ns = ring.NewPoly(B), ring.NewPoly(B), ring.NewPoly(B)
    ring.NTT(pa, na); ring.NTT(pb, nb)
    ring.Add(pa, pb, ns); ring.NTT(ns, ns)
    ring.Add(na, nb, na)
    assert np.array_equal(ns.numpy(), na.numpy())
    ring.INTT(nb, nb)
    assert np.array_equal(nb.numpy(), b)
    sr = oracle.SubRingConsts(N, mods[5])
    ring.NTT(pb, nb)
    assert np.array_equal(nb.numpy()[2, 5], oracle.ntt(b[2, 5], sr))
    assert np.array_equal(nb.numpy()[3, 15], oracle.ntt(b[3, 15], oracle.SubRingConsts(N, mods[15])))
    # X^i * X^j = X^(i+j) with sign flip past N (negacyclic): MForm + MulCoeffsMontgomery as schemes/ckks/evaluator.go:821-834
    x = np.zeros((1, 16, N), dtype=np.uint64); y = np.zeros((1, 16, N), dtype=np.uint64)
    x[0, :, N - 3] = 5; y[0, :, 7] = 11
    px, py = rh.DevicePoly.from_numpy(ring, x), rh.DevicePoly.from_numpy(ring, y)
    ring.NTT(px, px); ring.NTT(py, py); ring.MForm(px, px); ring.MulCoeffsMontgomery(px, py, px); ring.INTT(px, px)
    z = px.numpy()[0]
    for i, q in enumerate(mods):
        exp = np.zeros(N, dtype=np.uint64); exp[4] = q - 55
        assert np.array_equal(z[i], exp)
    ring.close()


def test_metric_full_batch_properties(rh, oracle):
    # the metric's full configuration (N = 2^16, 16 limbs, 1024 polys = 8 GiB per block, the pipelined fused launches): round trip and
    # linearity over the WHOLE batch (compared on the device), three (poly, limb) rows against the oracle
    import torch
    N, mods, B = 1 << 16, QI60[:16], 1024
    dev = torch.device("cuda", 0)
    ring = rh.Ring(N, mods)
    ring.set_stream(torch.cuda.current_stream().cuda_stream)
    qs = torch.tensor(mods, dtype=torch.int64, device=dev).view(1, 16, 1)
    g = torch.Generator(device=dev); g.manual_seed(99)
    a = torch.randint(0, 1 << 62, (B, 16, N), dtype=torch.int64, device=dev, generator=g) % qs
    b = torch.randint(0, 1 << 62, (B, 16, N), dtype=torch.int64, device=dev, generator=g) % qs
    na, nb = torch.empty_like(a), torch.empty_like(b)
    pa, pb, pna, pnb = (rh.DevicePoly.from_torch(ring, t) for t in (a, b, na, nb))
    ring.NTT(pa, pna); ring.NTT(pb, pnb)
    rows = [(0, 0), (517, 9), (1023, 15)]
    got = {r: na[r[0], r[1]].cpu().numpy().view(np.uint64) for r in rows}
    src = {r: a[r[0], r[1]].cpu().numpy().view(np.uint64) for r in rows}
    for (k, i) in rows:
        assert np.array_equal(got[(k, i)], oracle.ntt(src[(k, i)], oracle.SubRingConsts(N, mods[i])))
    ring.Add(pa, pb, pb)                              # b <- a + b (coefficient domain)
    ring.Add(pna, pnb, pnb)                           # nb <- NTT(a) + NTT(b)
    ring.NTT(pb, pb)                                  # b <- NTT(a + b), in place
    torch.cuda.synchronize()
    assert torch.equal(b, nb)
    ring.INTT(pna, pna)                               # back, in place
    torch.cuda.synchronize()
    assert torch.equal(na, a)
    ring.close()


def test_errors_match_reference_behaviour(rh):
    ring = rh.Ring(64, QI60[:1])
    with pytest.raises(rh.RingHipError):          # short slice -> panic in ring/ntt.go:212-214
        ring.SubRings[0].NTT(np.zeros(10, dtype=np.uint64))
    with pytest.raises(rh.RingHipError):
        ring.AtLevel(3)
    ring.close()
    with pytest.raises(rh.RingHipError):          # duplicate moduli rejected (ring/ring.go:326-331)
        rh.Ring(64, [QI60[0], QI60[0]])


@pytest.mark.parametrize("logN,chunk", [(13, 2), (16, 3)])
def test_fused_pipeline_spans_give_identical_results(rh, oracle, logN, chunk):
    # tuning knob "chunk_polys": software-pipelined spans (ntt_fwd_fused) must not change a single bit
    N, mods = 1 << logN, QI60[:2]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(logN)
    B = 7
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    p = rh.DevicePoly.from_numpy(ring, a)
    ref = ring.NewPoly(B)
    ring.set_tuning("one_pass", 0)                    # (N = 2^13 would take the one-pass kernel, which has no spans)
    ring.NTT(p, ref)
    ring.set_tuning("chunk_polys", chunk)
    ring.NTT(p, p)                                    # in place, 7 polys in spans of `chunk`
    assert np.array_equal(p.numpy(), ref.numpy())
    srs = [oracle.SubRingConsts(N, q) for q in mods]
    assert np.array_equal(p.numpy()[6, 1], oracle.ntt(a[6, 1], srs[1]))
    ring.close()


@pytest.mark.parametrize("logN,chunk", [(10, -1), (13, 2), (15, 3), (16, -1)])
def test_ntt_many_blocks_equal_separate_calls(rh, oracle, logN, chunk):
    # rh_ring_ntt_many: one software pipeline through several blocks (in place and out of place) == one Ring.NTT per block == oracle
    N, mods = 1 << logN, QI60[:3]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(100 + logN)
    sizes = [5, 1, 4]
    blocks = [np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(b)]) for b in sizes]
    refs = []
    for a in blocks:
        p, o = rh.DevicePoly.from_numpy(ring, a), ring.NewPoly(a.shape[0])
        ring.NTT(p, o); refs.append(o.numpy())
    if chunk > 0:
        ring.set_tuning("chunk_polys", chunk)
    else:
        ring.set_tuning("auto_span_rows", 6)          # spans of 2 polys: 10 polys in three blocks are pipelined
    if logN == 13:
        ring.set_tuning("one_pass", 0)                # the pipeline through several blocks is the two-pass rings' path
    ps = [rh.DevicePoly.from_numpy(ring, a) for a in blocks]
    outs = [ps[0], ring.NewPoly(sizes[1]), ps[2]]     # blocks 0 and 2 in place, block 1 out of place
    ring.NTTMany(list(zip(ps, outs)))
    for o, r in zip(outs, refs):
        assert np.array_equal(o.numpy(), r)
    srs = [oracle.SubRingConsts(N, q) for q in mods]
    assert np.array_equal(outs[2].numpy()[3, 2], oracle.ntt(blocks[2][3, 2], srs[2]))
    assert np.array_equal(ps[1].numpy(), blocks[1])   # the out-of-place input is untouched
    ring.NTTMany([])                                  # no block: nothing to do
    ring.close()


@pytest.mark.parametrize("logN", [12, 13, 16])
def test_asm_tile_kernel_equals_cxx_kernel(rh, oracle, logN):
    # the hand-scheduled forward tile kernel (default) and the C++ one must agree bit for bit, incl. worst-case inputs
    N, mods = 1 << logN, [QI60[0], QI60[15], 0x10000000006e0001 if False else QI60[7]]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(77 + logN)
    B = 3
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    a[0] = np.array(mods, dtype=np.uint64)[:, None] - np.uint64(1)       # all q-1: maximal lazy growth
    a[1, :, ::2] = 0
    p = rh.DevicePoly.from_numpy(ring, a)
    o1, o2 = ring.NewPoly(B), ring.NewPoly(B)
    ring.set_tuning("asm_tile", 1); ring.NTT(p, o1)
    ring.set_tuning("asm_tile", 0); ring.NTT(p, o2)
    assert np.array_equal(o1.numpy(), o2.numpy())
    srs = [oracle.SubRingConsts(N, q) for q in mods]
    for k in range(B):
        for i in range(3):
            assert np.array_equal(o1.numpy()[k, i], oracle.ntt(a[k, i], srs[i]))
    ring.close()


@pytest.mark.parametrize("logN,B", [(13, 3), (16, 5), (14, 300)])
def test_inverse_asm_and_pipeline_equal_cxx(rh, oracle, logN, B):
    # hand-scheduled inverse tile body (and, for B >= 256, the fused inverse pipeline) vs the C++ kernels, incl. lazy inputs < 2q
    N, mods = 1 << logN, [QI60[0], QI60[9]]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(logN + B)
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    a[0] = np.array(mods, dtype=np.uint64)[:, None] - np.uint64(1)
    a[1, :, ::2] += np.array(mods, dtype=np.uint64)[:, None]            # < 2q: accepted like the reference's invbutterfly
    p = rh.DevicePoly.from_numpy(ring, a)
    o1, o2 = ring.NewPoly(B), ring.NewPoly(B)
    ring.set_tuning("one_pass", 0)
    ring.set_tuning("asm_tile", 1); ring.INTT(p, o1)
    ring.set_tuning("asm_tile", 0); ring.INTT(p, o2)
    x1 = o1.numpy()
    assert np.array_equal(x1, o2.numpy())
    srs = [oracle.SubRingConsts(N, q) for q in mods]
    for k in (0, 1, B - 1):
        for i in range(2):
            assert np.array_equal(x1[k, i], oracle.intt(a[k, i] % np.uint64(mods[i]), srs[i]))
    ring.set_tuning("asm_tile", 1); ring.INTT(p, p)                    # in place
    assert np.array_equal(p.numpy(), x1)
    ring.close()


@pytest.mark.parametrize("logN,L,B", [(16, 3, 2), (16, 16, 5), (16, 2, 1100), (15, 3, 3), (15, 2, 1100), (14, 5, 2), (14, 2, 1100)])
def test_asm_column_stages_identical(rh, oracle, logN, L, B):
    # N = 2^14..2^16: hand-scheduled column stages (standalone launch for small batches, fused launches for big ones) vs the C++ body
    N, mods = 1 << logN, QI60[:L]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(L * 31 + B)
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    p = rh.DevicePoly.from_numpy(ring, a)
    ref, o = ring.NewPoly(B), ring.NewPoly(B)
    ring.set_tuning("one_pass", 0)                      # (N = 2^14 would take the one-pass kernel: this test is about the two-pass launches)
    ring.set_tuning("asm_cols", 0); ring.NTT(p, ref)
    ring.set_tuning("asm_cols", 1); ring.NTT(p, o); ring.sync()
    assert np.array_equal(o.numpy(), ref.numpy())
    ring.NTT(p, p); ring.sync()
    assert np.array_equal(p.numpy(), ref.numpy())
    sr = oracle.SubRingConsts(N, mods[L - 1])
    assert np.array_equal(p.numpy()[B - 1, L - 1], oracle.ntt(a[B - 1, L - 1], sr))
    ring.close()


@pytest.mark.parametrize("logN,L,B", [(16, 3, 2), (16, 16, 5), (16, 2, 1100), (15, 3, 3), (15, 2, 1100), (14, 5, 2), (14, 2, 1100)])
def test_asm_inverse_column_stages_identical(rh, oracle, logN, L, B):
    # N = 2^14..2^16: hand-scheduled inverse column stages with N^-1 folded in (standalone and fused launches) vs the C++ body
    N, mods = 1 << logN, QI60[:L]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(L * 17 + B)
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    p = rh.DevicePoly.from_numpy(ring, a)
    ref, o = ring.NewPoly(B), ring.NewPoly(B)
    ring.set_tuning("one_pass", 0)
    ring.set_tuning("asm_cols", 0); ring.INTT(p, ref)
    ring.set_tuning("asm_cols", 1); ring.INTT(p, o); ring.sync()
    assert np.array_equal(o.numpy(), ref.numpy())
    sr = oracle.SubRingConsts(N, mods[L - 1])
    assert np.array_equal(o.numpy()[B - 1, L - 1], oracle.intt(a[B - 1, L - 1], sr))
    ring.INTT(p, p); ring.NTT(p, p); ring.sync()        # in place, and back
    assert np.array_equal(p.numpy(), a)
    ring.close()


@pytest.mark.parametrize("logN,L,B", [(13, 3, 3), (14, 3, 3), (13, 16, 40), (14, 5, 300), (13, 2, 4200), (14, 2, 2100)])
def test_one_pass_kernels_identical_to_two_pass(rh, oracle, logN, L, B):
    # N = 2^13 / 2^14: the whole limb row in one workgroup's LDS (ntt_fwd_onepass_asm / ntt_inv_onepass_asm, the default) against the two-pass
    # launches and the oracle: worst-case inputs, in place, inverse inputs < 2q, and batches >= 512 MiB (the non-temporal bodies)
    N, mods = 1 << logN, QI60[:L]
    ring = rh.Ring(N, mods)
    rng = np.random.default_rng(1000 * logN + L + B)
    qv = np.array(mods, dtype=np.uint64)[:, None]
    if B <= 300:
        a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    else:                                                  # big batch: a few random polys, tiled
        base = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(7)])
        a = np.concatenate([base] * (B // 7 + 1))[:B].copy()
    a[0] = qv - np.uint64(1)
    a[1, :, ::2] = 0
    p = rh.DevicePoly.from_numpy(ring, a)
    f1, f2 = ring.NewPoly(B), ring.NewPoly(B)
    ring.set_tuning("one_pass", 1); ring.NTT(p, f1)
    ring.set_tuning("one_pass", 0); ring.NTT(p, f2); ring.sync()
    x1 = f1.numpy()
    assert np.array_equal(x1, f2.numpy())
    srs = [oracle.SubRingConsts(N, q) for q in mods]
    for k in (0, 1, B - 1):
        for i in (0, L - 1):
            assert np.array_equal(x1[k, i], oracle.ntt(a[k, i], srs[i]))
    # inverse: canonical inputs and inputs < 2q
    y = x1.copy()
    y[2, :, 1::2] += qv
    py = rh.DevicePoly.from_numpy(ring, y)
    g1, g2 = ring.NewPoly(B), ring.NewPoly(B)
    ring.set_tuning("one_pass", 1); ring.INTT(py, g1)
    ring.set_tuning("one_pass", 0); ring.INTT(py, g2); ring.sync()
    z1 = g1.numpy()
    assert np.array_equal(z1, g2.numpy())
    assert np.array_equal(z1, a)
    # in place, forward then back
    ring.set_tuning("one_pass", 1)
    ring.NTT(p, p); ring.sync()
    assert np.array_equal(p.numpy(), x1)
    ring.INTT(p, p); ring.sync()
    assert np.array_equal(p.numpy(), a)
    ring.close()
