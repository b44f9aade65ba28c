"""GPU: the BASELINE configs at their REAL sizes against the oracle (VERDICT r01 "configs untested").

  * config 5 (N = 2^16, Q = Qi60[0:24], P = Pi60[0:6], beta = 4): GadgetProduct, GadgetProductHoisted, ...ThenAdd, the
    limb-sharded key switch (1 rank) and CKKS MulRelin + Rescale vs the oracle composition of tests/test_gpu_keyswitch.py.
    N = 2^14 and 2^15 as well, so the hand-scheduled column stages ntt_fwd_cols_asm<2..4> run with a row stride Ls != L
    (rh_std_ntt_fwd_strided: the digit's own limbs are skipped) against the oracle.
  * the headline launch shape: Ring.NTT / INTT at L = 16 with B = 260 polys (fused column + tile spans of 128 polys,
    ntt_fwd_fused_asm<4, true> / ntt_inv_fused_asm<4, true>) spot-checked on limbs 0, 7, 15 of polys 0, 129, 259.
  * config 4 ring (N = 3*2^16) with all 24 limbs, batch 2: forward / inverse 3N transform and matrix_ckks.Evaluator.Mul.
The oracle costs ~12 ms per limb transform at N = 2^16, so every case here is seconds of CPU."""
import numpy as np
import pytest

from conftest import QI60, PI60, uniform_mod
from test_gpu_keyswitch import oracle_gadget_product

pytestmark = pytest.mark.gpu


def _ks_case(N, nq, np_, npoly, seed):
    Q, P = QI60[:nq], PI60[:np_]
    rng = np.random.default_rng(seed)
    beta = (nq - 1 + np_) // np_
    evkQ = np.stack([np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(2)]) for _ in range(beta)])
    evkP = np.stack([np.stack([np.stack([uniform_mod(rng, p, N) for p in P]) for _ in range(2)]) for _ in range(beta)])
    mk = lambda: np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(npoly)])
    return Q, P, beta, evkQ, evkP, mk


@pytest.mark.parametrize("logN", [14, 15, 16])
def test_config5_gadget_product_real_size(rh, oracle, logN):
    """rlwe.Evaluator.GadgetProduct (core/rlwe/evaluator_gadget_product.go:16-30, 123-188, 455-478), Q = 24, P = 6"""
    N, nq, np_, npoly = 1 << logN, 24, 6, 3
    Q, P, beta, evkQ, evkP, mk = _ks_case(N, nq, np_, npoly, 1000 + logN)
    assert beta == 4
    cx, a0, a1 = mk(), mk(), mk()
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    ev = rh.rlwe.Evaluator(rq, rp)
    gct = rh.rlwe.GadgetCiphertext(rq, rp, evkQ, evkP)
    levelQ, levelP = nq - 1, np_ - 1
    pcx = rh.DevicePoly.from_numpy(rq, cx)
    direct = rh.Ciphertext([rq.NewPoly(npoly), rq.NewPoly(npoly)], is_ntt=True)
    ev.GadgetProduct(levelQ, pcx, gct, direct)
    d0, d1 = direct.Value[0].numpy(), direct.Value[1].numpy()
    exp = {}
    for k in (0, npoly - 1):                                       # first and last poly of the batch, every limb
        exp[k] = oracle_gadget_product(oracle, rh, N, Q, P, levelQ, levelP, cx[k], evkQ, evkP)
        assert np.array_equal(d0[k], exp[k][0]), "component 0, poly %d" % k
        assert np.array_equal(d1[k], exp[k][1]), "component 1, poly %d" % k
    assert np.array_equal(pcx.numpy(), cx)                         # input untouched
    # hoisted (:326-349, 373-453) == direct, bit for bit
    dq_dp = ev.DecomposeNTT(levelQ, levelP, pcx, True)
    h = rh.Ciphertext([rq.NewPoly(npoly), rq.NewPoly(npoly)], is_ntt=True)
    ev.GadgetProductHoisted(levelQ, dq_dp, gct, h)
    assert np.array_equal(h.Value[0].numpy(), d0) and np.array_equal(h.Value[1].numpy(), d1)
    del dq_dp
    # ...ThenAdd: the ring.Add of Relinearize / Automorphism in ModDown's epilogue
    t = rh.Ciphertext([rq.NewPoly(npoly), rq.NewPoly(npoly)], is_ntt=True)
    ev.GadgetProductThenAdd(levelQ, pcx, gct, rh.DevicePoly.from_numpy(rq, a0), rh.DevicePoly.from_numpy(rq, a1), t)
    t0, t1 = t.Value[0].numpy(), t.Value[1].numpy()
    ADD = rh.OPS["ADD"]
    for k in exp:
        for i, q in enumerate(Q):
            assert np.array_equal(t0[k, i], oracle.vec_op(ADD, a0[k, i], exp[k][0][i], exp[k][0][i], 0, 0, q))
            assert np.array_equal(t1[k, i], oracle.vec_op(ADD, a1[k, i], exp[k][1][i], exp[k][1][i], 0, 0, q))
    ev.close(); rq.close(); rp.close()


def test_config5_gadget_product_benchmark_batch(rh, oracle):
    """the batch the key-switch numbers are quoted on (64 polys, N = 2^16, Q = 24, P = 6: the multi-poly key multiply-accumulate
    workgroups, the pipelined digit transforms without their final reduction): polys 0, 31, 63 of the batch against the oracle, and the
    whole batch against the hoisted product of the same input"""
    import torch
    N, nq, np_, B = 1 << 16, 24, 6, 64
    Q, P, beta, evkQ, evkP, _mk = _ks_case(N, nq, np_, 1, 4242)
    dev = torch.device("cuda", 0)
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    for r in (rq, rp):
        r.set_stream(torch.cuda.current_stream().cuda_stream)
    ev = rh.rlwe.Evaluator(rq, rp)
    gct = rh.rlwe.GadgetCiphertext(rq, rp, evkQ, evkP)
    qs = torch.tensor(Q, dtype=torch.int64, device=dev).view(1, nq, 1)
    g = torch.Generator(device=dev); g.manual_seed(5)
    cx = torch.randint(0, 1 << 62, (B, nq, N), dtype=torch.int64, device=dev, generator=g) % qs
    o0, o1, h0, h1 = (torch.empty_like(cx) for _ in range(4))
    pcx = rh.DevicePoly.from_torch(rq, cx)
    direct = rh.Ciphertext([rh.DevicePoly.from_torch(rq, o0), rh.DevicePoly.from_torch(rq, o1)], is_ntt=True)
    ev.GadgetProduct(nq - 1, pcx, gct, direct)
    torch.cuda.synchronize()
    for k in (0, 31, 63):
        e0, e1 = oracle_gadget_product(oracle, rh, N, Q, P, nq - 1, np_ - 1, cx[k].cpu().numpy().view(np.uint64), evkQ, evkP)
        assert np.array_equal(o0[k].cpu().numpy().view(np.uint64), e0) and np.array_equal(o1[k].cpu().numpy().view(np.uint64), e1)
    dec = ev.DecomposeNTT(nq - 1, np_ - 1, pcx, True)
    hoisted = rh.Ciphertext([rh.DevicePoly.from_torch(rq, h0), rh.DevicePoly.from_torch(rq, h1)], is_ntt=True)
    ev.GadgetProductHoisted(nq - 1, dec, gct, hoisted)
    torch.cuda.synchronize()
    assert torch.equal(h0, o0) and torch.equal(h1, o1)
    ev.close(); rq.close(); rp.close()


def test_config5_limb_sharded_one_rank_real_size(rh, oracle):
    """sharding.LimbShardedKeySwitch (rh_kshard_*) with one rank at N = 2^16, Q = 24, P = 6 vs the oracle composition"""
    import torch
    from matrix_fhe_lattigo_amd import sharding
    N, nq, np_, npoly = 1 << 16, 24, 6, 2
    Q, P, beta, evkQ, evkP, mk = _ks_case(N, nq, np_, npoly, 77)
    cx = mk()
    ks = sharding.LimbShardedKeySwitch(N, Q, P, 0, 1, dist=None)
    kq, kp = ks.shard_key(evkQ, evkP)
    dcx, dkq, dkp = ks.to_device(ks.shard_q(cx)), ks.to_device(kq), ks.to_device(kp)
    ct0, ct1 = torch.empty_like(dcx), torch.empty_like(dcx)
    ks.GadgetProduct(dcx, dkq, dkp, ct0, ct1)
    torch.cuda.synchronize()
    g0, g1 = ct0.cpu().numpy().view(np.uint64), ct1.cpu().numpy().view(np.uint64)
    e0, e1 = oracle_gadget_product(oracle, rh, N, Q, P, nq - 1, np_ - 1, cx[1], evkQ, evkP)
    assert np.array_equal(g0[1], e0) and np.array_equal(g1[1], e1)
    ks.close()


def test_config5_ckks_mulrelin_rescale_real_size(rh, oracle):
    """ckks.Evaluator.MulRelin + Rescale (schemes/ckks/evaluator.go:786-881, 500-535) at N = 2^16, Q = 24, P = 6"""
    from test_gpu_ckks import oracle_tensor, vop
    N, nq, np_, B = 1 << 16, 24, 6, 2
    Q, P, beta, evkQ, evkP, mk = _ks_case(N, nq, np_, B, 4242)
    a, b = np.stack([mk(), mk()]), np.stack([mk(), mk()])          # (component, poly, limb, N)
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    rlk = rh.rlwe.GadgetCiphertext(rq, rp, evkQ, evkP)
    ev = rh.ckks.Evaluator(rq, rp, rlk=rlk)
    dp = lambda x: rh.DevicePoly.from_numpy(rq, x)
    ct0, ct1 = rh.Ciphertext([dp(a[0]), dp(a[1])], is_ntt=True), rh.Ciphertext([dp(b[0]), dp(b[1])], is_ntt=True)
    out = rh.Ciphertext([rq.NewPoly(B), rq.NewPoly(B)], is_ntt=True)
    ev.MulRelin(ct0, ct1, out, relin=True)
    got = [v.numpy() for v in out.Value]
    res = rh.Ciphertext([rq.NewPoly(B), rq.NewPoly(B)], is_ntt=True)
    ev.Rescale(out, res)
    gres = [v.numpy() for v in res.Value]
    srQ = [oracle.SubRingConsts(N, q) for q in Q]
    k = 1
    t = oracle_tensor(oracle, rh, a[:, k], b[:, k], Q)
    g0, g1 = oracle_gadget_product(oracle, rh, N, Q, P, nq - 1, np_ - 1, t[2], evkQ, evkP)
    for c, g in ((0, g0), (1, g1)):
        e = np.stack([vop(oracle, rh, "ADD", t[c][i], g[i], g[i], Q[i]) for i in range(nq)])
        assert np.array_equal(got[c][k], e), "MulRelin component %d" % c
        coeff = np.stack([oracle.intt(e[i], srQ[i]) for i in range(nq)])
        down = oracle.div_by_last_modulus_many(coeff, Q, 1, True)
        want = np.stack([oracle.ntt(down[i], srQ[i]) for i in range(nq - 1)])
        assert np.array_equal(gres[c][k, :nq - 1], want), "Rescale component %d" % c
    ev.close(); rq.close(); rp.close()


def test_headline_launch_shape_vs_oracle(rh, oracle):
    """Ring.NTT / Ring.INTT at the metric's shape (N = 2^16, L = 16) with a batch that takes the fused pipelined launches
    (260 polys = spans of 128 + 128 + 4): limbs 0, 7, 15 of polys 0, 129, 259 against the oracle, both directions."""
    import torch
    N, L, B = 1 << 16, 16, 260
    mods = QI60[:L]
    ring = rh.Ring(N, mods, device=0)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(260)
    qs = torch.tensor(mods, dtype=torch.int64, device=dev).view(1, L, 1)
    data = torch.randint(0, 1 << 62, (B, L, N), dtype=torch.int64, device=dev, generator=g) % qs
    src = data.clone()
    poly = rh.DevicePoly.from_torch(ring, data)
    ring.set_stream(torch.cuda.current_stream().cuda_stream)
    ring.NTT(poly, poly)
    torch.cuda.synchronize()
    srs = {i: oracle.SubRingConsts(N, mods[i]) for i in (0, 7, 15)}
    for p in (0, 129, 259):
        for i in (0, 7, 15):
            x = src[p, i].cpu().numpy().view(np.uint64)
            assert np.array_equal(data[p, i].cpu().numpy().view(np.uint64), oracle.ntt(x, srs[i])), "NTT poly %d limb %d" % (p, i)
    # inverse on fresh NTT-domain-looking data (any residues are valid inputs): out of place, then the round trip
    out = torch.empty_like(data)
    ring.INTT(poly, rh.DevicePoly.from_torch(ring, out))
    torch.cuda.synchronize()
    for p in (0, 129, 259):
        for i in (0, 7, 15):
            y = data[p, i].cpu().numpy().view(np.uint64)
            assert np.array_equal(out[p, i].cpu().numpy().view(np.uint64), oracle.intt(y, srs[i])), "INTT poly %d limb %d" % (p, i)
    assert torch.equal(out, src)
    ring.close()


@pytest.mark.parametrize("N", [3 << 16, 3 << 14], ids=["N=3*2^16", "N=3*2^14"])
def test_config4_ring_all_24_limbs(rh, oracle, N):
    """config 4: 3N ring with all 24 moduli at BOTH readings of its "logN = 16" (logN := Order2 gives N = 3*2^16, the headline;
    N = 3*2^14 = 49152 is the other), batch 2: forward / inverse transform on 3 spot limbs vs the oracle's
    fast restatement (itself pinned to integer_dft.py fixtures and the Horner definition in tests/test_oracle_ntt3n.py),
    round trip on every limb, and matrix_ckks.Evaluator.Mul (degree 1 x degree 1) vs the oracle call sequence
    (schemes/matrix_ckks/evaluator.go:114-192) on the same limbs.  Mul end-to-end stays parity-unpinned by the reference."""
    from test_gpu_schemes import primes_3n, omega_for
    L, B = 24, 2
    mods = primes_3n(oracle, N, L)
    om = [omega_for(q, N) for q in mods]
    ring = rh.Ring(N, mods, kind=rh.Matrix3N, omega3n=om)
    rng = np.random.default_rng(2424)
    mk = lambda: np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])
    x0, x1, y0, y1 = mk(), mk(), mk(), mk()
    p = rh.DevicePoly.from_numpy(ring, x0)
    ring.NTT(p, p)
    f = p.numpy()
    spots = (0, 11, 23)
    fx = {}
    for i in spots:
        fx[i] = oracle.ntt3n_forward(x0[1, i], mods[i], om[i])
        assert np.array_equal(f[1, i], fx[i]), "3N forward limb %d" % i
    ring.INTT(p, p)
    assert np.array_equal(p.numpy(), x0)
    for i in spots:
        assert np.array_equal(oracle.ntt3n_backward(fx[i], mods[i], om[i]), x0[1, i])
    # Mul: NTT the four inputs, tensor without MForm (values carry the 2^-64 factor), INTT the three outputs
    dp = lambda a: rh.DevicePoly.from_numpy(ring, a)
    ev = rh.MatrixCKKSEvaluator(ring, block_order=False)       # the reference's NTT-domain order throughout
    out = rh.Ciphertext([ring.NewPoly(B) for _ in range(3)])
    ev.Mul(rh.Ciphertext([dp(x0), dp(x1)]), rh.Ciphertext([dp(y0), dp(y1)]), out)
    got = [v.numpy() for v in out.Value]
    MUL, MULADD = rh.OPS["MUL_MONT"], rh.OPS["MUL_MONT_THEN_ADD"]
    z = np.zeros(N, dtype=np.uint64)
    for i in spots:
        q, w = mods[i], om[i]
        A0, A1, B0, B1 = (oracle.ntt3n_forward(v[1, i], q, w) for v in (x0, x1, y0, y1))
        c0 = oracle.vec_op(MUL, A0, B0, z, 0, 0, q)
        c2 = oracle.vec_op(MUL, A1, B1, z, 0, 0, q)
        c1 = oracle.vec_op(MULADD, A1, B0, oracle.vec_op(MUL, A0, B1, z, 0, 0, q), 0, 0, q)
        for c, e in ((0, c0), (1, c1), (2, c2)):
            assert np.array_equal(got[c][1, i], oracle.ntt3n_backward(e, q, w)), "Mul component %d limb %d" % (c, i)
    # the DEFAULT evaluator: device NTT domain in block order, tagged per block (no permutation pass): identical coefficient-domain output,
    # and the NTT-domain inputs it left behind read back in the reference's order
    ev2 = rh.MatrixCKKSEvaluator(ring)
    assert ring.ntt3n_layout == "block"
    out2 = rh.Ciphertext([ring.NewPoly(B) for _ in range(3)])
    cx = rh.Ciphertext([dp(x0), dp(x1)])
    ev2.Mul(cx, rh.Ciphertext([dp(y0), dp(y1)]), out2)
    assert cx.IsNTT and cx.Value[0].layout == "block" and out2.Value[0].layout is None
    for c in range(3):
        assert np.array_equal(out2.Value[c].numpy(), got[c]), "block-order Mul component %d" % c
    assert np.array_equal(cx.Value[0].numpy()[1, 11], fx[11])
    ring.close()


def test_config4_per_gpu_share_128_ciphertext_pairs(rh, oracle):
    """BASELINE config 4 at the share ONE GPU gets of its 1024 ciphertext pairs over 8 GPUs: 128 pairs, N = 3*2^16, 24 limbs (4.5 GiB per
    128-poly component; 14 components live = 63 GiB of HBM).  matrix_ckks.Evaluator.Mul with the default (block-order, tagged) device NTT
    domain: the WHOLE batch against the reference-order run on the device, and spot rows of spot ciphertexts against the oracle call sequence
    (schemes/matrix_ckks/evaluator.go:114-192).  Any span / stride / size_t issue at this size shows as a mismatch on the last ciphertexts."""
    import torch
    from test_gpu_schemes import primes_3n, omega_for
    N, L, B = 3 << 16, 24, 128
    mods = primes_3n(oracle, N, L)
    om = [omega_for(q, N) for q in mods]
    ring = rh.Ring(N, mods, kind=rh.Matrix3N, omega3n=om)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(4128)
    qs = torch.tensor(mods, dtype=torch.int64, device=dev).view(1, L, 1)

    def draw():
        t = torch.empty((B, L, N), dtype=torch.int64, device=dev)
        for b0 in range(0, B, 16):
            t[b0:b0 + 16] = torch.randint(0, 1 << 62, (16, L, N), dtype=torch.int64, device=dev, generator=g) % qs
        return t
    src = [draw() for _ in range(4)]                                   # x0, x1, y0, y1
    spots_k, spots_i = (0, 77, B - 1), (0, 13, L - 1)
    host = {(c, k, i): src[c][k, i].cpu().numpy().view(np.uint64).copy() for c in range(4) for k in spots_k for i in spots_i}
    dp = lambda t: rh.DevicePoly.from_torch(ring, t)
    res = {}
    for name, blk in (("reference", False), ("block", True)):
        ev = rh.MatrixCKKSEvaluator(ring, block_order=blk)
        work = [t.clone() for t in src]                                # Mul transforms its inputs in place (:136-149)
        outs = [torch.empty((B, L, N), dtype=torch.int64, device=dev) for _ in range(3)]
        ev.Mul(rh.Ciphertext([dp(work[0]), dp(work[1])]), rh.Ciphertext([dp(work[2]), dp(work[3])]), rh.Ciphertext([dp(o) for o in outs]))
        torch.cuda.synchronize()
        res[name] = outs
        del work
    for c in range(3):
        assert torch.equal(res["block"][c], res["reference"][c]), "component %d: block-order run differs from the reference-order run" % c
    MUL, MULADD = rh.OPS["MUL_MONT"], rh.OPS["MUL_MONT_THEN_ADD"]
    z = np.zeros(N, dtype=np.uint64)
    for k in spots_k:
        for i in spots_i:
            q, w = mods[i], om[i]
            A0, A1, B0, B1 = (oracle.ntt3n_forward(host[(c, k, i)], q, w) for c in range(4))
            e = [oracle.vec_op(MUL, A0, B0, z, 0, 0, q),
                 oracle.vec_op(MULADD, A1, B0, oracle.vec_op(MUL, A0, B1, z, 0, 0, q), 0, 0, q),
                 oracle.vec_op(MUL, A1, B1, z, 0, 0, q)]
            for c in range(3):
                assert np.array_equal(res["block"][c][k, i].cpu().numpy().view(np.uint64), oracle.ntt3n_backward(e[c], q, w)), (c, k, i)
    ring.close()


def test_working_sets_beyond_the_infinity_cache_take_the_non_temporal_bodies(rh, oracle):
    """launches whose rows exceed 512 MiB run the generated bodies with non-temporal data streams (engine.hip: rh_streams_beyond_cache; the same
    instructions with a cache-policy hint): a two-launch forward / inverse transform (70 polys x 16 limbs at N = 2^16 = 560 MiB, not pipelined)
    and DivRoundByLastModulusNTT on the same block (re-expansion column stages + subtract-multiply tile stages), spot rows against the oracle"""
    import torch
    N, L, B = 1 << 16, 16, 70
    Q = QI60[:L]
    dev = torch.device("cuda", 0)
    ring = rh.Ring(N, Q)
    g = torch.Generator(device=dev); g.manual_seed(99)
    qs = torch.tensor(Q, dtype=torch.int64, device=dev).view(1, L, 1)
    x = torch.randint(0, 1 << 62, (B, L, N), dtype=torch.int64, device=dev, generator=g) % qs
    host = {k: x[k].cpu().numpy().view(np.uint64).copy() for k in (0, 37, B - 1)}
    p = rh.DevicePoly.from_torch(ring, x)
    ring.NTT(p, p)
    srs = [oracle.SubRingConsts(N, q) for q in Q]
    for k, a in host.items():
        for i in (0, 9, L - 1):
            assert np.array_equal(x[k, i].cpu().numpy().view(np.uint64), oracle.ntt(a[i], srs[i])), (k, i)
    out = torch.zeros((B, L, N), dtype=torch.int64, device=dev)
    po = rh.DevicePoly.from_torch(ring, out)
    ring.DivRoundByLastModulusNTT(p, po)
    for k, a in host.items():
        down = oracle.div_by_last_modulus_many(a, Q, 1, True)
        for i in (0, L - 2):
            assert np.array_equal(out[k, i].cpu().numpy().view(np.uint64), oracle.ntt(down[i], srs[i])), ("rescale", k, i)
    ring.INTT(p, p)
    for k, a in host.items():
        assert np.array_equal(x[k].cpu().numpy().view(np.uint64), a), ("round trip", k)
    ring.close()
