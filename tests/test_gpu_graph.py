"""GPU: the library's launches are capturable into a HIP graph by the caller (scalars travel by value, no allocation or
synchronous copy on a warmed-up handle): a whole key switch replayed as ONE graph launch gives the same bits as the
direct calls.  Capture is the caller's (here: torch.cuda.CUDAGraph on the stream handed to rh_ring_set_stream)."""
import numpy as np
import pytest

from conftest import QI60, PI60, uniform_mod

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("small_rows", [0, 512], ids=["digit-pipeline", "all-digits-per-launch"])
def test_key_switch_and_polymul_replayed_from_a_graph(rh, small_rows):
    import torch
    N, nq, np_, B = 4096, 6, 2, 2
    Q, P = QI60[:nq], PI60[:np_]
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    rq.set_tuning("ks_small_rows", small_rows)       # 0: the large-batch launch sequence; 512 (the default): the small-batch one
    be = rh.BasisExtender(rq, rp)
    rng = np.random.default_rng(11)
    beta = (nq - 1 + np_) // np_
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
    mkq = lambda n: np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(n)])
    mkp = lambda n: np.stack([np.stack([uniform_mod(rng, p, N) for p in P]) for _ in range(n)])
    cx_a, cx_b = mkq(B), mkq(B)
    tcx, tkq, tkp = t(cx_a), t(mkq(beta * 2)), t(mkp(beta * 2))
    tc0, tc1, tm = torch.zeros_like(tcx), torch.zeros_like(tcx), torch.zeros_like(tcx)
    pcx, pc0, pc1, pm = (rh.DevicePoly.from_torch(rq, x) for x in (tcx, tc0, tc1, tm))

    def work():
        be.GadgetProduct(nq - 1, np_ - 1, pcx, tkq.data_ptr(), tkp.data_ptr(), beta, pc0, pc1)
        rq.MForm(pc0, pm); rq.MulCoeffsMontgomery(pm, pc1, pm); rq.INTT(pm, pm)      # a few more ring calls behind it
        rq.MulRNSScalarMontgomery(pm, [3] * nq, pm)                                    # a by-value scalar op

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for r in (rq, rp):
            r.set_stream(side.cuda_stream)
        work()                                       # warm-up on the capture stream: plans, scratch and tables get allocated here
        side.synchronize()
        direct_a = [x.clone() for x in (tc0, tc1, tm)]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            work()
        for x in (tc0, tc1, tm):
            x.zero_()
        g.replay(); side.synchronize()
        assert all(torch.equal(x, y) for x, y in zip((tc0, tc1, tm), direct_a))
        tcx.copy_(t(cx_b))                           # new input, same buffers: replay again, then compare with direct calls
        g.replay(); side.synchronize()
        replay_b = [x.clone() for x in (tc0, tc1, tm)]
        work(); side.synchronize()
        assert all(torch.equal(x, y) for x, y in zip((tc0, tc1, tm), replay_b))
        assert not torch.equal(replay_b[0], direct_a[0])
    be.close(); rq.close(); rp.close()
