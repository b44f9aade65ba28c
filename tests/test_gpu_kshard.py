"""GPU: limb-sharded hybrid key switch against the unsharded rh_bext_gadget_product, which
tests/test_gpu_keyswitch.py pins to the oracle composition.  Multi-rank cases run as separate processes that share
the box's one GPU and exchange limbs over gloo (host-staged); on a multi-GPU node the same class uses RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import QI60, PI60

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(N, nq, np_, npoly, seed):
    Q, P = QI60[:nq], PI60[:np_]
    rng = np.random.default_rng(seed)
    beta = (nq - 1 + np_) // np_
    u = lambda q, shape: (rng.integers(0, 1 << 62, size=shape, dtype=np.uint64) % np.uint64(q))
    cx = np.stack([np.stack([u(q, N) for q in Q]) for _ in range(npoly)])
    evkQ = np.stack([np.stack([np.stack([u(q, N) for q in Q]) for _ in range(2)]) for _ in range(beta)])
    evkP = np.stack([np.stack([np.stack([u(p, N) for p in P]) for _ in range(2)]) for _ in range(beta)])
    return Q, P, beta, cx, evkQ, evkP


def _unsharded(rh, N, Q, P, beta, cx, evkQ, evkP):
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    be = rh.BasisExtender(rq, rp)
    npoly, nq, np_ = cx.shape[0], len(Q), len(P)
    pcx = rh.DevicePoly.from_numpy(rq, cx)
    dq = rh.DevicePoly.from_numpy(rq, evkQ.reshape(beta * 2, nq, N))
    dp = rh.DevicePoly.from_numpy(rp, evkP.reshape(beta * 2, np_, N))
    ct0, ct1 = rh.DevicePoly(rq, npoly, nq), rh.DevicePoly(rq, npoly, nq)
    be.GadgetProduct(nq - 1, np_ - 1, pcx, dq.ptr, dp.ptr, beta, ct0, ct1)
    out = ct0.numpy(), ct1.numpy()
    be.close(); rq.close(); rp.close()
    return out


def _run_shard(rh, sharding, N, Q, P, cx, evkQ, evkP, rank, world, dist):
    import torch
    ks = sharding.LimbShardedKeySwitch(N, Q, P, rank, world, dist=dist)
    kq, kp = ks.shard_key(evkQ, evkP)
    dcx = ks.to_device(ks.shard_q(cx))
    dkq = ks.to_device(kq)
    dkp = ks.to_device(kp) if kp is not None else None
    ct0, ct1 = torch.empty_like(dcx), torch.empty_like(dcx)
    npoly = dcx.shape[0]
    # the whole product behind the C ABI (rh_kshard_gadget_product): this host only supplies the all-gather (a ctypes callback here)
    ks.GadgetProduct(dcx, dkq, dkp, ct0, ct1)
    auto = 4 if (world > 1 and npoly >= 4) else 1
    chunks_run = len(range(0, npoly, -(-npoly // auto)))
    assert ks.exchanges == (2 * chunks_run if world > 1 else 0), (ks.exchanges, chunks_run)
    d0, d1 = torch.empty_like(dcx), torch.empty_like(dcx)
    for ch in (1, 2, 3):                                                # chunk pipelines on two side streams: same bits
        d0.zero_(); d1.zero_()
        ks.GadgetProduct(dcx, dkq, dkp, d0, d1, chunks=ch)
        torch.cuda.synchronize()
        assert torch.equal(d0, ct0) and torch.equal(d1, ct1), ch
    ks.GadgetProduct(dcx, dkq, dkp, d0, d1, orchestrate="python")       # round 2's host-side sequence of the same calls
    torch.cuda.synchronize()
    assert torch.equal(d0, ct0) and torch.equal(d1, ct1)
    ks.GadgetProduct(dcx, dkq, dkp, d0, d1, per_digit=True)             # the digit-by-digit form gives the same bits
    torch.cuda.synchronize()
    assert torch.equal(d0, ct0) and torch.equal(d1, ct1)
    res = ct0.cpu().numpy().view(np.uint64), ct1.cpu().numpy().view(np.uint64), list(ks.ownQ), list(ks.ownP), ks.beta
    ks.close()
    return res


# N >= 2^14: the pipelined digit-block transform (rows skip the digit's own limbs); (7, 3): digits of 3, 3, 1 limbs (single-prime branch)
@pytest.mark.parametrize("N,nq,np_", [(4096, 5, 2), (64, 6, 3), (8192, 7, 1 + 1), (16384, 7, 3), (16384, 6, 2)])
def test_single_rank_shard_path_equals_unsharded(rh, N, nq, np_):
    from matrix_fhe_lattigo_amd import sharding
    Q, P, beta, cx, evkQ, evkP = _case(N, nq, np_, 3, N + nq)
    e0, e1 = _unsharded(rh, N, Q, P, beta, cx, evkQ, evkP)
    g0, g1, ownQ, ownP, b = _run_shard(rh, sharding, N, Q, P, cx, evkQ, evkP, 0, 1, None)
    assert b == beta and ownQ == list(range(nq)) and ownP == list(range(np_))
    assert np.array_equal(g0, e0) and np.array_equal(g1, e1)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q, N, nq, np_):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import matrix_fhe_lattigo_amd as rh
    from matrix_fhe_lattigo_amd import sharding
    Q, P, beta, cx, evkQ, evkP = _case(N, nq, np_, 5, 99)              # same case on every rank; each keeps its limbs (5 polys: chunks of 2, 2, 1)
    g0, g1, ownQ, ownP, _ = _run_shard(rh, sharding, N, Q, P, cx, evkQ, evkP, rank, world, dist)
    ok = None
    if rank == 0:
        e0, e1 = _unsharded(rh, N, Q, P, beta, cx, evkQ, evkP)
        ok = (e0, e1)
    parts = sharding.gather_shards((g0, g1, ownQ, ownP), dist)          # the final gather
    if rank == 0:
        full0, full1 = np.zeros_like(ok[0]), np.zeros_like(ok[1])
        seen = []
        for p0, p1, oq, _op in parts:
            full0[:, oq] = p0; full1[:, oq] = p1; seen += oq
        q.put((sorted(seen) == list(range(nq)), bool(np.array_equal(full0, ok[0])), bool(np.array_equal(full1, ok[1])),
               [len(p[3]) for p in parts]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,N,nq,np_", [(2, 4096, 5, 2), (4, 64, 5, 2), (3, 8192, 6, 3), (2, 16384, 7, 3)])
def test_multi_rank_limb_shard_gloo(world, N, nq, np_):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q, N, nq, np_)) for r in range(world)]
    for p in ps:
        p.start()
    import queue as _queue
    import time as _time
    res, t0 = None, _time.time()
    while res is None and _time.time() - t0 < 300:                       # a rank that dies leaves the queue empty: do not wait the full time-out for it
        try:
            res = q.get(timeout=2)
        except _queue.Empty:
            if any(p.exitcode not in (None, 0) for p in ps):
                break
    if res is None:
        for p in ps:
            if p.is_alive():
                p.terminate()
        raise AssertionError("a rank failed: exit codes %s" % [p.exitcode for p in ps])
    covered, ok0, ok1, pcounts = res
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert covered and ok0 and ok1
    if world == 4:
        assert 0 in pcounts                                             # a rank that owns no P limb took part


def test_orchestrated_entry_error_behaviour(rh):
    # rh_kshard_gadget_product: a handle that owns a subset of the limbs needs the owner map; more than one rank needs the caller's all-gather;
    # a failing all-gather surfaces as a status with its code in the message (never an abort across the ABI); a wrong owner map is refused
    import ctypes as C
    import torch
    from matrix_fhe_lattigo_amd import sharding
    N, nq, np_ = 4096, 5, 2
    Q, P = QI60[:nq], PI60[:np_]
    L = rh.lib()
    ks = sharding.LimbShardedKeySwitch(N, Q, P, 0, 2, dist=None)               # rank 0 of 2: limbs 0, 2, 4 of Q and P1
    dev = torch.device("cuda", 0)
    nQ, nP = len(ks.ownQ), len(ks.ownP)
    cx = torch.zeros((2, nQ, N), dtype=torch.int64, device=dev)
    kq = torch.zeros((ks.beta, 2, nQ, N), dtype=torch.int64, device=dev)
    kp = torch.zeros((ks.beta, 2, nP, N), dtype=torch.int64, device=dev)
    c0, c1 = torch.empty_like(cx), torch.empty_like(cx)
    args = (ks._h, cx.data_ptr(), kq.data_ptr(), kp.data_ptr(), c0.data_ptr(), c1.data_ptr(), 2)
    assert L.rh_kshard_gadget_product(*args, rh.ringhip.ALLGATHER_FN(), None, 1) == -1 and b"all-gather" in L.rh_last_error()
    failing = rh.ringhip.ALLGATHER_FN(lambda ctx, s, r, w, st: 7)
    assert L.rh_kshard_gadget_product(*args, failing, None, 1) == -3 and b"returned 7" in L.rh_last_error()
    torch.cuda.synchronize()
    bad = (C.c_int * (nq + np_))(*([1] * (nq + np_)))                           # every limb owned by rank 1: not what rank 0 was created with
    assert L.rh_kshard_set_world(ks._h, 2, 0, bad) == -1 and b"owner map" in L.rh_last_error()
    w = C.c_size_t()
    assert L.rh_kshard_exchange_words(ks._h, 64, 4, C.byref(w)) == 0 and w.value > 0
    ks.close()
    # a fresh partial handle without set_world
    rq = rh.Ring(N, [Q[0], Q[2]])
    h = C.c_void_p()
    allQ, allP = rh.ringhip._u64(Q), rh.ringhip._u64(P)
    oq = (C.c_int * 2)(0, 2)
    rh.ringhip._check(L.rh_kshard_create(C.byref(h), rq._h, None, rh.ringhip._p(allQ), nq - 1, rh.ringhip._p(allP), np_ - 1, oq, 2, oq, 0))
    assert L.rh_kshard_gadget_product(h, cx.data_ptr(), kq.data_ptr(), None, c0.data_ptr(), c1.data_ptr(), 1, rh.ringhip.ALLGATHER_FN(), None, 1) == -1
    assert b"rh_kshard_set_world" in L.rh_last_error()
    L.rh_kshard_destroy(h); rq.close()
