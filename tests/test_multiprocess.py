"""CPU, 2 processes over gloo: the N>1 path of the bench/host logic -- batch sharding without a data-path collective,
max-over-ranks timing, final gather.  The per-rank compute stands in with the oracle (no GPU here)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import matrix_fhe_lattigo_amd  # noqa: F401  (alias module)
    from matrix_fhe_lattigo_amd import sharding
    import oracle
    N, mods, B = 256, [0x1fffffffffe00001, 0x1fffffffffc80001], 5
    rng = np.random.default_rng(7)                               # same batch on every rank, each transforms its shard
    a = np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(m) for m in mods]) for _ in range(B)])
    lo, hi = sharding.poly_shard(B, rank, world)
    srs = [oracle.SubRingConsts(N, m) for m in mods]
    local = np.stack([np.stack([oracle.ntt(a[k, i], srs[i]) for i in range(2)]) for k in range(lo, hi)]) if hi > lo else np.zeros((0, 2, N), dtype=np.uint64)
    t = sharding.max_over_ranks(1.0 + rank, dist)
    parts = sharding.gather_shards(local, dist)
    full = np.concatenate(parts)
    exp = np.stack([np.stack([oracle.ntt(a[k, i], srs[i]) for i in range(2)]) for k in range(B)])
    q.put((rank, t, bool(np.array_equal(full, exp)), (lo, hi)))
    dist.destroy_process_group()


def test_two_rank_batch_shard_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(abs(r[1] - 2.0) < 1e-9 for r in res)          # MAX over ranks of (1.0, 2.0)
    assert all(r[2] for r in res)                             # gathered shards == whole-batch transform
    assert res[0][3] == (0, 3) and res[1][3] == (3, 5)


def test_shard_helpers():
    sys.path.insert(0, ROOT)
    import matrix_fhe_lattigo_amd  # noqa: F401
    from matrix_fhe_lattigo_amd import sharding
    for total in (0, 1, 7, 1024):
        for world in (1, 2, 3, 8):
            spans = [sharding.poly_shard(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert sorted(sum((sharding.limb_shard(30, r, 8) for r in range(8)), [])) == list(range(30))
    # limb-sharded key switch (config 5): Q ++ P dealt round-robin; every limb has exactly one owner, ranks may own no P limb
    nq, np_ = 24, 6
    for world in (1, 2, 3, 8):
        owners = [sharding.qp_owner(i, world) for i in range(nq + np_)]
        assert all(0 <= o < world for o in owners)
        per = [owners.count(r) for r in range(world)]
        assert sum(per) == nq + np_ and max(per) - min(per) <= 1
    assert [sharding.qp_owner(nq + j, 8) for j in range(np_)] == [0, 1, 2, 3, 4, 5]        # ranks 6, 7 own no P limb


def _gather_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import matrix_fhe_lattigo_amd  # noqa: F401
    from matrix_fhe_lattigo_amd import sharding
    nq, np_, npoly, N = 7, 3, 2, 16
    full_q = torch.arange(npoly * nq * N, dtype=torch.int64).reshape(npoly, nq, N)            # limb i of poly k is recognisable
    full_p = -torch.arange(npoly * np_ * N, dtype=torch.int64).reshape(npoly, np_, N) - 1
    ownq = [i for i in range(nq) if sharding.qp_owner(i, world) == rank]
    ownp = [j for j in range(np_) if sharding.qp_owner(nq + j, world) == rank]
    ok = True
    for st, ed in ((0, 3), (3, 6), (6, 7), (0, 7)):                                           # digits of alpha = 3, and the whole chain
        got = sharding.gather_limbs(full_q[:, ownq], ownq, 0, st, ed, world, dist)
        ok = ok and bool(torch.equal(got, full_q[:, st:ed]))
    lp = full_p[:, ownp] if ownp else torch.zeros((npoly, 0, N), dtype=torch.int64)           # a rank may own no P limb
    got = sharding.gather_limbs(lp, ownp, nq, 0, np_, world, dist)
    ok = ok and bool(torch.equal(got, full_p))
    q.put((rank, ok, len(ownp)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_limb_exchange_gloo(world):
    # the key switch's only data-path collective, on host tensors: padding, ownership, reorder
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    if world == 4:
        assert any(r[2] == 0 for r in res)


def _gather_polys_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import matrix_fhe_lattigo_amd  # noqa: F401
    from matrix_fhe_lattigo_amd import sharding
    sys.path.insert(0, ROOT)
    import bench
    # the final gather of a batch-sharded result (SURVEY 8e / north star): every rank contributes its (B/G, L, N) block
    B, L, N = 3, 4, 32
    block = (torch.arange(B * L * N, dtype=torch.int64).reshape(B, L, N) + 1000000 * rank)
    out = sharding.gather_polys(block, dist)
    ok = tuple(out.shape) == (world, B, L, N) and all(bool(torch.equal(out[r], block - 1000000 * rank + 1000000 * r)) for r in range(world))
    pre = torch.empty((world, B, L, N), dtype=torch.int64)
    ok = ok and sharding.gather_polys(block, dist, out=pre) is pre and bool(torch.equal(pre, out))
    # bench.py's rank plumbing on the same job: the AND over ranks of a per-rank flag, shards with remainders, the batch-shard key-switch split

    class A:
        dist_backend = "gloo"
    ok = ok and bench.all_ranks_ok(True, A, dist, "cpu") is True and bench.all_ranks_ok(rank != world - 1, A, dist, "cpu") is False
    spans = [sharding.poly_shard(64, r, world) for r in range(world)]
    ok = ok and spans[rank][1] - spans[rank][0] == 64 // world + (1 if rank < 64 % world else 0)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_final_gather_of_result_polys_gloo(world):
    # row e' of the verdict: one all-gather of every rank's result block; 8 ranks = the rendezvous of the driver's 8-GPU run
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_gather_polys_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world)) and all(r[1] for r in res)


def test_gather_polys_single_process():
    sys.path.insert(0, ROOT)
    import torch
    import matrix_fhe_lattigo_amd  # noqa: F401
    from matrix_fhe_lattigo_amd import sharding
    b = torch.arange(24, dtype=torch.int64).reshape(2, 3, 4)
    out = sharding.gather_polys(b, None)
    assert tuple(out.shape) == (1, 2, 3, 4) and torch.equal(out[0], b)
