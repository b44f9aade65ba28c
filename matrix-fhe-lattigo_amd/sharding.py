"""How the hot path shards over the GPUs of one node.

Limbs and polys are independent (ring/ntt.go:127-131), so NTT / vec-op workloads shard by batch with no data-path
collective; key-switching shards by limb and needs the digit's source limbs gathered.  One process per GPU."""


def poly_shard(total_polys, rank, world):
    """contiguous [lo, hi) slice of the batch owned by `rank`; sizes differ by at most one"""
    base, rem = divmod(total_polys, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def limb_shard(nlimbs, rank, world):
    """limbs {i : i mod world == rank}: round-robin so every rank holds a mix of Q and P limbs (config 5)"""
    return list(range(rank, nlimbs, world))


def max_over_ranks(value, dist=None, device=None):
    """bench contract: the job's step time is the MAX over ranks"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def gather_shards(local, dist):
    """final gather of per-rank results (the north star's only collective): list of per-rank objects on every rank"""
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, local)
    return out
