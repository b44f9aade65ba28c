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


# ---------------------------------------------------------------------------------------------------------------
# Limb-sharded hybrid key switch (BASELINE config 5)
# ---------------------------------------------------------------------------------------------------------------
def qp_owner(i, world):
    """owner of limb i of the concatenation Q ++ P (P limb j has index len(Q) + j)"""
    return i % world


def all_gather_blocks(block, world, dist, device=None):
    """all-gather of equally shaped blocks -> list indexed by rank.  Device tensors go through RCCL (backend "nccl");
    with "gloo" (tests) device blocks are staged through the host, host blocks are gathered as they are."""
    import torch
    if dist is None or world == 1:
        return [block]
    if dist.get_backend() == "nccl":
        out = [torch.empty_like(block) for _ in range(world)]
        dist.all_gather(out, block)
        return out
    host = block.cpu()
    outs = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(outs, host)
    return [o.to(block.device) for o in outs]


def gather_limbs(local, own, base, st, ed, world, dist):
    """The key switch's exchange step.  `local`: (npoly, len(own), N) block of the limbs this rank owns of one chain
    (Q or P), `own` their indices in that chain, `base` the chain's offset in Q ++ P (0 for Q, len(Q) for P).
    Returns limbs [st, ed) of the chain, (npoly, ed-st, N) in chain order, on every rank: each rank contributes its
    owned limbs of the range padded to the largest per-rank count, ONE all-gather, then a local reorder."""
    import torch
    npoly, N = local.shape[0], local.shape[2]
    mine = [k for k, g in enumerate(own) if st <= g < ed]
    counts = [sum(1 for g in range(st, ed) if qp_owner(base + g, world) == r) for r in range(world)]
    cmax = max(counts)
    block = torch.zeros((npoly, cmax, N), dtype=local.dtype, device=local.device)
    if mine:
        block[:, :len(mine)] = local[:, mine]
    parts = all_gather_blocks(block, world, dist)
    if world == 1:
        return parts[0] if cmax == ed - st else parts[0][:, :ed - st].contiguous()
    src = torch.empty((npoly, ed - st, N), dtype=local.dtype, device=local.device)
    for r in range(world):                            # one indexed copy per rank: its limbs of the range, in chain order
        idx = [g - st for g in range(st, ed) if qp_owner(base + g, world) == r]
        if idx:
            src[:, idx] = parts[r][:, :len(idx)]
    return src


class LimbShardedKeySwitch:
    """rlwe.Evaluator.GadgetProduct (core/rlwe/evaluator_gadget_product.go:16-30) with the limbs of Q ++ P dealt
    round-robin over the ranks of one node: rank r owns limbs {i : i mod world == r} and the matching slice of the
    evaluation key.  All arithmetic is the HIP library's (rh_kshard_*, csrc/kshard.hip); this class only moves limbs:

      * one all-gather of every limb of INTT(cx) (8*N bytes per limb and poly; digit by digit with per_digit=True),
      * before ModDown, one all-gather of the k+1 limbs of both P-part accumulators.

    Collectives run through torch.distributed on device tensors (backend "nccl" = RCCL over xGMI); with the "gloo"
    backend (tests) the same blocks are staged through the host.  Outputs stay limb-sharded.
    Tensors are int64 views of the uint64 words, shape (npoly, owned limbs, N), on `device`."""

    def __init__(self, N, Q, P, rank, world, dist=None, device=0):
        import ctypes as C
        import numpy as np
        import torch
        from . import ringhip as rh
        self.rh, self.torch, self.dist = rh, torch, dist
        self.N, self.Q, self.P, self.rank, self.world = int(N), [int(q) for q in Q], [int(p) for p in P], rank, world
        self.levelQ, self.levelP = len(Q) - 1, len(P) - 1
        self.device = torch.device("cuda", device)
        nq = len(Q)
        self.ownQ = [i for i in range(nq) if qp_owner(i, world) == rank]
        self.ownP = [j for j in range(len(P)) if qp_owner(nq + j, world) == rank]
        if not self.ownQ:
            raise rh.RingHipError("rank %d owns no Q limb (%d limbs over %d ranks)" % (rank, nq, world))
        self.ringQ = rh.Ring(N, [self.Q[i] for i in self.ownQ], device=device)
        self.ringP = rh.Ring(N, [self.P[j] for j in self.ownP], device=device) if self.ownP else None
        h = C.c_void_p()
        allQ, allP = rh._u64(self.Q), rh._u64(self.P)
        oq = (C.c_int * len(self.ownQ))(*self.ownQ)
        op = (C.c_int * max(len(self.ownP), 1))(*self.ownP)
        rh._check(rh.lib().rh_kshard_create(C.byref(h), self.ringQ._h, self.ringP._h if self.ringP else None, rh._p(allQ), self.levelQ,
                                            rh._p(allP), self.levelP, oq, len(self.ownQ), op, len(self.ownP)))
        self._h = h
        self.beta = rh.lib().rh_kshard_num_digits(h)
        stream = torch.cuda.current_stream(self.device).cuda_stream       # kernels and collectives in one stream order
        self.ringQ.set_stream(stream)
        if self.ringP:
            self.ringP.set_stream(stream)

    # ---- helpers -------------------------------------------------------------------------------------------
    def digit_range(self, d):
        import ctypes as C
        st, ed = C.c_int(), C.c_int()
        self.rh._check(self.rh.lib().rh_kshard_digit_range(self._h, d, C.byref(st), C.byref(ed)))
        return st.value, ed.value

    def shard_q(self, full):
        """(npoly, len(Q), N) array -> this rank's (npoly, owned, N) rows"""
        return full[:, self.ownQ]

    def shard_key(self, evkQ_full, evkP_full):
        """(beta, 2, len(Q), N) / (beta, 2, len(P), N) -> the owned limb slices, same leading layout"""
        return evkQ_full[:, :, self.ownQ], (evkP_full[:, :, self.ownP] if self.ownP else None)

    def to_device(self, arr):
        import numpy as np
        return self.torch.from_numpy(np.ascontiguousarray(arr).view(np.int64)).to(self.device)

    def _dp(self, ring, t):
        return self.rh.DevicePoly.from_torch(ring, t)

    def _gather_limbs(self, local, own, base, st, ed):
        return gather_limbs(local, own, base, st, ed, self.world, self.dist)

    # ---- the product ---------------------------------------------------------------------------------------
    def GadgetProduct(self, cx, evkQ, evkP, ct0, ct1, per_digit=False):
        """cx: owned limbs of the NTT-domain input, (npoly, owned Q, N); evkQ / evkP: owned key slices
        (beta, 2, owned, N) on the device; ct0 / ct1: outputs, owned Q limbs (NTT domain, canonical).
        Default: TWO exchanges per product -- every limb of INTT(cx) in one all-gather (rh_kshard_product: all digits in one call,
        the single-GPU product's structure), then the P parts of both accumulators in one all-gather before ModDown.
        per_digit=True: the digit-by-digit form (one gather per digit, rh_kshard_digit), kept for comparison."""
        torch, rh, L = self.torch, self.rh, self.rh.lib()
        npoly, nq, npl = cx.shape[0], len(self.ownQ), len(self.ownP)
        cxinv = torch.empty_like(cx)
        self.ringQ.INTT(self._dp(self.ringQ, cx), self._dp(self.ringQ, cxinv))                   # ringQ.INTT (:138)
        acc0 = torch.empty((npoly, max(npl, 1), self.N), dtype=torch.int64, device=self.device)
        acc1 = torch.empty_like(acc0)
        pP = lambda t: t.data_ptr() if npl else None
        if per_digit:
            for d in range(self.beta):
                st, ed = self.digit_range(d)
                src = self._gather_limbs(cxinv, self.ownQ, 0, st, ed)
                rh._check(L.rh_kshard_digit(self._h, d, src.data_ptr(), cx.data_ptr(), evkQ.data_ptr(), pP(evkP) if npl else None,
                                            ct0.data_ptr(), ct1.data_ptr(), pP(acc0), pP(acc1), npoly))
        else:
            src = cxinv if self.world == 1 else self._gather_limbs(cxinv, self.ownQ, 0, 0, self.levelQ + 1)
            rh._check(L.rh_kshard_product(self._h, src.data_ptr(), cx.data_ptr(), evkQ.data_ptr(), pP(evkP) if npl else None,
                                          ct0.data_ptr(), ct1.data_ptr(), pP(acc0), pP(acc1), npoly))
        # eval.ModDown (:33-46): the P parts of both accumulators travel together
        both = torch.cat((acc0, acc1), dim=0)                                                   # (2 npoly, owned P, N)
        if npl:
            self.ringP.INTTLazy(self._dp(self.ringP, both), self._dp(self.ringP, both))
        srcP = self._gather_limbs(both, self.ownP, len(self.Q), 0, self.levelP + 1)
        for c, ct in enumerate((ct0, ct1)):
            rh._check(L.rh_kshard_moddown(self._h, srcP[c * npoly:(c + 1) * npoly].data_ptr(), ct.data_ptr(), ct.data_ptr(), npoly))

    def close(self):
        if getattr(self, "_h", None):
            self.rh.lib().rh_kshard_destroy(self._h)
            self._h = None
            self.ringQ.close()
            if self.ringP:
                self.ringP.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
