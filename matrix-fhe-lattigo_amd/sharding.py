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


def gather_polys(block, dist=None, out=None, force=False):
    """The north star's one collective for the batch-sharded workloads (metric, configs 3 / 4): the final gather of the results.
    `block`: this rank's (B/G, L, N) device tensor (every rank the same shape: pad the last shard).  Returns the (G, B/G, L, N) tensor of
    every rank's block on every rank -- ONE all_gather_into_tensor on device memory (RCCL over xGMI with backend "nccl"); under "gloo"
    (CPU rehearsal) the block is staged through the host.  `out` lets the caller supply the destination (no allocation when timing);
    `force` runs the collective even in a world of one (pre-flight of the RCCL call on a single GPU)."""
    import torch
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        if out is None:
            return block.unsqueeze(0)
        out[0].copy_(block)
        return out
    world = dist.get_world_size()
    if out is None:
        out = torch.empty((world,) + tuple(block.shape), dtype=block.dtype, device=block.device)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out.view(-1), block.contiguous().view(-1))
        return out
    host = block.contiguous().cpu()
    parts = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(parts, host)
    for r in range(world):
        out[r].copy_(parts[r])
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
    evaluation key.  All arithmetic AND the orchestration are the HIP library's (rh_kshard_gadget_product, csrc/kshard.hip: INTT ->
    exchange -> product -> exchange -> ModDown, chunks of the batch pipelined on two streams); this class supplies the one thing a
    host must: the all-gather the library calls back for its two exchanges per chunk,

      * every limb of INTT(cx) (8*N bytes per limb and poly),
      * before ModDown, the k+1 limbs of both P-part accumulators.

    The callback runs torch.distributed on views of an exchange arena registered with the handle (backend "nccl" = RCCL over xGMI; with
    "gloo", the tests' backend, the same blocks are staged through the host).  Outputs stay limb-sharded.
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
        # the whole product behind the C ABI (rh_kshard_gadget_product): tell the handle who owns which limb of Q ++ P and hand it the
        # all-gather this host has -- torch.distributed here, ncclAllGather from a cgo host
        owner = [qp_owner(i, world) for i in range(nq + len(P))]
        rh._check(rh.lib().rh_kshard_set_world(h, world, rank, (C.c_int * len(owner))(*owner)))
        self._arena, self._arena_words = None, 0
        self._cb = rh.ALLGATHER_FN(self._allgather)                       # keep the thunk alive as long as the handle
        self.exchanges, self.exchange_words, self.cb_error = 0, 0, None   # counters of the last product (tests, bench)

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

    # ---- the all-gather the library calls back (rh_allgather_fn) ------------------------------------------------
    def _allgather(self, ctx, send, recv, words, stream):
        """every rank contributes `words` uint64 at `send` and receives world * words at `recv`, rank-major, enqueued on `stream`.
        Both pointers lie in the arena tensor registered with rh_kshard_set_exchange, so they map back to tensor views."""
        torch = self.torch
        try:
            base = self._arena.data_ptr()
            so, ro = (send - base) // 8, (recv - base) // 8
            if not (0 <= so and so + words <= self._arena_words and 0 <= ro and ro + self.world * words <= self._arena_words):
                raise RuntimeError("exchange block outside the registered arena")
            s, r = self._arena[so:so + words], self._arena[ro:ro + self.world * words]
            self.exchanges += 1
            self.exchange_words += (self.world - 1) * words                # words this rank RECEIVES from the others
            if self.gather_override is not None:                           # dry runs (per-rank arithmetic without a fabric)
                self.gather_override(s, r, words, stream)
                return 0
            # `stream` is NULL (None through ctypes) when the library runs a single chunk on the ring's own stream and that is torch's default
            ts = torch.cuda.ExternalStream(stream, device=self.device) if stream else torch.cuda.default_stream(self.device)
            with torch.cuda.stream(ts):
                if self.dist.get_backend() == "nccl":
                    self.dist.all_gather_into_tensor(r, s)                 # RCCL over xGMI, ordered after / before the kernels of `stream`
                else:                                                      # gloo (tests on one GPU): staged through the host
                    host = s.cpu()
                    parts = [torch.empty_like(host) for _ in range(self.world)]
                    self.dist.all_gather(parts, host)
                    r.copy_(torch.cat(parts))
            return 0
        except Exception as e:                                             # never let an exception cross the C frames
            self.cb_error = repr(e)
            return 1

    gather_override = None

    def _ensure_arena(self, npoly, chunks):
        import ctypes as C
        w = C.c_size_t()
        self.rh._check(self.rh.lib().rh_kshard_exchange_words(self._h, npoly, chunks, C.byref(w)))
        if w.value > self._arena_words:
            self.torch.cuda.synchronize(self.device)
            self._arena = self.torch.empty(w.value, dtype=self.torch.int64, device=self.device)
            self._arena_words = w.value
            self.rh._check(self.rh.lib().rh_kshard_set_exchange(self._h, self._arena.data_ptr(), w.value))

    # ---- the product ---------------------------------------------------------------------------------------
    def GadgetProduct(self, cx, evkQ, evkP, ct0, ct1, per_digit=False, orchestrate="c", chunks=0):
        """rlwe.Evaluator.GadgetProduct on the owned limbs.  orchestrate="c" (default): ONE call of rh_kshard_gadget_product -- the INTT ->
        exchange -> product -> exchange -> ModDown sequence, cut into `chunks` chunks on two streams (0: auto), lives behind the C ABI and
        only the all-gather is this host's.  orchestrate="python": the same sequence issued call by call from here (round 2's form; kept as
        a cross-check, and for per_digit=True)."""
        if orchestrate == "c" and not per_digit:
            npoly, npl = cx.shape[0], len(self.ownP)
            self.exchanges = self.exchange_words = 0
            if self.world > 1:
                self._ensure_arena(npoly, chunks)
            rc = self.rh.lib().rh_kshard_gadget_product(self._h, cx.data_ptr(), evkQ.data_ptr(), evkP.data_ptr() if npl else None,
                                                        ct0.data_ptr(), ct1.data_ptr(), npoly, self._cb if self.world > 1 else self.rh.ALLGATHER_FN(), None, chunks)
            if rc and self.cb_error:
                raise self.rh.RingHipError("all-gather callback: %s" % self.cb_error)
            self.rh._check(rc)
            return
        return self._gadget_product_python(cx, evkQ, evkP, ct0, ct1, per_digit)

    def _gadget_product_python(self, cx, evkQ, evkP, ct0, ct1, per_digit=False):
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
