#!/usr/bin/env python3
"""Inputs of the config 5 scaling table: what ONE rank computes per key switch under the two ways of sharding
rlwe.Evaluator.GadgetProduct (N = 2^16, Q = Qi60[0:24], P = Pi60[0:6], beta = 4) over G GPUs, measured on one MI355X.

  limb-shard  rank 0 of G in {1, 2, 4, 8} runs rh_kshard_gadget_product on its limbs of the WHOLE batch with the exchange replaced by a no-op
              (the arithmetic and the pack / unpack copies are real, the fabric is absent): per-rank milliseconds per product + the bytes
              that rank would receive over xGMI per product.
  batch-shard a rank runs the unsharded rh_bext_gadget_product on B / G polys with the whole 120 MiB key.

JSON on stdout -> profiles/r03_kshard_pricing.json; DESIGN.md adds the xGMI time at a stated all-gather rate."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import matrix_fhe_lattigo_amd as rh                         # noqa: E402
from matrix_fhe_lattigo_amd import sharding                 # noqa: E402
from bench import QI60, PI60                                # noqa: E402

N, Q, P = 1 << 16, QI60[:24], PI60[:6]
dev = torch.device("cuda", 0)


def uniform(shape, mods):
    t = torch.randint(0, 1 << 62, shape, dtype=torch.int64, device=dev)
    return t % torch.tensor(mods, dtype=torch.int64, device=dev).view(*([1] * (len(shape) - 2)), len(mods), 1)


def time_ms(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps


def limb_shard(world, B, chunks):
    ks = sharding.LimbShardedKeySwitch(N, Q, P, 0, world, dist=None)
    ks.gather_override = lambda s, r, words, stream: None       # the fabric is absent: what remains is this rank's own work
    cx = uniform((B, len(ks.ownQ), N), [Q[i] for i in ks.ownQ])
    kq = uniform((ks.beta, 2, len(ks.ownQ), N), [Q[i] for i in ks.ownQ])
    kp = uniform((ks.beta, 2, len(ks.ownP), N), [P[j] for j in ks.ownP]) if ks.ownP else None
    c0, c1 = torch.empty_like(cx), torch.empty_like(cx)
    ms = time_ms(lambda: ks.GadgetProduct(cx, kq, kp, c0, c1, chunks=chunks), 10 if B >= 32 else 30)
    out = {"ms_per_product": ms, "exchanges": ks.exchanges, "bytes_received_per_gpu": ks.exchange_words * 8, "owned_q": len(ks.ownQ), "owned_p": len(ks.ownP)}
    ks.close()
    return out


def batch_shard(nb):
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    st = torch.cuda.current_stream().cuda_stream
    rq.set_stream(st); rp.set_stream(st)
    be = rh.BasisExtender(rq, rp)
    beta = (len(Q) - 1 + len(P)) // len(P)
    cx = uniform((nb, len(Q), N), Q)
    kq, kp = uniform((beta * 2, len(Q), N), Q), uniform((beta * 2, len(P), N), P)
    c0, c1 = torch.empty_like(cx), torch.empty_like(cx)
    dp = lambda r, t: rh.DevicePoly.from_torch(r, t)
    pcx, p0, p1 = dp(rq, cx), dp(rq, c0), dp(rq, c1)
    ms = time_ms(lambda: be.GadgetProduct(len(Q) - 1, len(P) - 1, pcx, kq.data_ptr(), kp.data_ptr(), beta, p0, p1), 10 if nb >= 32 else 30)
    be.close(); rq.close(); rp.close()
    return ms


def main():
    res = {"what": "per-rank milliseconds per key switch of a batch (N=2^16, Q=24, P=6, beta=4), one MI355X; exchange replaced by a no-op for limb-shard",
           "limb_shard": {}, "batch_shard_ms_by_polys_per_rank": {}}
    for nb in (1, 2, 4, 8, 16, 32, 64):
        res["batch_shard_ms_by_polys_per_rank"][str(nb)] = batch_shard(nb)
        sys.stderr.write("batch-shard %d polys: %.3f ms\n" % (nb, res["batch_shard_ms_by_polys_per_rank"][str(nb)]))
    for world in (1, 2, 4, 8):
        for B in (1, 8, 64):
            for chunks in ((1,) if (world == 1 or B < 4) else (1, 4)):
                r = limb_shard(world, B, chunks)
                res["limb_shard"]["G%d_B%d_chunks%d" % (world, B, chunks)] = r
                sys.stderr.write("limb-shard G=%d B=%d chunks=%d: %.3f ms, %.1f MiB received\n" % (world, B, chunks, r["ms_per_product"], r["bytes_received_per_gpu"] / 2 ** 20))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
