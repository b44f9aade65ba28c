#!/usr/bin/env python3
"""Single-ciphertext latency (the way the reference's callers issue work: one ct per call) at the config 5 ring, N = 2^16, Q = 24, P = 6 limbs:
the hybrid key switch (rlwe.Evaluator.GadgetProduct) and a few ring calls behind it, issued (a) call by call and (b) as ONE HIP graph
replay captured by the caller on the stream handed to rh_ring_set_stream.  JSON on stdout -> profiles/r03_latency_batch1.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import matrix_fhe_lattigo_amd as rh
from bench import QI60, PI60

dev = torch.device("cuda", 0)
N, Q, P = 1 << 16, QI60[:24], PI60[:6]
res = {"what": "wall-clock per call, one ciphertext component per call, N=2^16 Q=24 P=6 (beta=4); eager = call by call, graph = one hipGraph replay", "rows": []}


def uniform(shape, mods):
    t = torch.randint(0, 1 << 62, shape, dtype=torch.int64, device=dev)
    return t % torch.tensor(mods, dtype=torch.int64, device=dev).view(*([1] * (len(shape) - 2)), len(mods), 1)


tunes = [a.split("=") for a in sys.argv[1:] if "=" in a]           # e.g. ks_small_rows=0: the pipelined digit stream at every batch size
res["tuning"] = dict(tunes)
for B in (1, 2, 4, 8):
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    for k, v in tunes:
        rq.set_tuning(k, int(v)); rp.set_tuning(k, int(v))
    be = rh.BasisExtender(rq, rp)
    beta = (len(Q) - 1 + len(P)) // len(P)
    cx = uniform((B, len(Q), N), Q)
    kq, kp = uniform((beta * 2, len(Q), N), Q), uniform((beta * 2, len(P), N), P)
    c0, c1 = torch.zeros_like(cx), torch.zeros_like(cx)
    pcx, p0, p1 = (rh.DevicePoly.from_torch(rq, t) for t in (cx, c0, c1))
    side = torch.cuda.Stream()

    def work():
        be.GadgetProduct(len(Q) - 1, len(P) - 1, pcx, kq.data_ptr(), kp.data_ptr(), beta, p0, p1)

    with torch.cuda.stream(side):
        rq.set_stream(side.cuda_stream); rp.set_stream(side.cuda_stream)
        for _ in range(3):
            work()
        side.synchronize()
        ref = (c0.clone(), c1.clone())
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            work()
        side.synchronize()
        eager = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(50):
            work(); side.synchronize()
        eager_sync = (time.perf_counter() - t0) / 50
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            work()
        c0.zero_(); c1.zero_()
        g.replay(); side.synchronize()
        same = torch.equal(c0, ref[0]) and torch.equal(c1, ref[1])
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        side.synchronize()
        graph = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay(); side.synchronize()
        graph_sync = (time.perf_counter() - t0) / 50
    # the operations a reference caller issues on ONE ciphertext: ct x ct multiply + relinearise + rescale (schemes/ckks/evaluator.go:786-881, 500-535)
    # and a rotation (core/rlwe/evaluator_automorphism.go:25-60), call by call with a synchronisation each (the latency a caller sees)
    extra = {}
    if B == 1:
        gct = rh.rlwe.GadgetCiphertext.__new__(rh.rlwe.GadgetCiphertext)
        gct.digits, gct.levelQ, gct.levelP = beta, len(Q) - 1, len(P) - 1
        gct.Q, gct.P = rh.DevicePoly.from_torch(rq, kq), rh.DevicePoly.from_torch(rp, kp)
        cev = rh.ckks.Evaluator(rq, rp, rlk=gct)
        kev = rh.rlwe.Evaluator(rq, rp, galois_keys={5: gct})
        mk = lambda: rh.DevicePoly.from_torch(rq, uniform((1, len(Q), N), Q))
        ctA, ctB, ctO, ctR = (rh.Ciphertext([mk(), mk()], is_ntt=True) for _ in range(4))
        with torch.cuda.stream(side):
            def timed_sync(fn, reps=50):
                fn(); fn(); side.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn(); side.synchronize()
                return (time.perf_counter() - t0) / reps * 1e6
            extra["ckks_MulRelin_us_with_sync"] = timed_sync(lambda: cev.MulRelin(ctA, ctB, ctO, relin=True))
            extra["ckks_Rescale_us_with_sync"] = timed_sync(lambda: cev.Rescale(ctO, ctR))
            extra["rlwe_Automorphism_us_with_sync"] = timed_sync(lambda: kev.Automorphism(ctA, 5, ctO))
        kev.close(); cev.close()
    row = {"op": "GadgetProduct", "polys": B, "eager_us_back_to_back": eager * 1e6, "eager_us_with_sync": eager_sync * 1e6,
           "graph_us_back_to_back": graph * 1e6, "graph_us_with_sync": graph_sync * 1e6, "graph_equals_eager": bool(same)}
    row.update(extra)
    res["rows"].append(row)
    sys.stderr.write(json.dumps(row) + "\n")
    del g
    be.close(); rq.close(); rp.close()
print(json.dumps(res, indent=1))
