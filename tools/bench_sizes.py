#!/usr/bin/env python3
"""Ring.NTT / INTT rate by ring degree (N = 2^12 .. 2^17, 16 limbs, 4 GiB batches): does every size the reference's parameter sets use
(ring/ring.go:318: N up to 2^17) sit near the headline's fraction of HBM?  JSON on stdout.
Arguments: key=value tuning pairs applied to every ring (e.g. one_pass=0: the two-pass launches at N = 2^13 / 2^14 too), logn=13,14 to pick sizes."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import matrix_fhe_lattigo_amd as rh
from bench import QI60

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
res = []
tunes = [a.split("=") for a in sys.argv[1:] if "=" in a and not a.startswith("logn=")]
sizes = [int(v) for a in sys.argv[1:] if a.startswith("logn=") for v in a[5:].split(",")] or [12, 13, 14, 15, 16, 17]
for logn in sizes:
    N, L = 1 << logn, 16
    B = (4 << 30) // (N * L * 8)
    ring = rh.Ring(N, QI60[:L]); ring.set_stream(stream.cuda_stream)
    for k, v in tunes:
        ring.set_tuning(k, int(v))
    x = torch.randint(0, 1 << 60, (B, L, N), dtype=torch.int64, device=dev)
    p = rh.DevicePoly.from_torch(ring, x)
    row = {"logN": logn, "batch": B}
    for name, fn in (("NTT", lambda: ring.NTT(p, p)), ("INTT", lambda: ring.INTT(p, p))):
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(10):
            fn()
        e1.record(stream); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        row[name] = {"ms": ms, "frac_of_8TBps": 16.0 * N * L * B / (ms * 1e-3) / 8e12, "limb_ntt_per_s": B * L / (ms * 1e-3)}
    res.append(row)
    del p, x
    ring.close(); torch.cuda.empty_cache()
    sys.stderr.write(json.dumps(row) + "\n")
print(json.dumps(res, indent=1))
