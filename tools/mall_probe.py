"""Probe: does the 256 MiB Infinity Cache retain just-WRITTEN data (so a second pass can read it back on-die)?
Measures read bandwidth of a buffer right after (a) writing it, (b) reading it, for several sizes."""
import torch, time
dev = torch.device("cuda")
def bw(fn, nbytes, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        pre(); 
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return nbytes / (ts[len(ts)//2] * 1e-3) / 1e12
for mb in (16, 32, 64, 128, 192, 256, 384, 512, 1024, 4096):
    n = mb * (1 << 20) // 8
    a = torch.empty(n, dtype=torch.int64, device=dev)
    b = torch.empty(n, dtype=torch.int64, device=dev)
    big = torch.empty(1 << 27, dtype=torch.int64, device=dev)   # 1 GiB flush buffer
    out = {}
    # read after write
    pre = lambda: a.fill_(3)
    out["read_after_write"] = bw(lambda: a.sum(), n * 8)
    pre = lambda: (a.sum())
    out["read_after_read"] = bw(lambda: a.sum(), n * 8)
    pre = lambda: big.fill_(1)
    out["read_cold"] = bw(lambda: a.sum(), n * 8)
    pre = lambda: big.fill_(1)
    out["copy_cold"] = bw(lambda: b.copy_(a), 2 * n * 8)
    pre = lambda: a.fill_(3)
    out["copy_after_write"] = bw(lambda: b.copy_(a), 2 * n * 8)
    print(mb, "MiB:", {k: round(v, 2) for k, v in out.items()}, "TB/s", flush=True)
