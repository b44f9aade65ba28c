"""Config 3 experiment (round 3): Ring.PolyMul (the tile stages of all three transforms as one kernel) against Ring.NTTMany + Ring.INTTMul; see DESIGN.md 6."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import matrix_fhe_lattigo_amd as rh
from conftest import QI60
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream()
N, L, B = 1 << 15, 16, 512
ring = rh.Ring(N, QI60[:L]); ring.set_stream(stream.cuda_stream)
qs = torch.tensor(QI60[:L], dtype=torch.int64, device=dev).view(1, L, 1)
a = torch.randint(0, 1 << 62, (B, L, N), dtype=torch.int64, device=dev) % qs
b = torch.randint(0, 1 << 62, (B, L, N), dtype=torch.int64, device=dev) % qs
pa, pb = rh.DevicePoly.from_torch(ring, a), rh.DevicePoly.from_torch(ring, b)
for _ in range(6):
    ring.PolyMul(pa, pb, pa)
torch.cuda.synchronize()
