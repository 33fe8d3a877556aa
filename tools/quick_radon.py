"""Radon-only loop for profiling (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mr_slam_amd import ring
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
g = torch.Generator(device="cuda:0").manual_seed(0)
img = (torch.rand((B, 120, 120), device="cuda:0", generator=g) * (torch.rand((B, 120, 120), device="cuda:0", generator=g) < 0.4)).contiguous()
plan = ring.ring_plan(0)
for _ in range(3): plan.forward(img, raw=False, normalized=True)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): plan.forward(img, raw=False, normalized=True)
b.record(); torch.cuda.synchronize()
print(f"radon B={B}: {a.elapsed_time(b)/10:.4f} ms")
