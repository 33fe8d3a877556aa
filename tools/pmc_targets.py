"""Launches the kernels whose hardware counters the bench JSON quotes, at the bench's shapes, a few times each, so that
`rocprofv3 --pmc <counters> -- python tools/pmc_targets.py` (one pass per counter group, no tracing) sees them.
tools/pmc_summary.py condenses the CSVs into profiles/r02_pmc.json."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from mr_slam_amd import bev, gicp, ring

dev = "cuda:0"
B = 1024
(xyz, offs), = bench.make_shard(B, 1, 0, dev)
img = torch.empty((B, 1, 120, 120), dtype=torch.float32, device=dev)
plan = ring.ring_plan(0)
for _ in range(3):
    bev.cart_bev(xyz, offs, 1, 1, 120, 120, 1, out=img.view(B, -1))
    bev.polar_bev(xyz, offs, 1, 1, 40, 120, 20)
    _, norm = plan.forward(img.view(B, 120, 120), raw=False, normalized=True)
for _ in range(3):      # the single-launch form of the three kernels above (k_bev_radon2)
    ring.ring_descriptors_fused(xyz, offs, raw=False, normalized=True)
spec = ring.half_spectrum(norm)
db = spec[torch.arange(10000, device=dev) % B].contiguous()
for nq in (1, 4):
    for _ in range(3):
        ring.corr_sweep_fft(spec[:nq].contiguous(), db)
idx = torch.randint(0, B, (B,), device=dev, dtype=torch.int32)
for _ in range(3):
    ring.spectrum_corr_pairs_db(norm, spec, idx)
srcs, tgts = bench._gicp_pairs(16, 0)
g = gicp.GicpBatch(16, 0)
g.set_params(k_correspondences=15, max_correspondence_distance=5.0, force_iterations=3)
g.set_sources(srcs); g.set_targets(tgts)
g.align()
torch.cuda.synchronize()
print("pmc targets done")
