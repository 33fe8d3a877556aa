"""Launches the kernels whose hardware counters the bench JSON quotes, at the bench's shapes, a few times each, so that
`rocprofv3 --pmc <counters> -- python tools/pmc_targets.py` (one pass per counter group, no tracing) sees them.
tools/pmc_summary.py condenses the CSVs into profiles/<tag>_pmc.json."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from mr_slam_amd import bev, gicp, ring

dev = "cuda:0"
B, G = 1024, 16                     # the fused descriptor kernel is profiled at the bench's launch size: 16 x 1024 scans
shard = bench.make_shard(B, G, 0, dev)
xyz, offs = shard[0]
img = torch.empty((B, 1, 120, 120), dtype=torch.float32, device=dev)
plan = ring.ring_plan(0)
for _ in range(3):
    bev.cart_bev(xyz, offs, 1, 1, 120, 120, 1, out=img.view(B, -1))
    bev.polar_bev(xyz, offs, 1, 1, 40, 120, 20)
    _, norm = plan.forward(img.view(B, 120, 120), raw=False, normalized=True)
whole = bench.make_shard.whole
goffs = torch.arange(G * B + 1, dtype=torch.int64, device=dev) * bench.N_POINTS
gout = torch.empty((G * B, 120, 120), dtype=torch.float32, device=dev)
for _ in range(3):      # the single-launch form of the three kernels above (k_bev_radon3 / k_bev_radon2), 16 384 scans per launch
    ring.ring_descriptors_fused(whole.view(-1), goffs, raw=False, normalized=True, out_norm=gout)
del gout
spec = ring.half_spectrum(norm)
db = spec[torch.arange(10000, device=dev) % B].contiguous()
for nq in (1, 4):                   # the LDS-DMA kernel on row-layout entries (k_ring_sweep_dma<..., TILED = false>): 1 query, and 4 (round 6)
    for _ in range(3):
        ring.corr_sweep_fft(spec[:nq].contiguous(), db)
tiled = ring.spec_to_tiled(db)      # the database's resident format (mrs_loopdb): k_ring_sweep_dma<..., TILED = true>
for _ in range(3):
    ring.corr_sweep_fft_tiled(spec[:1].contiguous(), tiled)
for _ in range(3):                  # round 6: four queries per sweep on the same pipeline (12 waves, default cache policy: its own instantiation)
    ring.corr_sweep_fft_tiled_q(spec[:4].contiguous(), tiled)
del tiled
db6 = torch.stack([db[:2000].roll(k, 0) for k in range(6)], 1).contiguous()      # RING++: 2 000 entries x 6 channels, tiled: k_ring_sweep_dma<MC> + k_ring_mc_finish
tiled6 = ring.spec_to_tiled(db6)
for _ in range(3):
    ring.corr_sweep_fft_tiled(db6[:1].contiguous(), tiled6)
del tiled6, db6
idx = torch.randint(0, B, (B,), device=dev, dtype=torch.int32)
for _ in range(3):
    ring.spectrum_corr_pairs_db(norm, spec, idx)
# GICP kernels at the bench's timed shape (BASELINE configs[2]: 256 pairs x 120k points): covariances, a cold / moving / settled alignment
NP = int(os.environ.get("MRS_PMC_GICP_PAIRS", "256"))
srcs, tgts = bench._gicp_pairs(NP, 0)
g = gicp.GicpBatch(NP, 0)
g.set_params(k_correspondences=15, max_correspondence_distance=5.0, force_iterations=8)
g.set_sources(srcs); g.set_targets(tgts)
g.align()
del g, srcs, tgts
from mr_slam_amd import pointfeat
NS = int(os.environ.get("MRS_PMC_FEAT_SCANS", "64"))
pts = whole[0, :NS].permute(0, 2, 1).reshape(NS * bench.N_POINTS, 3).contiguous()
pointfeat.point_features(pts, np.arange(NS + 1, dtype=np.int64) * bench.N_POINTS, 30, want=("planes",))      # k_knn_cov<30> + k_feat_from_knn (RING++ front end)
torch.cuda.synchronize()
print("pmc targets done")
