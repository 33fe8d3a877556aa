"""The fused descriptor kernel (k_bev_radon2) at the bench's launch size -- 16 x 1024 scans per launch -- for the rocprofv3 --pmc
passes of tools/run_r03_first.sh (the r02 counters were taken at 1024 scans per launch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from mr_slam_amd import ring

dev = "cuda:0"
B, G = 1024, int(os.environ.get("PMC_FUSED_GROUP", "16"))
bench.make_shard(B, G, 0, dev)
whole = bench.make_shard.whole
offs = torch.arange(G * B + 1, dtype=torch.int64, device=dev) * bench.N_POINTS
out = torch.empty((G * B, 120, 120), dtype=torch.float32, device=dev)
if os.environ.get("PMC_FUSED_GRID"):       # fewer persistent workgroups than compute units (tools/run_pmc_raster.sh)
    plan = ring.ring_plan(0)
    plan.set_option(plan.OPT_FUSED_GRID, int(os.environ["PMC_FUSED_GRID"]))
    plan.set_option(plan.OPT_FUSED_STAGGER_US, 0)
for _ in range(3):
    ring.ring_descriptors_fused(whole.view(-1), offs, raw=False, normalized=True, out_norm=out)
torch.cuda.synchronize()
print("pmc fused done", float(out[0, 0, 0]))
