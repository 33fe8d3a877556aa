"""Development aid: streaming rate of the fused descriptor kernel's rasteriser alone (MRS_DEV=1 MRS_FUSED_SKIP=2: no ray march) against the number of
persistent workgroups, the prefetch depth and the size of the scan set (a set below 256 MB is served by the MALL on repeated launches)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mr_slam_amd import ring
dev = "cuda:0"; N = bench.N_POINTS
bench.make_shard(1024, 4, 0, dev); whole = bench.make_shard.whole.view(-1)
plan = ring.ring_plan(0)
for nscan in (4096, 128, 64):
    offs = torch.arange(nscan + 1, dtype=torch.int64, device=dev) * N
    flat = whole[: nscan * 3 * N]
    buf = [None]
    def fused():
        buf[0] = ring.ring_descriptors_fused(flat, offs, raw=False, normalized=True, out_norm=buf[0])[2]
    for pf in (2, 4):
        plan.set_option(plan.OPT_FUSED_PREFETCH, pf)
        for grid in (16, 32, 256):
            if grid * 2 > nscan:
                continue
            plan.set_option(plan.OPT_FUSED_GRID, grid); plan.set_option(plan.OPT_FUSED_STAGGER_US, 0)
            ms = bench.ev_ms(fused, reps=5, warm=2)
            gb = nscan * 12 * N / 1e9
            print(f"skip={os.environ.get('MRS_FUSED_SKIP')} scans={nscan} ({gb*1e3:.0f} MB) pf={pf} grid={grid}: {ms:.3f} ms, {gb/ms*1e3:.0f} GB/s total, "
                  f"{gb/ms*1e3/grid:.1f} GB/s per workgroup", flush=True)
