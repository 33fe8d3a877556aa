#!/usr/bin/env python3
"""A/B of the single-launch descriptor kernel (mrs_ring_descriptors_batch) against the two-kernel sequence on bench-shaped scans:
same bits, time per 1024 scans for several launch sizes and tuning knobs.  Writes gpurun_out/ab_fused.json.
  python tools/ab_fused.py [--chunks 8] [--batch 1024]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mr_slam_amd import bev, ring  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=24)
ap.add_argument("--batch", type=int, default=1024)
args = ap.parse_args()
dev = "cuda:0"
B, CH, N = args.batch, args.chunks, bench.N_POINTS
t0 = time.perf_counter()
chunks = bench.make_shard(B, CH, 0, dev)
whole = bench.make_shard.whole
plan = ring.ring_plan(0)
img = torch.empty((B, 1, 120, 120), dtype=torch.float32, device=dev)
res = {"setup_s": time.perf_counter() - t0, "batch": B, "chunks": CH, "rows": []}


def separate(nch):
    out = []
    for c in range(nch):
        bev.cart_bev(chunks[c][0], chunks[c][1], 1, 1, 120, 120, 1, out=img.view(B, -1))
        out.append(plan.forward(img.view(B, 120, 120), raw=False, normalized=True)[1])
    return out


def fused(nch, out):
    offs = torch.arange(nch * B + 1, dtype=torch.int64, device=dev) * N
    return ring.ring_descriptors_fused(whole[:nch].view(-1), offs, raw=False, normalized=True, out_norm=out)[2]


# correctness first: the whole shard, default knobs
ref = torch.cat(separate(CH))
buf = torch.empty_like(ref)
got = fused(CH, buf)
torch.cuda.synchronize()
res["bit_identical"] = bool(torch.equal(ref.view(torch.int32), got.view(torch.int32)))
res["mismatching_sinograms"] = int((ref.view(CH * B, -1) != got.view(CH * B, -1)).any(1).sum())
print("bit identical:", res["bit_identical"], flush=True)

ms_sep = bench.ev_ms(lambda: separate(1), reps=5, warm=2)
res["separate_ms_per_1024"] = ms_sep * 1024 / B
print(f"separate kernels: {ms_sep:.4f} ms per launch of {B}", flush=True)
for variant in (0, 1):
    plan.set_option(plan.OPT_FUSED_VARIANT, variant)
    plan.set_option(plan.OPT_FUSED_STAGGER_US, 70); plan.set_option(plan.OPT_FUSED_PREFETCH, 2)
    got = fused(CH, buf)
    torch.cuda.synchronize()
    same = bool(torch.equal(ref.view(torch.int32), got.view(torch.int32)))
    res[f"bit_identical_variant{variant}"] = same
    print(f"variant {variant} bit identical: {same}", flush=True)
    for nch in sorted({1, min(16, CH), CH}):
        for stagger in (0, 70) if nch > 1 else (0,):
            for pf in (2, 4, 6) if variant == 1 else (2, 4):
                plan.set_option(plan.OPT_FUSED_STAGGER_US, stagger)
                plan.set_option(plan.OPT_FUSED_PREFETCH, pf)
                ms = bench.ev_ms(lambda: fused(nch, buf[:nch * B]), reps=4, warm=1)
                row = {"variant": variant, "launch_scans": nch * B, "stagger_us": stagger, "prefetch": pf, "ms_per_1024": ms / nch * 1024 / B}
                res["rows"].append(row)
                print(row, flush=True)
plan.set_option(plan.OPT_FUSED_VARIANT, 1)
for grid in (240, 512):     # fewer / more persistent workgroups than compute units
    plan.set_option(plan.OPT_FUSED_STAGGER_US, 0); plan.set_option(plan.OPT_FUSED_PREFETCH, 2); plan.set_option(plan.OPT_FUSED_GRID, grid)
    ms = bench.ev_ms(lambda: fused(CH, buf), reps=3, warm=1)
    res["rows"].append({"launch_scans": CH * B, "grid": grid, "ms_per_1024": ms / CH * 1024 / B})
    print(res["rows"][-1], flush=True)
plan.set_option(plan.OPT_FUSED_GRID, 0)
best = min(res["rows"], key=lambda r: r["ms_per_1024"])
res["best"] = best
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "ab_fused.json"), "w"), indent=1)
print("best:", best, "vs separate", res["separate_ms_per_1024"])
