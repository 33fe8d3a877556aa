#!/usr/bin/env python3
"""A/B of a development switch of the fused descriptor kernel on bench-shaped scans, in one process on one box: same bits, time per 1024 scans.
  MRS_DEV=1 python tools/quick_fused_ab.py MRS_FUSED_OVL 0 1 [--chunks 16]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mr_slam_amd import ring  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("var")
ap.add_argument("values", nargs="+")
ap.add_argument("--chunks", type=int, default=16)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--odd", action="store_true", help="an odd number of scans (the last pair holds one scan)")
args = ap.parse_args()
dev = "cuda:0"
B, CH, N = args.batch, args.chunks, bench.N_POINTS
bench.make_shard(B, CH, 0, dev)
whole = bench.make_shard.whole
nscan = CH * B - (1 if args.odd else 0)
offs = torch.arange(nscan + 1, dtype=torch.int64, device=dev) * N
flat = whole[:CH].view(-1)[: nscan * 3 * N]


def fused(out):
    return ring.ring_descriptors_fused(flat, offs, raw=False, normalized=True, out_norm=out)[2]


os.environ.pop(args.var, None)
ref = fused(None).clone()
buf = torch.empty_like(ref)
for rep in range(2):
    for v in args.values:
        os.environ[args.var] = v
        buf.zero_()
        got = fused(buf)
        torch.cuda.synchronize()
        same = bool(torch.equal(ref.view(torch.int32), got.view(torch.int32)))
        ms = bench.ev_ms(lambda: fused(buf), reps=5, warm=1)
        print({args.var: v, "bit_identical": same, "mismatching_sinograms": int((ref.view(nscan, -1) != got.view(nscan, -1)).any(1).sum()),
               "ms_per_1024": ms / nscan * 1024}, flush=True)
