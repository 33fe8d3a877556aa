"""Quick BEV timing on one GPU (development aid; the judged numbers come from bench.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mr_slam_amd import bev, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for kind in ("lidar", "uniform"):
    base = [synth.lidar_scan(s) if kind == "lidar" else synth.uniform_scan(s) for s in range(4)]
    xyz, offs = bev.pack_scans([base[i % 4] for i in range(B)], "cuda:0")
    for name, fn, cells in (("cart120x120x1", lambda: bev.cart_bev(xyz, offs, 1, 1, 120, 120, 1), 14400),
                            ("polar40x120x20", lambda: bev.polar_bev(xyz, offs, 1, 1, 40, 120, 20), 96000),
                            ("polar120x120x1", lambda: bev.polar_bev(xyz, offs, 1, 1, 120, 120, 1), 14400)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        byts = B * (12 * 120000 + 4 * cells)
        print(f"{kind:8s} {name:16s} B={B} {ms:8.3f} ms  {B/ms*1e3:10.0f} scans/s  {byts/ms/1e6:8.1f} GB/s")
