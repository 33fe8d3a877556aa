"""RING++ front end at 8 / 16 / 32 / 64 scans (development aid): is the time per scan constant?"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mr_slam_amd import pointfeat
S = 64
bench.make_shard(S, 1, 0, "cuda:0")
pts = bench.make_shard.whole[0, :S].permute(0, 2, 1).reshape(S * bench.N_POINTS, 3).contiguous()
offs = np.arange(S + 1, dtype=np.int64) * bench.N_POINTS
out = {}
for n in (8, 16, 32, 64, 16, 64):
    pointfeat.point_features(pts[:n * bench.N_POINTS], offs[:n + 1], 30, want=("planes",)); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        pointfeat.point_features(pts[:n * bench.N_POINTS], offs[:n + 1], 30, want=("planes",)); torch.cuda.synchronize()
        ts.append(round(1e3 * (time.perf_counter() - t), 3))
    out.setdefault(n, []).append(ts)
print(json.dumps(out))
