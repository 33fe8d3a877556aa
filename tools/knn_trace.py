"""Workgroup timeline of one k_knn_cov<30> launch (development aid): MRS_DEV=1 MRS_KNN_DBG=1 MRS_KNN_TRACE_FILE=/tmp/knn_trace.bin.
Prints duration percentiles, the longest workgroups and how many workgroups are alive over time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mr_slam_amd import pointfeat
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
bench.make_shard(64, 1, 0, "cuda:0")
pts = bench.make_shard.whole[0, :S].permute(0, 2, 1).reshape(S * bench.N_POINTS, 3).contiguous()
offs = np.arange(S + 1, dtype=np.int64) * bench.N_POINTS
for _ in range(2):
    pointfeat.point_features(pts, offs, 30, want=("planes",)); torch.cuda.synchronize()
tr = np.fromfile(os.environ["MRS_KNN_TRACE_FILE"], dtype=np.uint64).reshape(-1, 2)
gx = (bench.N_POINTS + 255) // 256
tr = tr[:gx * S].astype(np.int64)
ok = (tr[:, 0] > 0) & (tr[:, 1] > 0)
t0 = tr[ok, 0].min()
st = (tr[:, 0] - t0) / 100.0; en = (tr[:, 1] - t0) / 100.0          # microseconds
dur = en - st
print(f"S={S}: {ok.sum()} workgroups traced; launch span {en[ok].max():.0f} us; duration us: median {np.median(dur[ok]):.0f} p90 {np.percentile(dur[ok], 90):.0f} "
      f"p99 {np.percentile(dur[ok], 99):.0f} max {dur[ok].max():.0f}; sum {dur[ok].sum() / 1e3:.1f} ms = {dur[ok].sum() / en[ok].max():.0f} workgroups alive on average")
top = np.argsort(-np.where(ok, dur, 0))[:10]
for w in top:
    print(f"   wg {w} (cloud {w // gx}, block {w % gx}): start {st[w]:.0f} end {en[w]:.0f} dur {dur[w]:.0f}")
span = en[ok].max()
edges = np.linspace(0, span, 21)
alive = [int(((st[ok] <= t) & (en[ok] > t)).sum()) for t in edges[:-1] + span / 40]
print("   alive at 20 instants:", alive)
last_start = st[ok].max()
print(f"   last workgroup starts at {last_start:.0f} us; ends of the last 10: {np.sort(en[ok])[-10:].round(0)}")
