"""Timing of the k-NN selection and its two tails (development aid, round 5):
  RING++ front end, 64 scans of 120 k points, k = 30 (pointfeat.point_features -> k_knn_cov<30> + k_feat_from_knn, incl. the Morton sort)
  GICP covariances, 256 clouds, k = 15 (k_knn_cov<16> + k_cov_from_knn)
usage: quick_knn.py [--dbg]   (--dbg: one call of each with MRS_DEV=1 MRS_KNN_DBG=1 set by the caller: the counters go to stderr)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mr_slam_amd import gicp, pointfeat

S = 64
bench.make_shard(S, 1, 0, "cuda:0")
pts = bench.make_shard.whole[0, :S].permute(0, 2, 1).reshape(S * bench.N_POINTS, 3).contiguous()
offs = np.arange(S + 1, dtype=np.int64) * bench.N_POINTS
out = {}
pointfeat.point_features(pts[:8 * bench.N_POINTS], offs[:9], 30, want=("planes",)); torch.cuda.synchronize()
reps = 1 if "--dbg" in sys.argv else 4
ts = []
for _ in range(reps):
    torch.cuda.synchronize(); t = time.perf_counter()
    pl = pointfeat.point_features(pts, offs, 30, want=("planes",)); torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t))
out["feat_ms_64_scans"] = ts
out["feat_checksum"] = float(pl["planes"].double().nan_to_num(0.0, 0.0, 0.0).sum())
P = 256
srcs, tgts = bench._gicp_pairs(P, 0)
b = gicp.GicpBatch(P)
b.set_params(k_correspondences=15, max_correspondence_distance=5.0)
b.set_sources(srcs); b.set_targets(tgts); torch.cuda.synchronize()
ts = []
for _ in range(reps):
    torch.cuda.synchronize(); t = time.perf_counter(); b.compute_covariances(0); torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t))
out["cov_ms_256_clouds_k15"] = ts
out["cov_checksum"] = float(np.abs(b.covariances(0)).sum())
if "--dbg" not in sys.argv:
    b.compute_covariances(1)
    ms, cnt = b.profile(np.tile(np.eye(4), (P, 1, 1)), reps=3)
    out["profile_ms"] = {k: round(v, 4) for k, v in ms.items() if k in ("knn_select", "cov_from_knn")}
print(json.dumps(out))
