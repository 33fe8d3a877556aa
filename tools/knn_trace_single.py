"""Workgroup timeline of k_knn_cov<20> over ONE down-sampled cloud (the node's registration shape); MRS_DEV=1 MRS_KNN_DBG=1 MRS_KNN_TRACE_FILE=..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mr_slam_amd import gicp
from mr_slam_amd.compat import pygicp
srcs, tgts = bench._gicp_pairs(1, 0)
s = pygicp.downsample(srcs[0].astype(np.float64), 0.2).astype(np.float32)
b = gicp.GicpBatch(1)
b.set_params(k_correspondences=20)
b.set_sources([s])
for _ in range(3):
    b.compute_covariances(0); torch.cuda.synchronize()
tr = np.fromfile(os.environ["MRS_KNN_TRACE_FILE"], dtype=np.uint64).reshape(-1, 2).astype(np.int64)
gx = (s.shape[0] + 255) // 256
tr = tr[:gx]
t0 = tr[:, 0].min(); st = (tr[:, 0] - t0) / 100.0; en = (tr[:, 1] - t0) / 100.0; dur = en - st
print(f"{s.shape[0]} points, {gx} workgroups; span {en.max():.0f} us; life us: median {np.median(dur):.0f} p90 {np.percentile(dur, 90):.0f} max {dur.max():.0f}")
print("   longest:", [(int(w), int(dur[w])) for w in np.argsort(-dur)[:8]])
print("   ends sorted (last 12):", np.sort(en)[-12:].round(0))
