"""Development aid: covariance (exact kNN k = 15) time for 2 x P clouds of 120k points, and the RING++ point-feature front end (k = 30)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mr_slam_amd import gicp, pointfeat
P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
srcs, tgts = bench._gicp_pairs(P, 0)
b = gicp.GicpBatch(P, 0)
b.set_params(k_correspondences=15, max_correspondence_distance=5.0)
b.set_sources(srcs); b.set_targets(tgts)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b.compute_covariances(0); b.compute_covariances(1)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"covariances: {2 * P} clouds in {t * 1e3:.1f} ms = {1e3 * t / (2 * P) * 256:.1f} ms per 256 clouds, {2 * P / t:.0f} clouds/s", flush=True)
    b.set_sources(srcs); b.set_targets(tgts)
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
pts = torch.from_numpy(np.concatenate([s for s in srcs[:S]])).cuda()
offs = np.arange(S + 1, dtype=np.int64) * srcs[0].shape[0]
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = pointfeat.point_features(pts, offs, 30, want=("planes",))
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"point features k=30: {S} scans in {t * 1e3:.1f} ms = {S / t:.0f} scans/s", flush=True)
