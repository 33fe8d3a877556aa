"""Workgroup timeline of one k_nn_scan<2> launch over every source point of 256 (or argv[1]) pairs, warm seeds (mrs_gicp_batch_profile's
'search_round3_all'); development aid: MRS_DEV=1 MRS_KNN_DBG=1 MRS_NN_TRACE_FILE=/tmp/nn_trace.bin."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mr_slam_amd import gicp
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
srcs, tgts = bench._gicp_pairs(P, 0)
b = gicp.GicpBatch(P)
b.set_params(k_correspondences=15, max_correspondence_distance=5.0)
b.set_sources(srcs); b.set_targets(tgts); b.compute_covariances(0); b.compute_covariances(1)
T, conv, its = b.align()
ms, cnt = b.profile(T, reps=2)
print({k: round(v, 3) for k, v in ms.items()})
tr = np.fromfile(os.environ["MRS_NN_TRACE_FILE"], dtype=np.uint64).reshape(-1, 2).astype(np.int64)
gx = (bench.N_POINTS + 1023) // 1024 if os.environ.get("MRS_NN_TRACE_KERNEL") == "4" else (bench.N_POINTS + 511) // 512
tr = tr[:gx * P]
ok = (tr[:, 0] > 0) & (tr[:, 1] > 0)
t0 = tr[ok, 0].min()
st = (tr[:, 0] - t0) / 100.0; en = (tr[:, 1] - t0) / 100.0
dur = en - st
print(f"P={P}: {ok.sum()} workgroups; launch span {en[ok].max():.0f} us; life us: median {np.median(dur[ok]):.0f} p90 {np.percentile(dur[ok], 90):.0f} p99 {np.percentile(dur[ok], 99):.0f} "
      f"max {dur[ok].max():.0f}; {dur[ok].sum() / en[ok].max():.0f} alive on average")
for w in np.argsort(-np.where(ok, dur, 0))[:6]:
    print(f"   wg {w} (pair {w // gx}, block {w % gx}): start {st[w]:.0f} end {en[w]:.0f} life {dur[w]:.0f}")
span = en[ok].max()
print("   alive at 20 instants:", [int(((st[ok] <= t) & (en[ok] > t)).sum()) for t in np.linspace(0, span, 21)[:-1] + span / 40])
