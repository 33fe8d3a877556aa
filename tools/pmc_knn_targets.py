"""Targets of the k-NN counter passes (development aid): the RING++ front end on 16 scans (k = 30) and the GICP covariances of 32 clouds (k = 15), a few
launches each.  Run under rocprofv3 --kernel-trace --stats or --pmc ...; tools/run_r05_s.sh"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mr_slam_amd import gicp, pointfeat

S = 16
bench.make_shard(S, 1, 0, "cuda:0")
pts = bench.make_shard.whole[0, :S].permute(0, 2, 1).reshape(S * bench.N_POINTS, 3).contiguous()
offs = np.arange(S + 1, dtype=np.int64) * bench.N_POINTS
for _ in range(4):
    pointfeat.point_features(pts, offs, 30, want=("planes",))
torch.cuda.synchronize()
P = 32
srcs, tgts = bench._gicp_pairs(P, 0)
b = gicp.GicpBatch(P)
b.set_params(k_correspondences=15, max_correspondence_distance=5.0)
b.set_sources(srcs)
for _ in range(4):
    b.compute_covariances(0)
torch.cuda.synchronize()
