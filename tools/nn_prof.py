"""phase profile of the round-4 1-NN kernel (development aid): MRS_DEV=1 MRS_NN_PROF=1 python tools/nn_prof.py [pairs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mr_slam_amd import gicp
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
srcs, tgts = bench._gicp_pairs(n_pairs, 0)
b = gicp.GicpBatch(n_pairs)
b.set_params(k_correspondences=15, max_correspondence_distance=5.0, force_iterations=8)
b.set_sources(srcs); b.set_targets(tgts)
b.compute_covariances(0); b.compute_covariances(1)
torch.cuda.synchronize()
print("align (8 forced iterations: pass 1 is cold)", file=sys.stderr)
b.align()
