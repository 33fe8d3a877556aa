"""One registration at a time through the pygicp drop-in (the node's shape: main_RING.py:81-104), down-sampled clouds of the bench: host
time per phase.  Development aid; run under rocprofv3 --kernel-trace --stats for the kernel side."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mr_slam_amd.compat import pygicp
srcs, tgts = bench._gicp_pairs(1, 0)
s = pygicp.downsample(srcs[0].astype(np.float64), 0.2); t = pygicp.downsample(tgts[0].astype(np.float64), 0.2)
print("points", s.shape[0], t.shape[0])
acc = {}
def tick(name, t0):
    torch.cuda.synchronize(); acc.setdefault(name, []).append(1e3 * (time.perf_counter() - t0))
for rep in range(22):
    t0 = time.perf_counter(); g = pygicp.FastGICP(); tick("create", t0)
    t0 = time.perf_counter(); g.set_input_target(t); tick("set_target", t0)
    t0 = time.perf_counter(); g.set_input_source(s); tick("set_source", t0)
    g.set_max_correspondence_distance(5.0)
    t0 = time.perf_counter(); T = g.align(initial_guess=np.eye(4)); tick("align", t0)
    t0 = time.perf_counter(); f = g.get_fitness_score(1.0); tick("fitness", t0)
print({k: round(float(np.median(v[2:])), 3) for k, v in acc.items()}, "total", round(sum(float(np.median(v[2:])) for v in acc.values()), 3))
