"""Where the time of ONE pygicp-style registration goes (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mr_slam_amd import synth
from mr_slam_amd.compat import pygicp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
srcs, tgts = bench._gicp_pairs(1, 0)                      # metre-scale 120k-point pair of the bench's GICP leg
src = pygicp.downsample(srcs[0].astype(np.float64), 0.2); tgt = pygicp.downsample(tgts[0].astype(np.float64), 0.2)
print("points", len(src), len(tgt))


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


g = pygicp.FastGICP()
print(f"set_input_target {t(lambda: g.set_input_target(tgt)):.3f} ms")
print(f"set_input_source {t(lambda: g.set_input_source(src)):.3f} ms")
g.set_max_correspondence_distance(5.0)
print(f"align            {t(lambda: g.align()):.3f} ms  iterations {g._its}")
print(f"fitness          {t(lambda: g.get_fitness_score(1.0)):.3f} ms")
def whole():
    h = pygicp.FastGICP(); h.set_input_target(tgt); h.set_input_source(src); h.set_max_correspondence_distance(5.0); h.align(); return h.get_fitness_score(1.0)
print(f"whole            {t(whole):.3f} ms")
