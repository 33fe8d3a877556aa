"""Single-pair GICP latency breakdown (development aid): what one FastGICP.align() call costs end to end."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.spatial.transform import Rotation as Rot
from mr_slam_amd import gicp, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(0)
base = synth.lidar_scan(3, n, metric=True).astype(np.float64)
R = Rot.from_rotvec([0.01, -0.02, 0.05]).as_matrix()
src = [(base + rng.normal(0, 0.02, base.shape)).astype(np.float32) for _ in range(npairs)]
tgt = [(base @ R.T + [0.5, -0.3, 0.05] + rng.normal(0, 0.02, base.shape)).astype(np.float32) for _ in range(npairs)]


def sync():
    torch.cuda.synchronize()


for rep in range(3):
    t0 = time.perf_counter()
    b = gicp.GicpBatch(npairs)
    b.set_params(k_correspondences=15, max_correspondence_distance=5.0)
    t1 = time.perf_counter()
    b.set_sources(src); sync()
    t2 = time.perf_counter()
    b.set_targets(tgt); sync()
    t3 = time.perf_counter()
    b.compute_covariances(0); b.compute_covariances(1); sync()
    t4 = time.perf_counter()
    T, conv, its = b.align(); sync()
    t5 = time.perf_counter()
    f = b.fitness(T, 1.0) if hasattr(b, "fitness") else None
    sync()
    t6 = time.perf_counter()
    print(f"rep {rep}: create {1e3*(t1-t0):.2f}  set_src {1e3*(t2-t1):.2f}  set_tgt {1e3*(t3-t2):.2f}  cov {1e3*(t4-t3):.2f}  "
          f"align {1e3*(t5-t4):.2f} ({its.tolist()[:4]} it, conv {conv.tolist()[:4]}, nn {b.nn_passes})  fitness {1e3*(t6-t5):.2f}  total {1e3*(t6-t0):.2f} ms")
    del b
