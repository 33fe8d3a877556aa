#!/usr/bin/env python3
"""Development: wall time of one pygicp registration of the drop-in latency leg's shape (two ~35-40 k-point voxel-downsampled clouds), split
into its calls, `reps` times.  python tools/quick_pygicp_latency.py [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mr_slam_amd.compat import pygicp  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    reuse = len(sys.argv) > 2 and sys.argv[2] == "reuse"           # one registration object fed new clouds (the Mapping node's shape) instead of a new one each time
    srcs, tgts = bench._gicp_pairs(1, 0)
    s, t = pygicp.downsample(srcs[0].astype(np.float64), 0.2), pygicp.downsample(tgts[0].astype(np.float64), 0.2)
    tm = {k: [] for k in ("ctor", "set_target", "set_source", "align", "fitness", "total")}
    T = None
    for r in range(reps + 3):
        t0 = time.perf_counter()
        if not reuse or r == 0:
            g = pygicp.FastGICP()
        t1 = time.perf_counter()
        g.set_input_target(t)
        t2 = time.perf_counter()
        g.set_input_source(s); g.set_max_correspondence_distance(5.0)
        t3 = time.perf_counter()
        T = g.align(initial_guess=np.eye(4))
        t4 = time.perf_counter()
        f = g.get_fitness_score(1.0)
        t5 = time.perf_counter()
        if r >= 3:
            for k, v in zip(tm, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)):
                tm[k].append(1e3 * v)
    print("points", s.shape[0], t.shape[0], "fitness", f)
    print({k: round(float(np.median(v)), 3) for k, v in tm.items()})
    print(np.array2string(T, precision=6))


if __name__ == "__main__":
    main()
