"""One worker process of the all-core CPU baseline (bench.py's `cpu_baseline` leg; TEST / MEASUREMENT INFRASTRUCTURE like the rest of oracle/).

The host of a GPU box has 256 hardware threads; thousands of tiny FFTs and a kd-tree walk do not scale to that inside ONE process (2.6 pairs/s at
256 torch threads against 1 600 at 16).  What a deployment would do instead: one worker per 8 cores, each with the reference's own thread count
(main_RING.py:93 uses 4, global_manager.cpp:2438 uses 8), the sample dealt to the workers.  bench.py starts nproc / 8 of these, pinned to
disjoint core ranges, all released at the same wall-clock instant, and adds their rates.

  python -m oracle.cpu_worker <mode: ring|gicp> <input .npz> <worker index> <threads> <start time (time.time())> [iters]
prints one JSON line: {"units": n, "seconds": t, ...}"""
import json
import os
import sys
import time


def main():
    mode, path, widx, threads, t_start = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
    try:
        ncpu = os.cpu_count() or 1
        cores = [c % ncpu for c in range(widx * threads, (widx + 1) * threads)]
        os.sched_setaffinity(0, cores)
    except (AttributeError, OSError):
        pass
    os.environ["OMP_NUM_THREADS"] = str(threads)
    # no OMP_PROC_BIND: libgomp would pin the main thread to ONE place when it starts, and the Python thread pool of the rasteriser leg inherits
    # that mask (measured: 18 ms instead of 3.6 ms for 16 scans); the process mask above already keeps the worker on its 8 cores
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")      # idle OpenMP threads sleep: they share the worker's 8 cores with the thread pool below
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pyoracle as O
    data = np.load(path)
    if mode == "ring":
        import torch
        from concurrent.futures import ThreadPoolExecutor
        from oracle import corr_oracle as K
        # the correlation leg is thousands of 120-point FFTs: there is no intra-op parallelism to be had, and torch's parallel regions cost more
        # than the transforms (0.2 ms per pair on one thread; 16 ms at 2 threads, 368 ms at 256 on the same data).  One torch thread; the
        # parallelism of this leg is the number of worker processes.
        torch.set_num_threads(1)
        O.set_omp_threads(threads)
        soas = [data[k] for k in sorted(data.files)]
        ang = np.linspace(0, 2 * np.pi, 120).astype(np.float32)
        legs = [0.0, 0.0, 0.0]

        def work():
            t0 = time.perf_counter()
            with ThreadPoolExecutor(threads) as ex:
                imgs = np.stack(list(ex.map(lambda s: O.bev_cart(s, 1, 1, 120, 120, 1).reshape(-1, 3)[:, 2].reshape(120, 120), soas)))
            t1 = time.perf_counter()
            O.set_omp_threads(threads)     # torch re-applies ITS thread count (1) to the calling thread's OpenMP state inside every op
            sino = O.radon_parallel(imgs, ang, 120, 1.0)
            t2 = time.perf_counter()
            O.set_omp_threads(1)           # ... and torch's own parallel regions must not inherit the C legs' count
            tir = [K.tiring_from_sinogram(s[None]) for s in sino]
            for i in range(len(tir)):
                K.fast_corr(tir[i], tir[(i + 1) % len(tir)])
            legs[:] = [t1 - t0, t2 - t1, time.perf_counter() - t2]
        work()                                              # warm: libraries loaded, threads started, plans built, pages touched
        while time.time() < t_start:
            time.sleep(0.001)
        t0 = time.perf_counter()
        work()
        t = time.perf_counter() - t0
        print(json.dumps({"units": len(soas), "seconds": t, "late_s": max(0.0, time.time() - t - t_start), "bev_s": legs[0], "radon_s": legs[1], "fft_corr_s": legs[2]}))
    else:
        iters = int(sys.argv[6])
        g = O.Gicp(k=15, max_corr=5.0, threads=threads)
        src, tgt = data["src"], data["tgt"]
        while time.time() < t_start:
            time.sleep(0.001)
        t0 = time.perf_counter()
        g.set_source(src); g.set_target(tgt)
        t1 = time.perf_counter()
        g.covariances(0); g.covariances(1)
        t2 = time.perf_counter()
        _, _, its, trials = g.align(np.eye(4), force_iters=iters)
        t3 = time.perf_counter()
        print(json.dumps({"units": 1, "seconds": t3 - t0, "align_s": t3 - t2, "covariance_s": t2 - t1, "kdtree_build_s": t1 - t0, "iterations": int(its)}))


if __name__ == "__main__":
    main()
