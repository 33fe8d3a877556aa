#!/usr/bin/env python3
"""Development: the batched voxel grid (mrs_voxel_downsample_batch) on 64 raw 130 k-point clouds, as bench.py's ingest legs feed it; wall time per
batch, and a run under rocprofv3 --kernel-trace --stats shows its kernels.  python tools/quick_voxel_batch.py [reps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mr_slam_amd import preprocess, synth  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    R = 64
    base = [synth.lidar_scan(900 + s, 130_000, metric=True) for s in range(4)]
    raws = []
    for i in range(R):
        p = base[i % 4]
        th = 0.1 * i
        c, sn = np.float32(np.cos(th)), np.float32(np.sin(th))
        q = np.empty((p.shape[0], 4), np.float32)
        q[:, 0] = c * p[:, 0] - sn * p[:, 1]; q[:, 1] = sn * p[:, 0] + c * p[:, 1]; q[:, 2] = p[:, 2]; q[:, 3] = 0.5
        raws.append(torch.from_numpy(q).cuda())
    cat = torch.cat(raws)
    offs = np.concatenate([[0], np.cumsum([r.shape[0] for r in raws])]).astype(np.int64)
    out, doffs = preprocess.voxel_down_sample_batch(cat, offs, 0.2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out, doffs = preprocess.voxel_down_sample_batch(cat, offs, 0.2)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    print(f"voxel_down_sample_batch: {1e3 * t:.3f} ms per batch of {R} scans ({R / t:.0f} scans/s), {int(doffs[-1]) / R:.0f} centroids per scan, "
          f"checksum {float(out.double().sum()):.6f}")


if __name__ == "__main__":
    main()
