#!/usr/bin/env python3
"""bench.py -- loop-closure hot path throughput on N MI355X of one node.

One "step" = one pass of the hot path over one batch of synthetic 120 000-point scans:
  B scan pairs per rank:  Cartesian BEV (A3/A4) -> Radon sinogram + normalisation (R1/R2) ->
  rotation correlation of every new descriptor with its loop candidate (C1)
  [N > 1: + RCCL all-gather of the new descriptors so that every rank holds the whole DB].
value = loop-candidate pairs/s summed over ranks (each pair includes building the descriptor of
a 120k-point scan; the candidate descriptor comes from the resident database).
Extra fields: database-sweep rate (pairs/s with descriptors resident), per-stage kernel times,
the HBM roofline of the BEV scatter kernel and the CPU baseline (oracle port, rank 0, N=1).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mr_slam_amd import bev, ring, synth  # noqa: E402

N_POINTS = 120_000
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy ceiling)


def make_batch(batch, rank, device):
    """`batch` distinct 120k-point scans: 4 ray-cast base scenes per rank, the rest are rigidly
    rotated/translated copies (cheap to generate, different cell pattern each)."""
    rng = np.random.default_rng(1000 + rank)
    base = [synth.lidar_scan(100 * rank + s, N_POINTS, metric=True) for s in range(4)]
    scans = []
    for i in range(batch):
        p = base[i % 4]
        if i >= 4:
            th = rng.uniform(0, 2 * np.pi)
            R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], np.float32)
            p = p @ R.T + np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), 0], np.float32)
        q = synth.preprocess(p)
        if q.shape[0] < N_POINTS:   # rotation pushed a few points out of the crop: pad by repeating
            q = np.concatenate([q, q[: N_POINTS - q.shape[0]]])
        scans.append(q[:N_POINTS])
    return bev.pack_scans(scans, device), scans


def cpu_baseline(scans, n_sample=512):
    """The same workload on the host cores with the oracle port (BEV restatement in C, Radon
    restatement in C + OpenMP, fast_corr restatement on torch CPU).  Bounded sample."""
    from oracle import pyoracle as O
    from oracle import corr_oracle as K
    cores = min(os.cpu_count() or 1, 16)       # tiny FFTs do not scale past a few threads
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    ang = np.linspace(0, 2 * np.pi, 120).astype(np.float32)
    sample = scans[:n_sample]
    soas = [synth.to_soa(s) for s in sample]
    # warm-up (library load, thread pools)
    w = O.bev_cart(soas[0], 1, 1, 120, 120, 1).reshape(-1, 3)[:, 2].reshape(1, 120, 120)
    ws = O.radon_parallel(w, ang, 120, 1.0)
    wt = K.tiring_from_sinogram(ws)
    K.fast_corr(wt, wt)
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    # the reference rasteriser is single-threaded per scan; scans are spread over the cores (ctypes drops the GIL)
    with ThreadPoolExecutor(cores) as ex:
        imgs = np.stack(list(ex.map(lambda s: O.bev_cart(s, 1, 1, 120, 120, 1).reshape(-1, 3)[:, 2].reshape(120, 120), soas)))
    t1 = time.perf_counter()
    sino = O.radon_parallel(imgs, ang, 120, 1.0)
    t2 = time.perf_counter()
    tir = [K.tiring_from_sinogram(s[None]) for s in sino]
    for i in range(len(tir)):
        K.fast_corr(tir[i], tir[(i + 1) % len(tir)])
    t3 = time.perf_counter()
    out = {"value": len(sample) / (t3 - t0), "unit": "pairs/s", "cores": cores, "kind": "port",
           "sample": f"{len(sample)} scans x 120k pts: C BEV restatement (1 thread per scan, scans over {cores} threads), "
                     f"C Radon restatement (OpenMP over images), torch-CPU fast_corr ({cores} threads)",
           "ms_per_pair": {"bev": 1e3 * (t1 - t0) / len(sample), "radon": 1e3 * (t2 - t1) / len(sample),
                           "fft_corr": 1e3 * (t3 - t2) / len(sample)}}
    out["gicp"] = cpu_gicp_baseline(cores)
    if O.ref_polar() is not None:   # the reference's own CPU polar rasteriser, unmodified
        t0 = time.perf_counter()
        for s in soas:
            O.ref_bev_polar(s, 1, 1, 40, 120, 20, 1)
        out["reference_polar_bev_scans_per_s"] = len(sample) / (time.perf_counter() - t0)
    return out


def cpu_gicp_baseline(cores, iters=20):
    """fast_gicp restatement (kd-tree + OpenMP, oracle/gicp_oracle.cpp) on ONE 120k x 120k pair of the GICP leg's
    shape: `iters` forced outer iterations, k = 15, max_corr 5.0; covariances timed separately, like the GPU leg."""
    from oracle import pyoracle as O
    from scipy.spatial.transform import Rotation as Rot
    rng = np.random.default_rng(2000)
    p = synth.lidar_scan(500, N_POINTS, metric=True).astype(np.float64)
    R = Rot.from_rotvec([0.01, -0.02, 0.04]).as_matrix()
    src = (p + rng.normal(0, 0.02, p.shape)).astype(np.float32)
    tgt = (p @ R.T + [0.4, -0.3, 0.05] + rng.normal(0, 0.02, p.shape)).astype(np.float32)
    g = O.Gicp(k=15, max_corr=5.0, threads=cores)
    t0 = time.perf_counter()
    g.set_source(src); g.set_target(tgt)        # builds both kd-trees
    t1 = time.perf_counter()
    g.covariances(0); g.covariances(1)
    t2 = time.perf_counter()
    _, _, its, trials = g.align(np.eye(4), force_iters=iters)
    t3 = time.perf_counter()
    return {"iters_per_s": its / (t3 - t2), "iterations": its, "lm_trials": trials, "align_s": t3 - t2,
            "covariance_clouds_per_s": 2 / (t2 - t1), "kdtree_build_s": t1 - t0, "cores": cores,
            "sample": f"1 pair x 120k pts, {its} forced iterations, k=15, max_corr 5.0 (restated fast_gicp, kd-tree + OpenMP)"}


def gicp_leg(device_index, rank, n_pairs, iters):
    """BASELINE configs[2] shape: submap pairs x 120k points, `iters` outer iterations with the
    convergence test disabled; covariances are timed separately (they are cached per submap)."""
    from mr_slam_amd import gicp
    from scipy.spatial.transform import Rotation as Rot
    rng = np.random.default_rng(2000 + rank)
    base = [synth.lidar_scan(500 + 10 * rank + s, N_POINTS, metric=True) for s in range(2)]
    srcs, tgts = [], []
    for i in range(n_pairs):
        p = base[i % 2].astype(np.float64)
        ang = np.deg2rad(rng.uniform(0, 5))
        axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
        R = Rot.from_rotvec(ang * axis).as_matrix()
        t = rng.normal(size=3); t *= rng.uniform(0, 1) / np.linalg.norm(t)
        srcs.append((p + rng.normal(0, 0.02, p.shape)).astype(np.float32))
        tgts.append((p @ R.T + t + rng.normal(0, 0.02, p.shape)).astype(np.float32))
    b = gicp.GicpBatch(n_pairs, device_index)
    b.set_params(k_correspondences=15, max_correspondence_distance=5.0, force_iterations=iters)
    b.set_sources(srcs)
    b.set_targets(tgts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b.compute_covariances(0)
    b.compute_covariances(1)
    torch.cuda.synchronize()
    t_cov = time.perf_counter() - t0
    b.set_params(force_iterations=2)
    b.align()                                   # warm-up (2 iterations)
    b.set_params(force_iterations=iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T, conv, its = b.align()
    torch.cuda.synchronize()
    t_align = time.perf_counter() - t0
    assert (its == iters).all()
    return {"pairs": n_pairs, "iterations": iters, "points": N_POINTS,
            "iters_per_s": n_pairs * iters / t_align, "align_s": t_align,
            "nn_passes": b.nn_passes, "nn_pass_ms": 1e3 * t_align / (n_pairs * b.nn_passes),
            "nn_search": "exact brute force over Morton-ordered LDS tiles with conservative bounding-box culling",
            "covariance_s": t_cov, "covariance_clouds_per_s": 2 * n_pairs / t_cov, "k": 15,
            "max_correspondence_distance": 5.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="scan pairs per rank per step")
    ap.add_argument("--db", type=int, default=16384, help="database size for the sweep-rate leg (x 58 560 B)")
    ap.add_argument("--gicp-pairs", type=int, default=256, help="120k-pt pairs per rank in the GICP leg (BASELINE configs[2]: 256; 0 = skip)")
    ap.add_argument("--gicp-iters", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or os.environ.get("MRS_BENCH_FORCE_DIST") == "1"   # the env var exercises the RCCL path on 1 GPU
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device(device))

    B = args.batch
    (xyz, offs), scans = make_batch(B, rank, device)
    plan = ring.ring_plan(local_rank)
    img = torch.empty((B, 1, 120, 120), dtype=torch.float32, device=device)
    # resident database of candidate descriptors (one candidate per new scan)
    # database entries are Hermitian half spectra [61][120] complex64 (58 560 B) of the normalised sinograms
    _, _, cand = ring.ring_descriptors(xyz, offs)
    cand = ring.half_spectrum(cand).roll(1, 0).contiguous()
    out_dist = torch.empty(B, dtype=torch.float32, device=device)
    out_ang = torch.empty(B, dtype=torch.int32, device=device)
    # N > 1: the all-gather of step i overlaps the kernels of step i+1 (async RCCL op on its own stream,
    # double-buffered destination; the gathered descriptors only feed the database, not the next step)
    # exchange format: fp16 replicas (29 280 B per descriptor; the owner keeps the exact fp32 entry) -- at >1 M
    # descriptors/s/GPU the fp32 spectra would exceed what the xGMI links carry
    gathered = [torch.empty((world * B, 61, 120, 2), dtype=torch.float16, device=device) for _ in range(2)] if dist_on else None
    pending = {"work": None, "keep": None, "n": 0}

    ev = {k: [] for k in ("bev", "radon", "corr")}

    def step(record):
        def mark():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        e0 = mark() if record else None
        bev.cart_bev(xyz, offs, 1, 1, 120, 120, 1, out=img.view(B, -1))
        e1 = mark() if record else None
        _, norm = plan.forward(img.view(B, 120, 120), raw=False, normalized=True)
        e2 = mark() if record else None
        # half spectrum of the new descriptors (kept: they are the next database entries; fp16 replica for the
        # other ranks) + correlation with the candidates, one launch
        spec, spec16, _, _ = ring.spectrum_corr_pairs(norm, cand, want_f16=dist_on, out=(out_dist, out_ang))
        e3 = mark() if record else None
        if dist_on:
            if pending["work"] is not None:
                pending["work"].wait()
            pending["keep"] = spec16                    # keep the source alive until the op completes
            pending["work"] = dist.all_gather_into_tensor(gathered[pending["n"] & 1], spec16, async_op=True)
            pending["n"] += 1
        if record:
            ev["bev"].append((e0, e1)); ev["radon"].append((e1, e2)); ev["corr"].append((e2, e3))

    def fence():
        if dist_on:
            if pending["work"] is not None:
                pending["work"].wait()
                pending["work"] = None
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kern_ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()}

    # database sweep leg (descriptors resident): 8 queries against a DB of args.db entries
    sweep = None
    if rank == 0:
        nq = 4
        db = cand[torch.arange(args.db, device=device) % B].contiguous()   # args.db * 58 560 B (> 256 MB L3 by default)
        q = cand[:nq].contiguous()
        for _ in range(2):
            ring.corr_sweep_fft(q, db)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            ring.corr_sweep_fft(q, db)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        sweep = {"pairs_per_s": nq * args.db / ms * 1e3, "db": args.db, "queries": nq, "ms": ms,
                 "bytes_per_pair": 58560, "algorithmic_gbs": nq * args.db * 58560 / ms / 1e6,
                 "kernel": "k_ring_corr_fft (half-spectrum DB, in-register real FFT-120)"}

    gicp_res = None
    if args.gicp_pairs > 0:
        fence()
        gicp_res = gicp_leg(local_rank, rank, args.gicp_pairs, args.gicp_iters)
        if dist_on:   # whole-job GICP rate: all ranks' iterations / slowest rank's time
            t = torch.tensor([gicp_res["align_s"]], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            gicp_res["iters_per_s"] = world * args.gicp_pairs * args.gicp_iters / float(t.item())
            gicp_res["pairs"] = world * args.gicp_pairs

    if rank == 0:
        cells = 120 * 120
        bev_bytes = B * (12 * N_POINTS + 4 * cells)          # 12 B/point + 4 B/cell
        achieved = bev_bytes / (kern_ms["bev"] * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected
        # per MI355X_MICROARCH.md; counters cannot be read from inside this process)
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            traffic = pmc.get(f"k_cart_lds@grid{B * 1024}", {}).get("hbm_bytes")
        except OSError:
            pass
        line = {
            "metric": "loop-candidate pairs/sec (BEV+Radon+corr), 120k-pt scans",
            "value": world * B * args.steps / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1] batched: 120k-pt synthetic lidar scans -> Cartesian BEV "
                                   "120x120x1 -> Radon 120x120 -> normalise -> half spectrum -> FFT-domain rotation correlation vs 1 candidate",
                       "pairs_per_rank_per_step": B, "points_per_scan": N_POINTS,
                       "parallelism": f"scan-sharded x{world}" + (" + RCCL all-gather of fp16 descriptor replicas" if dist_on else "")},
            "kernel_ms": kern_ms,
            "roofline": {"kernel": "k_cart_lds (BEV scatter)", "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": bev_bytes,
                         "traffic_source": "profiles/r01_pmc_traffic.json (PMC pass of this command)" if traffic else None},
            "sweep": sweep,
            "gicp": gicp_res,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scans)
    else:
        line = None
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        sys.stdout.flush()
        sys.stderr.flush()
        try:   # librccl printf()s a banner into libc's stdout buffer, which would otherwise be flushed at exit, after the JSON
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)     # the ONE JSON line, last thing on stdout


if __name__ == "__main__":
    main()
