#!/usr/bin/env python3
"""bench.py -- loop-closure hot path throughput on N MI355X of one node.

One "step" = one pass of the hot path over the rank's resident shard of synthetic 120 000-point scans
(`--chunks` launches of `--batch` scan pairs each; every scan is distinct and already in HBM):
  per launch:  Cartesian BEV (A3/A4) -> Radon sinogram + normalisation (R1/R2) -> half spectrum + rotation
  correlation of every new descriptor with its loop candidate (C1) -> one query swept against the database
  [N > 1: + RCCL all-gather of the new descriptors' fp16 replicas; the candidates and the swept database of the NEXT
   launch are rows of that replicated database; candidates whose replica score falls within 2e-3 of the acceptance
   threshold are re-scored exactly on the owner rank at the end of the step].
value = loop-candidate pairs/s summed over ranks (each pair includes building the descriptor of a 120k-point scan).

Besides the contract line this prints (same JSON object): per-stage kernel times, HBM rooflines of the Cartesian and the
polar BEV scatter, the Radon kernel's VALU / LDS figures, database-sweep legs in the shape of BASELINE configs[3]
(10 k-entry RING / RING++ / DiSCO databases, 1 and 4 queries), the single-GPU shard of configs[4] (RING++ database of
50 000 / 8 entries, one elevation-map frame, GICP), the GICP leg of configs[2] (20 forced iterations, a cold 5-iteration
run and a run to convergence), single-call drop-in latencies, and the CPU baseline (oracle port, rank 0, N = 1).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--chunks C]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mr_slam_amd import bev, ring, shard, synth  # noqa: E402

N_POINTS = 120_000
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy ceiling)
DIST_THRESHOLD = 0.48     # RING_ros/config.py:17
FUSE_DEFAULT = 16         # launches whose scans share one fused BEV + Radon kernel (0 = separate kernels): measured 1.97-1.99 M pairs/s
                          # at 8-24 against 1.76 M with the two kernels per launch
def _latest_pmc():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")))      # the newest round's counter passes
    return files[-1] if files else os.path.join(ROOT, "profiles", "r02_pmc.json")


PMC_FILE = _latest_pmc()
PMC_NAME = os.path.relpath(PMC_FILE, ROOT)


def make_shard(batch, chunks, rank, device):
    """chunks x batch DISTINCT pre-processed scans, built on the GPU: 4 ray-cast base scenes per rank (host, ~0.3 s each),
    every scan a rigidly rotated / translated copy pushed through the reference pre-processing (load_pc_infer,
    RING_ros/util.py:91-112: crop |x|,|y| < 70, 0 < z < 30, scale by 70 / 70 / 30); points that leave the crop are replaced
    by the scan's first surviving point so that every scan keeps exactly 120 000 points.  Returns a list of
    (xyz_soa, offsets) per launch, in the ragged SoA layout of the rasterisers."""
    base = torch.stack([torch.from_numpy(synth.lidar_scan(100 * rank + s, N_POINTS, metric=True)) for s in range(4)]).to(device)
    g = torch.Generator(device=device).manual_seed(1000 + rank)
    offs = torch.arange(batch + 1, dtype=torch.int64, device=device) * N_POINTS
    out = []
    sub = 64
    whole = torch.empty((chunks, batch, 3, N_POINTS), dtype=torch.float32, device=device)   # one allocation: launches over several chunks
    for c in range(chunks):
        xyz = whole[c]
        for i0 in range(0, batch, sub):
            n = min(sub, batch - i0)
            th = torch.rand(n, generator=g, device=device) * (2 * np.pi)
            t = torch.rand((n, 2), generator=g, device=device) * 6.0 - 3.0
            if c == 0 and i0 == 0:
                th[:4] = 0.0; t[:4] = 0.0                       # the base scenes themselves
            p = base[(torch.arange(n, device=device) + i0) % 4]            # [n, N, 3]
            cs, sn = torch.cos(th)[:, None], torch.sin(th)[:, None]
            x = cs * p[:, :, 0] - sn * p[:, :, 1] + t[:, 0:1]
            y = sn * p[:, :, 0] + cs * p[:, :, 1] + t[:, 1:2]
            z = p[:, :, 2]
            ok = (x.abs() < 70.0) & (y.abs() < 70.0) & (z < 30.0) & (z > 0.0)
            first = ok.float().argmax(1, keepdim=True)
            x = torch.where(ok, x, x.gather(1, first)); y = torch.where(ok, y, y.gather(1, first)); z = torch.where(ok, z, z.gather(1, first))
            xyz[i0:i0 + n, 0] = x / 70.0; xyz[i0:i0 + n, 1] = y / 70.0; xyz[i0:i0 + n, 2] = z / 30.0
        out.append((xyz.view(-1), offs))
    make_shard.whole = whole
    return out


def host_scans(xyz_soa, n):
    """the first n scans of a launch back on the host as [N,3] arrays (CPU baseline sample)"""
    a = xyz_soa.view(-1, 3, N_POINTS)[:n].permute(0, 2, 1).contiguous().cpu().numpy()
    return [a[i] for i in range(n)]


def ev_ms(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def valu_roofline(pmc_entry, ms_per_launch, clock_ghz=2.4, simds=256 * 4):
    """VALU-pipe floor of a kernel from its PMC instruction counts.  A wave64 VALU instruction occupies its SIMD for 2 cycles (fp32 add / mul,
    integer add, logic, moves), 4 (conversions, fract / floor / max, 24- and 32-bit multiplies, shift-adds, every packed-fp32 and fp64 operation)
    or 8 (rcp / sqrt): measured, profiles/r03_ubench.md.  No counter reports pipe-busy cycles, so the floor is the class-weighted cycle count of
    tools/pmc_summary.py (`valu_pipe_cycles_est`) / (SIMDs x clock) when the profile has the class counters, and always the two bounds
    insts x 2 and insts x 4.  Clock = the 2.4 GHz maximum, i.e. the most demanding floor; returns None without counters."""
    c = (pmc_entry or {}).get("counters") or {}
    n = c.get("SQ_INSTS_VALU")
    if not n or not ms_per_launch:
        return None
    per = simds * clock_ghz * 1e9 * 1e-3
    est = (pmc_entry or {}).get("valu_pipe_cycles_est")
    out = {"insts_valu_per_launch": float(n), "floor_ms_bounds": [float(n) * 2.0 / per, float(n) * 4.0 / per],
           "frac_bounds": [float(n) * 2.0 / per / ms_per_launch, float(n) * 4.0 / per / ms_per_launch], "clock_ghz": clock_ghz,
           "note": "wave VALU instructions x cycles per instruction / (1024 SIMDs x clock); cycles per instruction by class, profiles/r03_ubench.md"}
    if est:
        out["floor_ms"] = float(est) / per
        out["frac"] = out["floor_ms"] / ms_per_launch
        out["mean_cycles_per_inst"] = float(est) / float(n)
    return out


def load_pmc():
    try:
        return json.load(open(PMC_FILE))
    except (OSError, ValueError):
        return {}


# --------------------------------------------------------------------------------------------------- CPU baseline
def cpu_budget():
    """(hardware threads the OS reports, cores' worth of CPU time the container may use, where that limit comes from).  On the GPU boxes `nproc`
    says 256 while the cgroup grants 16 cores of time (cpu.max = "1600000 100000"): threads beyond the quota are throttled, not run."""
    nproc = os.cpu_count() or 1
    try:
        nproc = min(nproc, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota, src = float(nproc), "no cgroup CPU quota found"
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota, src = float(q) / float(per), f"cgroup v2 cpu.max = {q} {per}"
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota, src = q / per, f"cgroup v1 cfs_quota_us / cfs_period_us = {q} / {per}"
        except (OSError, ValueError):
            pass
    return nproc, max(1, int(min(nproc, quota))), src


def _cpu_workers(mode, inputs, threads, extra=()):
    """nproc / threads worker processes of oracle/cpu_worker.py, pinned to disjoint core ranges, released together; -> list of their JSON lines.
    inputs: one .npz path per worker."""
    import subprocess
    t_start = time.time() + 8.0 + 0.03 * len(inputs)           # imports + warm-up of every worker happen before this instant (late_s says if not)
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", mode, path, str(w), str(threads), repr(t_start), *map(str, extra)],
                              cwd=os.path.dirname(os.path.abspath(__file__)), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for w, path in enumerate(inputs)]
    out = []
    for pr in procs:
        so, _ = pr.communicate(timeout=600)
        lines = [ln for ln in so.splitlines() if ln.startswith("{")]
        if pr.returncode == 0 and lines:
            out.append(json.loads(lines[-1]))
    return out


def cpu_baseline(scans):
    """The same workload on the host cores with the oracle port (BEV restatement in C, Radon restatement in C + OpenMP,
    fast_corr restatement on torch CPU) on a bounded sample, on the cores the container may really use (cpu_budget: the cgroup quota, not `nproc`);
    `value` / `cores` are those of the fastest setting, every setting tried is in `at_threads`."""
    from oracle import pyoracle as O
    from oracle import corr_oracle as K
    from concurrent.futures import ThreadPoolExecutor
    nproc, usable, quota_src = cpu_budget()
    cpu_model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    ang = np.linspace(0, 2 * np.pi, 120).astype(np.float32)

    def run(cores, soas):
        # the correlation leg is thousands of 120-point FFTs: torch's intra-op threads buy nothing there (0.2 ms per pair on one thread) and have a
        # pathological mode on some hosts (16 ms per pair at 2 threads); one torch thread, the other legs on `cores` threads
        torch.set_num_threads(1)
        os.environ["OMP_NUM_THREADS"] = str(cores)
        O.set_omp_threads(cores)
        w = O.bev_cart(soas[0], 1, 1, 120, 120, 1).reshape(-1, 3)[:, 2].reshape(1, 120, 120)
        ws = O.radon_parallel(w, ang, 120, 1.0)
        O.set_omp_threads(1)
        wt = K.tiring_from_sinogram(ws)
        K.fast_corr(wt, wt)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:      # the reference rasteriser is single-threaded per scan; ctypes drops the GIL
            imgs = np.stack(list(ex.map(lambda s: O.bev_cart(s, 1, 1, 120, 120, 1).reshape(-1, 3)[:, 2].reshape(120, 120), soas)))
        t1 = time.perf_counter()
        O.set_omp_threads(cores)        # torch re-applies ITS thread count to the calling thread's OpenMP state inside every op
        sino = O.radon_parallel(imgs, ang, 120, 1.0)
        t2 = time.perf_counter()
        O.set_omp_threads(1)            # ... and torch's own parallel regions must not inherit the C legs' count (one OpenMP runtime serves both)
        tir = [K.tiring_from_sinogram(s[None]) for s in sino]
        for i in range(len(tir)):
            K.fast_corr(tir[i], tir[(i + 1) % len(tir)])
        t3 = time.perf_counter()
        n = len(soas)
        return {"value": n / (t3 - t0), "cores": cores, "sample_scans": n,
                "ms_per_pair": {"bev": 1e3 * (t1 - t0) / n, "radon": 1e3 * (t2 - t1) / n, "fft_corr": 1e3 * (t3 - t2) / n}}
    # SURVEY.md 8(d) asks for the box's cores.  What a process may USE is the container's CPU quota, not what `nproc` prints: the GPU boxes report 256
    # hardware threads and grant 16 cores of CPU time -- a 256-thread run is throttled to 2.7 pairs/s, 32 pinned 8-thread workers together do no
    # better than one 16-thread process (measured, round 5).  The baseline therefore runs at the usable core count: one process up to 16 threads,
    # and, where more cores are really available, usable / 8 pinned worker processes x 8 threads released together (oracle/cpu_worker.py).
    all_soas = [synth.to_soa(s) for s in scans]
    few = run(min(usable, 16), all_soas)
    out = {"value": few["value"], "unit": "pairs/s", "cores": few["cores"], "nproc": nproc, "usable_cores": usable, "cpu_quota": quota_src,
           "cpu_model": cpu_model, "kind": "port",
           "sample": f"{len(scans)} scans x 120k pts: C BEV restatement (1 thread per scan, scans over {few['cores']} threads), "
                     f"C Radon restatement (OpenMP over images), torch-CPU fast_corr ({few['cores']} threads)",
           "ms_per_pair": few["ms_per_pair"], "at_threads": {str(few["cores"]): few}}
    if usable > 16:
        import tempfile
        nw, per = max(1, usable // 8), 16
        with tempfile.TemporaryDirectory() as td:
            paths = []
            for w in range(nw):
                pth = os.path.join(td, f"w{w}.npz")
                np.savez(pth, **{f"s{i:03d}": all_soas[(w * per + i) % len(all_soas)] for i in range(per)})
                paths.append(pth)
            res = _cpu_workers("ring", paths, 8)
        if res:
            rate = sum(r["units"] for r in res) / max(r["seconds"] for r in res)
            out["at_threads"][str(usable)] = {"value": rate, "cores": usable, "workers": len(res), "threads_per_worker": 8, "sample_scans": sum(r["units"] for r in res),
                                              "slowest_worker_s": max(r["seconds"] for r in res), "fastest_worker_s": min(r["seconds"] for r in res)}
            out["value_at_usable_cores"] = rate
            if rate > out["value"]:
                out.update({"value": rate, "cores": usable,
                            "sample": f"{len(res)} worker processes x 8 threads on disjoint cores, {per} scans x 120k pts each (C BEV / Radon restatements, "
                                      "torch-CPU fast_corr), released together; rate = all scans / slowest worker"})
    soas = all_soas
    out["gicp"] = cpu_gicp_baseline(usable)
    # the reference's own CPU rasterisers, compiled from its sources (kind "reference"): one thread, and one scan per thread on every core
    def ref_rate(fn, label):
        sample = soas[:128]
        t0 = time.perf_counter()
        for s in sample:
            fn(s)
        out[f"reference_{label}_bev_scans_per_s"] = len(sample) / (time.perf_counter() - t0)
        many = (soas * ((4 * usable + len(soas) - 1) // len(soas)))[:max(4 * usable, len(sample))]
        with ThreadPoolExecutor(usable) as ex:
            t0 = time.perf_counter()
            list(ex.map(fn, many))
            out[f"reference_{label}_bev_scans_per_s_all_cores"] = len(many) / (time.perf_counter() - t0)
    if O.ref_polar() is not None:
        ref_rate(lambda s: O.ref_bev_polar(s, 1, 1, 40, 120, 20, 1), "polar")
    if O.ref_lib("cart") is not None:
        # one thread only: the host build of the reference's CUDA rasteriser goes through a cudaMalloc / cudaMemcpy stand-in with process-wide
        # state (oracle/ref_cuda_host), which serialises and thrashes under threads (39 scans/s on 256 threads against 785 on one)
        sample = soas[:128]
        t0 = time.perf_counter()
        for s in sample:
            O.ref_bev_cart(s, 1, 1, 120, 120, 1)
        out["reference_cart_bev_scans_per_s"] = len(sample) / (time.perf_counter() - t0)
    out["reference_bev_threads"] = usable
    return out


def write_slot(c, parity, n_launch, two_sets):
    """database slot launch c of a step writes: slot c of the step's set (two sets written alternately, N = 1) or slot c of the only set"""
    return parity * n_launch + c if two_sets else c


def read_slot(c, parity, n_launch, depth, two_sets):
    """database slot launch c reads = the entries built `depth` launches earlier.  Two sets: slot c - depth of this step's set or, for the step's
    first `depth` launches, the tail of the OTHER set (written at the end of the previous step).  One set: slot c - depth, or one of `depth`
    extra slots behind the set that receive a copy of the previous step's tail when a step starts."""
    if two_sets:
        return parity * n_launch + c - depth if c >= depth else (1 - parity) * n_launch + n_launch - depth + c
    return c - depth if c >= depth else n_launch + c


def verify_timed_outputs(n, whole, norm_group, group_first, out_dist, out_ang, cand_idx, db_launch_of, seed=0):
    """Checker leg, run AFTER the timed region (the oracle is the checker, never the thing measured): proves that the timed kernels did the
    work.  For n random (launch, scan) picks out of the LAST fused descriptor launch of the timed loop:
      * the normalised sinogram the timed launch left in HBM is bit-identical to a fresh launch of the same kernel on that scan, whose BEV
        image and raw sinogram are bit-identical to the oracle's (bev_oracle.c / radon_oracle.c) and whose normalisation is within 2e-5 of
        the reference's (util.py:197 restated);
      * the (distance, angle) the timed correlation kernel wrote for that scan and its candidate equals fast_corr (util.py:362-374
        restated) of the two ORACLE descriptors: angle exact, distance within 1e-5."""
    from oracle import pyoracle as O
    from oracle import corr_oracle as K
    rng = np.random.default_rng(seed)
    CHl, B = out_dist.shape
    G = norm_group.shape[0] // B
    ang = np.linspace(0, 2 * np.pi, 120).astype(np.float32)
    offs1 = torch.arange(2, dtype=torch.int64, device=whole.device) * N_POINTS

    def oracle_descriptor(c, i):
        soa = whole[c, i].reshape(-1).cpu().numpy()
        img = O.bev_cart(soa, 1, 1, 120, 120, 1).reshape(-1, 3)[:, 2].reshape(1, 120, 120)
        sino = O.radon_parallel(img, ang, 120, 1.0)
        return img, sino, K.tiring_from_sinogram(sino)
    res = {"checked": 0, "bev_mismatches": 0, "sinogram_mismatches": 0, "timed_vs_fresh_launch_mismatches": 0, "angle_mismatches": 0,
           "max_err_norm": 0.0, "max_err_dist": 0.0}
    for _ in range(n):
        g, i = int(rng.integers(0, G)), int(rng.integers(0, B))
        c = group_first + g
        img_o, sino_o, tir_o = oracle_descriptor(c, i)
        img, sino, norm = ring.ring_descriptors_fused(whole[c, i].reshape(-1), offs1, want_bev=True, raw=True, normalized=True)
        timed = norm_group[g * B + i]
        res["timed_vs_fresh_launch_mismatches"] += int(not torch.equal(timed, norm[0]))
        res["bev_mismatches"] += int(not np.array_equal(img[0].cpu().numpy(), img_o[0]))
        res["sinogram_mismatches"] += int(not np.array_equal(sino[0].cpu().numpy(), sino_o[0]))
        res["max_err_norm"] = max(res["max_err_norm"], float(np.abs(timed.cpu().numpy() - K.ring_normalize(sino_o)[0].numpy()).max()))
        cl, ci = db_launch_of(c), int(cand_idx[c, i])
        _, _, tir_c = oracle_descriptor(cl, ci)
        wd, wa, _ = K.fast_corr(tir_o, tir_c)
        res["max_err_dist"] = max(res["max_err_dist"], abs(float(out_dist[c, i]) - float(wd)))
        res["angle_mismatches"] += int(int(out_ang[c, i]) != int(wa))
        res["checked"] += 1
    res["max_err"] = max(res["max_err_norm"], res["max_err_dist"])
    res["ok"] = bool(res["bev_mismatches"] == 0 and res["sinogram_mismatches"] == 0 and res["timed_vs_fresh_launch_mismatches"] == 0 and
                     res["angle_mismatches"] == 0 and res["max_err_norm"] < 2e-5 and res["max_err_dist"] < 1e-5)
    res["what"] = ("random scans of the last fused launch of the timed loop: timed normalised sinogram == fresh launch (bits), BEV + raw sinogram "
                   "== oracle (bits), normalisation < 2e-5, timed (dist, angle) vs fast_corr of the oracle descriptors (1e-5, exact)")
    return res


def _gicp_pairs(n_pairs, rank, seed0=500):
    from scipy.spatial.transform import Rotation as Rot
    rng = np.random.default_rng(2000 + rank)
    base = [synth.lidar_scan(seed0 + 10 * rank + s, N_POINTS, metric=True) for s in range(2)]
    srcs, tgts = [], []
    for i in range(n_pairs):
        p = base[i % 2].astype(np.float64)
        ang = np.deg2rad(rng.uniform(0, 5))
        axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
        R = Rot.from_rotvec(ang * axis).as_matrix()
        t = rng.normal(size=3); t *= rng.uniform(0, 1) / np.linalg.norm(t)
        srcs.append((p + rng.normal(0, 0.02, p.shape)).astype(np.float32))
        tgts.append((p @ R.T + t + rng.normal(0, 0.02, p.shape)).astype(np.float32))
    return srcs, tgts


def cpu_gicp_baseline(nproc, iters=20):
    """fast_gicp restatement (kd-tree + OpenMP, oracle/gicp_oracle.cpp) on ONE 120k x 120k pair of the GICP leg's shape, at the reference's own
    thread settings (4: main_RING.py:93, 8: global_manager.cpp:2438) and at `nproc`."""
    from oracle import pyoracle as O
    srcs, tgts = _gicp_pairs(1, 0)
    out = {"sample": f"1 pair x 120k pts, {iters} forced iterations, k=15, max_corr 5.0 (restated fast_gicp, kd-tree + OpenMP)", "by_threads": {}}
    for th in sorted({4, 8, nproc}):
        g = O.Gicp(k=15, max_corr=5.0, threads=th)
        t0 = time.perf_counter()
        g.set_source(srcs[0]); g.set_target(tgts[0])        # builds both kd-trees
        t1 = time.perf_counter()
        g.covariances(0); g.covariances(1)
        t2 = time.perf_counter()
        _, _, its, trials = g.align(np.eye(4), force_iters=iters)
        t3 = time.perf_counter()
        out["by_threads"][str(th)] = {"iters_per_s": its / (t3 - t2), "iterations": its, "lm_trials": trials, "nn_passes": g.nn_passes, "align_s": t3 - t2,
                                      "covariance_clouds_per_s": 2 / (t2 - t1), "kdtree_build_s": t1 - t0, "cores": th,
                                      "pairs_per_s_incl_covariances_and_trees": 1.0 / (t3 - t0)}
    if nproc > 8:
        # every usable core: nproc / 8 independent pairs, one 8-thread registration each (the Mapping node's own setting, global_manager.cpp:2438);
        # `nproc` here is the usable core count (cpu_budget)
        import tempfile
        nw = max(1, nproc // 8)
        with tempfile.TemporaryDirectory() as td:
            pth = os.path.join(td, "pair.npz")
            np.savez(pth, src=srcs[0], tgt=tgts[0])
            res = _cpu_workers("gicp", [pth] * nw, 8, extra=(iters,))
        if res:
            slow = max(r["seconds"] for r in res)
            slow_align = max(r["align_s"] for r in res)
            out["all_cores"] = {"workers": len(res), "threads_per_worker": 8, "cores": nproc, "pairs_per_s_incl_covariances_and_trees": len(res) / slow,
                                "iters_per_s": sum(r["iterations"] for r in res) / slow_align, "slowest_worker_s": slow, "slowest_align_s": slow_align}
    best = max(out["by_threads"].values(), key=lambda r: r["iters_per_s"])
    out.update({k: best[k] for k in ("iters_per_s", "iterations", "lm_trials", "nn_passes", "align_s", "covariance_clouds_per_s", "kdtree_build_s", "cores")})
    out["note"] = "top-level figures = the fastest single registration among the thread counts tried (by_threads holds all of them); all_cores = " \
                  "nproc / 8 registrations of 8 threads each at the same time"
    return out


# ------------------------------------------------------------------------------------------------------- GICP leg
def gicp_leg(device_index, rank, n_pairs, iters):
    """BASELINE configs[2] shape: submap pairs x 120k points.  Three protocols on the same pairs:
      forced : `iters` outer iterations with the convergence test disabled (most of them are warm-started
               near-zero-motion passes once the pose has settled);
      cold   : the first 5 iterations from the identity guess with no warm start (every NN pass searches from scratch);
      natural: run to convergence (is_converged), pairs/s and iterations actually used.
    Covariances are timed separately (they are cached per submap)."""
    from mr_slam_amd import gicp
    srcs, tgts = _gicp_pairs(n_pairs, rank)
    w = gicp.GicpBatch(1, device_index)                # load the code objects before anything is timed
    w.set_sources([srcs[0][:4000]]); w.set_targets([tgts[0][:4000]]); w.align(); del w
    b = gicp.GicpBatch(n_pairs, device_index)
    b.set_params(k_correspondences=15, max_correspondence_distance=5.0)
    b.set_sources(srcs); b.set_targets(tgts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b.compute_covariances(0); b.compute_covariances(1)
    torch.cuda.synchronize()
    t_cov_first = time.perf_counter() - t0           # includes the first allocation of the 1.8 GB neighbour-index scratch (15-25 ms on some boxes)
    b.set_sources(srcs); b.set_targets(tgts)         # the same clouds handed over again: buffers and scratch are kept (a registration object's steady state)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b.compute_covariances(0); b.compute_covariances(1)
    torch.cuda.synchronize()
    t_cov = time.perf_counter() - t0

    def timed(**prm):
        b.set_params(**prm)
        torch.cuda.synchronize()
        t = time.perf_counter()
        T, conv, its = b.align()
        torch.cuda.synchronize()
        return time.perf_counter() - t, conv, its, b.nn_passes

    # cold: set_sources resets the warm-start seeds; nothing has been aligned yet
    t_cold, _, its_c, nn_c = timed(force_iterations=5)
    s_cold = b.searched_fraction
    t_forced, _, its_f, nn_f = timed(force_iterations=iters)          # seeds now warm from the cold run (like a re-check)
    s_forced = b.searched_fraction
    assert (its_f == iters).all() and (its_c == 5).all()
    # the headline protocol: `iters` forced iterations from the identity guess with NOTHING carried over (seeds, certificates and the neighbour
    # cache reset by handing the clouds over again; covariances recomputed outside the clock)
    b.set_sources(srcs); b.set_targets(tgts)
    b.compute_covariances(0); b.compute_covariances(1)
    t_cold_full, _, its_cf, nn_cf = timed(force_iterations=iters)
    s_cold_full = b.searched_fraction
    assert (its_cf == iters).all()
    b.set_sources(srcs)                                               # reset seeds (covariances recomputed lazily: not timed)
    b.compute_covariances(0)
    b.set_params(force_iterations=0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    T_nat, conv, its_n = b.align()
    torch.cuda.synchronize()
    t_nat = time.perf_counter() - t
    nn_n, s_nat = b.nn_passes, b.searched_fraction
    # the node's shape (main_RING.py:81-104, global_manager.cpp:2016-2021): a NEW scan against several STORED candidates.  The clouds live once in
    # a store batch (Morton order, boxes, hierarchy, covariances: paid when a scan arrives, kept while it is a stored submap) and are copied
    # into the pairs; fast_gicp itself recomputes both clouds' covariances for every pair.
    n_cand = min(8, n_pairs)
    n_new = max(1, n_pairs // n_cand)
    n_sh = n_new * n_cand
    stored = tgts[:2 * n_cand:2] if n_pairs >= 2 * n_cand else tgts[:n_cand]      # candidates: stored views of the scene the new scans see
    news = [srcs[(2 * i) % len(srcs)] for i in range(n_new)]
    store = gicp.GicpBatch(len(stored) + n_new, device_index)
    store.set_params(k_correspondences=15, max_correspondence_distance=5.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    store.set_targets(stored + news)
    torch.cuda.synchronize()
    t_store_ingest = time.perf_counter() - t0
    t0 = time.perf_counter()
    store.compute_covariances(1)
    torch.cuda.synchronize()
    t_store_cov = time.perf_counter() - t0
    bs = gicp.GicpBatch(n_sh, device_index)
    bs.set_params(k_correspondences=15, max_correspondence_distance=5.0)
    src_ids = np.repeat(len(stored) + np.arange(n_new), n_cand)
    tgt_ids = np.tile(np.arange(n_cand) % len(stored), n_new)
    bs.set_sources_from(store, src_ids); bs.set_targets_from(store, tgt_ids)        # first call: buffers are allocated
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bs.set_sources_from(store, src_ids); bs.set_targets_from(store, tgt_ids)
    torch.cuda.synchronize()
    t_from = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, conv_s, its_s = bs.align()
    torch.cuda.synchronize()
    t_sh = time.perf_counter() - t0
    per_cloud = (t_store_ingest + t_store_cov) / (len(stored) + n_new)                 # what one arriving scan costs: sort + boxes + hierarchy + covariances
    shared = {"new_scans": n_new, "candidates_per_scan": n_cand, "pairs": n_sh, "align_s": t_sh, "copy_into_pairs_s": t_from,
              "store_ingest_s_per_cloud": t_store_ingest / (len(stored) + n_new), "store_covariance_s_per_cloud": t_store_cov / (len(stored) + n_new),
              "converged": int(conv_s.sum()), "mean_iterations": float(np.mean(its_s)),
              "pairs_per_s_incl_covariances": n_sh / (t_sh + t_from + n_new * per_cloud),
              "note": "every new scan is sorted and gets its covariances ONCE (it becomes a stored submap afterwards), then meets its candidates through "
                      "mrs_gicp_batch_set_clouds_from; pairs_per_s_incl_covariances above this block pays both clouds of every pair"}
    del bs, store
    # every kernel of an outer iteration alone between HIP events, at the converged poses (numeric rooflines)
    kms, kcnt = b.profile(T_nat, reps=3)
    n_src, n_corr = kcnt["source_points"], kcnt["correspondences"]
    lin_bytes = 20 * n_src + 64 * n_corr          # every source point: 16 B point + 4 B index; every correspondence: + 24 B normal (the PLANE covariance is
                                                  # I - 0.999 n n^T, rebuilt in the kernel: round 5) + gathered 16 B point + 24 B normal
    knn_bytes = n_src * (16 + 4 * 15)             # the selection's compulsory traffic (point in, 15 indices out): it is VALU-bound, not HBM-bound
    cov_bytes = n_src * (4 * 15 + 16 + 24)        # indices + own point in, 24 B normal out (the 15 gathered points are L2 hits)

    def hbm(bytes_, ms):
        gbs = bytes_ / (ms * 1e-3) / 1e9
        return {"bytes": bytes_, "ms": ms, "achieved": gbs, "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS}
    roof = {
        "k_linearize": dict(hbm(lin_bytes, kms["linearize"]), bound="hbm", note="84 B per correspondence + 20 B per unmatched source point, gathered normals; "
                            "launched alone at the converged poses, 256 pairs"),
        "k_linearize_error_only": dict(hbm(lin_bytes, kms["linearize_error_only"]), bound="hbm", note="an LM trial: the same bytes, one sum instead of 28"),
        "k_nn_scan (round-3 search, every point, warm)": {"bound": "valu", "ms": kms["search_round3_all"], "queries_per_s": n_src / kms["search_round3_all"] * 1e3,
                                                          "note": "VALU-pipe busy fraction from the PMC pass of the same shape: profiles/*_pmc.json"},
        "k_nn_scan_g (round-4 search, every point, warm)": {"bound": "latency", "ms": kms["search_round4_all"], "queries_per_s": n_src / kms["search_round4_all"] * 1e3},
        "k_nn_certify (unchanged pose)": dict(hbm(n_src * 44, kms["certify"]), bound="hbm", note="16 B point + 4 B seed + 4 B bound in, 4 B index + 4 B bound out, "
                                              "16 B neighbour (gathered: keeping a copy beside the source point was built and bought nothing, DESIGN.md 4): 44 B per source point"),
        "certify + work-list search after a 1 mm step": {"ms": kms["certify_plus_worklist_1mm"], "worklist_queries": kcnt["worklist_queries_1mm"],
                                                         "share_of_points_searched": kcnt["worklist_queries_1mm"] / max(n_src, 1)},
        "k_knn_cov (selection)": dict(hbm(knn_bytes, kms["knn_select"]), bound="valu", clouds_per_s=n_pairs / kms["knn_select"] * 1e3),
        "k_cov_from_knn": dict(hbm(cov_bytes, kms["cov_from_knn"]), bound="hbm+fp64", clouds_per_s=n_pairs / kms["cov_from_knn"] * 1e3),
    }
    return {"pairs": n_pairs, "iterations": iters, "points": N_POINTS,
            "iters_per_s": n_pairs * iters / t_cold_full, "align_s": t_cold_full, "nn_passes": nn_cf,
            "nn_pass_ms": 1e3 * t_cold_full / (n_pairs * max(nn_cf, 1)), "searched_fraction": s_cold_full,
            "protocol": f"{iters} forced outer iterations from the identity guess, cold start (no seeds / certificates / cached neighbours carried over)",
            "warm": {"iters_per_s": n_pairs * iters / t_forced, "align_s": t_forced, "nn_passes": nn_f, "searched_fraction": s_forced,
                     "note": f"the same {iters} forced iterations started with the nearest-neighbour seeds of a previous 5-iteration run of the same pairs "
                             "(a re-check of a known pair): the figure earlier rounds reported as iters_per_s"},
            "cold": {"iterations": 5, "iters_per_s": n_pairs * 5 / t_cold, "align_s": t_cold, "nn_passes": nn_c, "searched_fraction": s_cold,
                     "note": "first 5 outer iterations from the identity guess, no warm start"},
            "natural": {"pairs_per_s": n_pairs / t_nat, "align_s": t_nat, "converged": int(conv.sum()),
                        "mean_iterations": float(np.mean(its_n)), "max_iterations": int(np.max(its_n)), "nn_passes": nn_n,
                        "iters_per_s": float(np.sum(its_n)) / t_nat, "searched_fraction": s_nat},
            "nn_search": "exact: first pass and pairs that moved > 2 cm by brute force over the Morton-ordered cloud (a wave walks 1024-point tiles / 16-point "
                         "minis, scalar candidate loads); other pairs certify last pass's neighbours by the triangle inequality (k_nn_certify) and search only "
                         "the uncertified queries on octree-cell leaves with per-query culling (k_nn_scan_g); one NN pass per outer iteration, LM trials "
                         "score the cached correspondences (upstream compute_error)",
            "pairs_per_s_incl_covariances": n_pairs / (t_nat + t_cov), "shared_submaps": shared,
            "covariance_s": t_cov, "covariance_steady_s": t_cov, "covariance_first_call_s": t_cov_first,
            "pairs_per_s_incl_covariances_first_call": n_pairs / (t_nat + t_cov_first), "covariance_clouds_per_s": 2 * n_pairs / t_cov, "k": 15,
            "max_correspondence_distance": 5.0, "kernel_ms": kms, "kernel_counts": kcnt, "roofline": roof}


# ---------------------------------------------------------------------------------------------------- sweep legs
def sweep_legs(device, spec_pool, n_db=10_000):
    """BASELINE configs[3] shape on one GPU: databases of 10 000 descriptors, 1 and 4 queries per launch.
    Bandwidth = database bytes streamed once per launch / launch time (the queries of a launch share the entry through
    L2, so the rate is the same number for 1 and 4 queries only if the kernel is bandwidth bound).
    One query (the node's loop, main_RING.py:133): the database in its resident format (DMA-tiled entries, what mrs_loopdb keeps) swept by the
    LDS-DMA kernel; `*_q1_row_layout` = the same query over [61][120] row-layout entries (mrs_ring_corr_fft_sweep[_mc])."""
    from mr_slam_amd import disco, node
    out = {}
    idx = torch.arange(n_db, device=device) % spec_pool.shape[0]
    db = spec_pool[idx].contiguous()                                   # RING: [n_db][61][120] complex64 = 58 560 B each

    def entry(nq, ms, bytes_per_entry, **kw):
        return dict({"pairs_per_s": nq * n_db / ms * 1e3, "ms": ms, "db_entries": n_db, "bytes_per_entry": bytes_per_entry,
                     "db_gbs": n_db * bytes_per_entry / ms / 1e6, "hbm_frac": n_db * bytes_per_entry / ms / 1e6 / HBM_PEAK_GBS}, **kw)
    TILED = 58624                                                      # MRS_RING_TILED_ENTRY_BYTES: what the tiled sweep streams per plane
    # Protocols.  One query (HBM-bound, what a callback issues is ONE launch): bursts of 3 + 10 launches, as in rounds 3-5 -- a longer burst of this
    # kernel measures LOWER (30 + 20 launches: 85 instead of 100 M pairs/s on the tiled entries; the row-layout kernel 86 either way).  Several
    # queries (VALU-bound): the first ~40 launches after host-side gaps run below the steady rate (115 -> 128 -> 137 -> 142 M pairs/s over
    # consecutive groups of 13 launches at 4 queries, tools/quick_sweep_order.py: the clocks come up), so those legs warm up for 30 launches
    W1, R1 = 3, 10                                                     # one query
    WQ, RQ = 30, 20                                                    # several queries, RING
    W6, R6 = 1, 3                                                      # RING++, one query
    W6Q, R6Q = 6, 8                                                    # RING++, several queries
    tiled = ring.spec_to_tiled(db)
    q1 = spec_pool[:1].contiguous()
    out["ring_q1"] = entry(1, ev_ms(lambda: ring.corr_sweep_fft_tiled(q1, tiled), reps=R1, warm=W1), TILED, layout="dma-tiled (mrs_loopdb)")
    out["ring_q1_row_layout"] = entry(1, ev_ms(lambda: ring.corr_sweep_fft(q1, db), reps=R1, warm=W1), 58560)
    # several queries per sweep (round 6): the one-query LDS-DMA pipeline per (query, entry), the queries' workgroups grouped per XCD so that
    # the entry leaves HBM once (profiles/r06_notes.md); hbm_frac = database bytes streamed ONCE per launch / time
    for nq in (4, 8):
        q = spec_pool[:nq].contiguous()
        out[f"ring_q{nq}"] = entry(nq, ev_ms(lambda: ring.corr_sweep_fft_tiled_q(q, tiled), reps=RQ, warm=WQ), TILED, layout="dma-tiled (mrs_loopdb_query_multi)")
    q = spec_pool[:4].contiguous()
    out["ring_q4_row_layout"] = entry(4, ev_ms(lambda: ring.corr_sweep_fft(q, db), reps=RQ, warm=WQ), 58560)
    del tiled
    db6 = torch.stack([db.roll(k, 0) for k in range(6)], 1).contiguous()   # RING++: [n_db][6][61][120] = 351 360 B each
    tiled6 = ring.spec_to_tiled(db6)
    q1 = db6[:1].contiguous()
    out["ringpp_q1"] = entry(1, ev_ms(lambda: ring.corr_sweep_fft_tiled(q1, tiled6), reps=R6, warm=W6), 6 * TILED, layout="dma-tiled (mrs_loopdb)")
    q = db6[:4].contiguous()
    out["ringpp_q4"] = entry(4, ev_ms(lambda: ring.corr_sweep_fft_tiled_q(q, tiled6), reps=R6Q, warm=W6Q), 6 * TILED, layout="dma-tiled (mrs_loopdb_query_multi)")
    del tiled6
    out["ringpp_q1_row_layout"] = entry(1, ev_ms(lambda: ring.corr_sweep_fft(q1, db6), reps=R6, warm=W6), 351360)
    out["ringpp_q4_row_layout"] = entry(4, ev_ms(lambda: ring.corr_sweep_fft(q, db6), reps=R6Q, warm=W6Q), 351360)
    del db6
    # DiSCO (disco_ros/main.py:284-291): nearest 1024-d signature over the database, then ONE phase correlation
    g = torch.Generator(device=device).manual_seed(3)
    sig_db = torch.rand((n_db, 1024), generator=g, device=device)
    spec_db = torch.view_as_complex(torch.randn((n_db, 1, 40, 120, 2), generator=g, device=device))
    ddb = node.DiscoDatabase(capacity=n_db)
    for i in range(n_db):
        ddb.append(sig_db[i], spec_db[i])
    qs, qf = (sig_db[0] + 0.01).contiguous(), spec_db[0].contiguous()
    ddb.query(qs, qf)
    t0 = time.perf_counter()
    for _ in range(20):
        ddb.query(qs, qf)
    ms = 1e3 * (time.perf_counter() - t0) / 20
    out["disco_q1"] = {"queries_per_s": 1e3 / ms, "pairs_per_s": n_db / ms * 1e3, "ms": ms, "db_entries": n_db, "bytes_per_entry": 4096,
                       "db_gbs": n_db * 4096 / ms / 1e6,
                       "note": "mrs_loopdb_query_disco, host wall time of the blocking call (device arguments): nearest signature + phase_corr of "
                               "the winner in two launches"}
    for nq in (1, 4):
        qs = sig_db[:nq].contiguous() + 0.01

        def disco_query():
            i, _ = disco.signature_search(qs, sig_db)
            return disco.phase_corr(spec_db[:nq], spec_db[i.long()])
        ms = ev_ms(disco_query)
        out["disco_q1_multi_kernel" if nq == 1 else "disco_q4"] = {"queries_per_s": nq / ms * 1e3, "pairs_per_s": nq * n_db / ms * 1e3, "ms": ms, "db_entries": n_db,
                                            "bytes_per_entry": 4096, "db_gbs": n_db * 4096 / ms / 1e6,
                                            "note": "signature search over the whole database + rocFFT phase_corr with the best entry (batched form)"}
    return out


def node_shape_leg(device, spec_pool, n_db=10_000, n_loop=1000):
    """The LoopDetection node's own shape (main_RING.py:126-140, 284-288): one descriptor appended per callback, every new scan scored against
    every stored entry.  (i) the reference's loop as written, through the drop-in (`fast_corr` per entry on host tensors), (ii) its twin:
    mrs_loopdb append + ONE query, host wall time of the Python call included.  `spec_pool`: half spectra [>= 256,61,120] (device)."""
    from mr_slam_amd import node
    n_pool = min(256, spec_pool.shape[0])
    half = spec_pool[:n_pool]
    pool = torch.cat([half, half[:, 1:60].flip(1).conj()], 1).contiguous()   # TIRING as generate_RING returns it (util.py:198): complex64 [.,120,120], rows 61.. Hermitian
    host = [pool[i:i + 1].cpu() for i in range(n_pool)]
    TIRING = [host[i % n_pool] for i in range(n_db)]
    cur = host[3 % n_pool]
    out = {"db_entries": n_db}
    # (i) the unchanged loop.  Through the drop-in, fast_corr keeps a device twin of every HOST tensor it is handed (ring.DeviceMirror; generate_RING
    # seeds it, so a node's own descriptors never upload): `first_visit` pays one 115 KB upload per tensor the mirror has not seen (descriptors
    # that came from elsewhere), the steady state is what every later callback sees -- the node's lists persist and are swept at every callback
    ring.device_mirror().clear()
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        hits = 0
        for idx in range(n_loop):
            dist, angle = ring.fast_corr(cur, TIRING[idx])
            if dist < 0.48:
                hits += 1
        rates.append(n_loop / (time.perf_counter() - t0))
    out["reference_loop_through_dropin"] = {"pairs_per_s": max(rates[1:]), "ms_per_pair": 1e3 / max(rates[1:]), "entries_timed": n_loop,
                                            "first_visit_pairs_per_s": rates[0], "distinct_host_tensors": min(n_pool, n_loop),
                                            "what": "for idx in range(len(candidates)): fast_corr(TIRING_current, TIRING_candidates[idx]) on host tensors, "
                                                    "unchanged; device twins of the host tensors (no upload after a tensor's first visit)"}
    ring.device_mirror().clear()
    # (ii) the twin
    db = node.LoopDatabase("ring", capacity=1024)
    t0 = time.perf_counter()
    for i in range(n_db):
        db.append(TIRING[i])
    torch.cuda.synchronize()
    t_app = time.perf_counter() - t0
    out["append"] = {"entries_per_s": n_db / t_app, "us_per_entry": 1e6 * t_app / n_db, "what": "TIRING<k>.append(pc_TIRING) from the host tensor, growth included"}
    cur_dev = pool[3 % n_pool:3 % n_pool + 1].contiguous()
    cur_spec = cur_dev[:, :61].contiguous()
    for name, qarg in (("query_host_tiring", cur), ("query_device_tiring", cur_dev), ("query_device_half_spectrum", cur_spec)):
        db.query(qarg, 0.48)
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            idxs, dists, angles = db.query(qarg, 0.48)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        out[name] = {"pairs_per_s": n_db / t, "ms": 1e3 * t, "ms_min": 1e3 * min(ts), "entries_under_threshold": int(len(idxs))}
    out["twin_pairs_per_s"] = out["query_host_tiring"]["pairs_per_s"]
    out["speedup_vs_reference_loop"] = out["twin_pairs_per_s"] / out["reference_loop_through_dropin"]["pairs_per_s"]
    _, _, _, alld, alla = db.query(cur, 0.48, want_all=True)
    d_ref = np.array([float(ring.fast_corr(cur, TIRING[i])[0]) for i in range(64)], np.float32)
    a_ref = np.array([int(ring.fast_corr(cur, TIRING[i])[1]) for i in range(64)])
    out["matches_pairwise"] = bool(np.array_equal(alla[:64], a_ref) and np.abs(alld[:64] - d_ref).max() < 1e-5)
    return out


def build_legs(device, chunks):
    """Descriptor GENERATION of the other two detectors and the ingest-inclusive RING rate, same synthetic scans:
      disco_build  : 1024 scans -> polar BEV 40 x 120 x 20 (disco_ros/main.py:112-125) -> DiSCO.forward without the UNet (DiSCO.py:315-334);
      ringpp_build : 64 scans -> exact kNN k = 30 + eigen features (util.py:123-170, 204-219) -> 9-plane feature BEV -> Radon of the 6 feature
                     channels -> |FFT along the detector axis| (util.py:220-250);
      ingest       : 64 raw metric clouds [n, 4] float32 -> voxel_down_sample(0.2) (main_RING.py:257-259, one call per scan like the node) ->
                     load_pc_infer crop / scale (util.py:91-112, one batched launch) -> fused descriptor kernel."""
    from mr_slam_amd import disco, pointfeat, preprocess
    out = {}
    xyz, offs = chunks[0]
    B = offs.numel() - 1
    ms_bev = ev_ms(lambda: bev.polar_bev(xyz, offs, 1, 1, 40, 120, 20), reps=3, warm=1)
    occ = bev.polar_bev(xyz, offs, 1, 1, 40, 120, 20)
    ms_desc = ev_ms(lambda: disco.disco_from_bev(occ), reps=3, warm=1)
    del occ
    out["disco_build"] = {"scans_per_s": B / (ms_bev + ms_desc) * 1e3, "batch": B, "ms": {"polar_bev_40x120x20": ms_bev, "disco_descriptor": ms_desc},
                          "note": "polar BEV scatter (HBM: 12 B/point + 384 KB of cells per scan) + 2-D FFT / signature of the 40 x 120 image"}
    S = min(64, B)
    pts = make_shard.whole[0, :S].permute(0, 2, 1).reshape(S * N_POINTS, 3).contiguous()      # [S * N, 3] AoS, pre-processed like load_pc_infer
    h_offs = np.arange(S + 1, dtype=np.int64) * N_POINTS
    d_offs = torch.from_numpy(h_offs).to(device)
    ms_feat = ev_ms(lambda: pointfeat.point_features(pts, h_offs, 30, want=("planes",)), reps=2, warm=1)
    planes = pointfeat.point_features(pts, h_offs, 30, want=("planes",))["planes"]
    ms_fbev = ev_ms(lambda: bev.feat_bev(planes, d_offs, 9, 1, 1, 120, 120, 1, layout=1), reps=3, warm=1)
    fb = bev.feat_bev(planes, d_offs, 9, 1, 1, 120, 120, 1, layout=1)
    plan = ring.ring_plan(torch.device(device).index or 0)
    imgs = fb.reshape(S * 6, 120, 120)
    ms_radon = ev_ms(lambda: plan.forward(imgs), reps=3, warm=1)
    sino, _ = plan.forward(imgs)
    ms_fft = ev_ms(lambda: ring.forward_row_fft(sino.view(S, 6, 120, 120)), reps=3, warm=1)
    tot = ms_feat + ms_fbev + ms_radon + ms_fft
    out["ringpp_build"] = {"scans_per_s": S / tot * 1e3, "batch": S, "k": 30,
                           "ms": {"knn_k30_features": ms_feat, "feature_bev_9_planes": ms_fbev, "radon_6_channels": ms_radon, "row_fft_magnitude": ms_fft},
                           "knn_points_per_s": S * N_POINTS / ms_feat * 1e3,
                           "bound": "k_knn_cov<30> + k_feat_from_knn: VALU-bound (36 k vector instructions per 64 queries, 5 waves per SIMD) -- exact k = 30 nearest of every point "
                                    "by culled brute force over the Morton-ordered cloud (pass 1: the 30 smallest distances by a v_med3_f32 chain fed from noted candidates; "
                                    "pass 2: the indices within the k-th distance from the minis pass 1 noted; order by rank), eigenvalues in fp64; inputs stay L2 resident "
                                    "(16 B per point in, 36 B per point out), so no HBM or MFMA roof applies"}
    del planes, fb, sino, imgs
    # ingest: raw clouds as the ROS message delivers them (x, y, z, intensity), ~130 k points before down-sampling
    R = min(64, B)
    raws = []
    base_raw = [synth.lidar_scan(900 + s, 130_000, metric=True) for s in range(4)]       # 4 ray-cast scenes, every scan a rotated copy
    for i in range(R):
        p = base_raw[i % 4]
        th = 0.1 * i
        c, sn = np.float32(np.cos(th)), np.float32(np.sin(th))
        q = np.empty((p.shape[0], 4), np.float32)
        q[:, 0] = c * p[:, 0] - sn * p[:, 1]; q[:, 1] = sn * p[:, 0] + c * p[:, 1]; q[:, 2] = p[:, 2]; q[:, 3] = 0.5
        raws.append(torch.from_numpy(q).to(device))

    raw_cat = torch.cat(raws)
    raw_offs = np.concatenate([[0], np.cumsum([r.shape[0] for r in raws])]).astype(np.int64)

    def ingest(batched, timing=None):
        t = [time.perf_counter()]
        if batched:                    # one set of launches for the whole batch (hash grid), centroids stay on the device with their offsets
            cat, doffs = preprocess.voxel_down_sample_batch(raw_cat, raw_offs, 0.2)
            ro = doffs.cpu().numpy()
        else:                          # one call and one host synchronisation per scan, like the node's open3d call
            down = [preprocess.voxel_down_sample(r, 0.2) for r in raws]
            cat = torch.cat(down)
            ro = np.concatenate([[0], np.cumsum([d.shape[0] for d in down])]).astype(np.int64)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        soa, so = preprocess.load_pc_infer_batch(cat, ro)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        ring.ring_descriptors_fused(soa, so, raw=False, normalized=True)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        if timing is not None:
            timing.append((t[1] - t[0], t[2] - t[1], t[3] - t[2], int(ro[-1])))
    res = {}
    for batched in (True, False):
        ingest(batched)
        tm = []
        for _ in range(3):
            ingest(batched, tm)
        v, c, f, kept = [float(np.mean([x[i] for x in tm])) for i in range(4)]
        res[batched] = {"scans_per_s": R / (v + c + f), "ms": {"voxel_down_sample_per_scan": 1e3 * v / R, "crop_scale_batch": 1e3 * c, "fused_descriptors": 1e3 * f},
                        "points_after_voxel_0.2": kept / R}
    out["ingest_from_host"] = host_fed_leg(device, raws, raw_offs)
    out["ingest"] = {"scans_per_s": res[True]["scans_per_s"], "batch": R, "raw_points_per_scan": 130_000,
                     "points_after_voxel_0.2": res[True]["points_after_voxel_0.2"], "ms": res[True]["ms"],
                     "per_scan_calls": res[False],
                     "note": "raw float32 [n, 4] clouds resident in HBM -> open3d-equivalent voxel grid 0.2 m (mrs_voxel_downsample_batch: hash grid, "
                             "fixed-point sums, one host synchronisation per batch; `per_scan_calls`: the sort-based single-scan call with one host "
                             "synchronisation per scan, the ROS callback's shape) -> load_pc_infer -> BEV + Radon + normalise; reported next to "
                             "`value`, never instead of it"}
    return out


def host_fed_leg(device, raws, raw_offs, n_batches=12):
    """What the callbacks really hand over (main_RING.py:251-260): HOST arrays, float32 [n, 4] per scan.  Batches of R scans lie in PINNED host
    memory; a copy stream brings batch b + 1 to one of two device buffers (hipMemcpyAsync) while the compute stream runs voxel_down_sample(0.2)
    -> load_pc_infer -> fused BEV + Radon + normalise on batch b.  Reported next to a pure copy of the same batches on the same box: the wire is
    the bound of a host-fed node, not HBM (every headline leg reads scans that are already resident)."""
    from mr_slam_amd import preprocess
    R = len(raws)
    host = torch.cat(raws).cpu().pin_memory()                       # one batch worth of scans, [sum n, 4] float32, pinned
    nbytes = host.numel() * 4
    offs = raw_offs
    dev = [torch.empty_like(host, device=device) for _ in range(2)]
    copy = torch.cuda.Stream(device=device)
    comp = torch.cuda.current_stream()

    def run(process, nb):
        ready = [torch.cuda.Event() for _ in range(2)]
        freed = [torch.cuda.Event() for _ in range(2)]
        for e in freed:
            e.record(comp)
        last = None

        def issue(b):
            with torch.cuda.stream(copy):
                copy.wait_event(freed[b % 2])                       # the buffer's previous batch has been consumed
                dev[b % 2].copy_(host, non_blocking=True)
                ready[b % 2].record(copy)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        issue(0)
        for b in range(nb):
            if b + 1 < nb:
                issue(b + 1)                                        # in flight while batch b is processed (the batched voxel grid synchronises the host once)
            comp.wait_event(ready[b % 2])
            if process:
                cat, doffs = preprocess.voxel_down_sample_batch(dev[b % 2], offs, 0.2)
                soa, so = preprocess.load_pc_infer_batch(cat, doffs.cpu().numpy())
                last = ring.ring_descriptors_fused(soa, so, raw=False, normalized=True)
            freed[b % 2].record(comp)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, last
    run(True, 2)
    t_copy, _ = run(False, n_batches)
    t_all, last = run(True, n_batches)
    # the last batch's descriptors against the same scans processed from a resident copy (bits)
    cat, doffs = preprocess.voxel_down_sample_batch(torch.cat(raws), offs, 0.2)
    soa, so = preprocess.load_pc_infer_batch(cat, doffs.cpu().numpy())
    want = ring.ring_descriptors_fused(soa, so, raw=False, normalized=True)
    same = bool(torch.equal(last[-1], want[-1]))
    copy_gbs = n_batches * nbytes / t_copy / 1e9
    return {"scans_per_s": n_batches * R / t_all, "batches": n_batches, "scans_per_batch": R, "bytes_per_scan": nbytes / R,
            "host_to_device_gbs": n_batches * nbytes / t_all / 1e9, "pinned_copy_only_gbs": copy_gbs, "pinned_copy_only_scans_per_s": n_batches * R / t_copy,
            "frac_of_pinned_copy": t_copy / t_all, "last_batch_bit_identical_to_resident_input": same,
            "note": "pinned host float32 [n, 4] clouds (130 k points, 2.08 MB per scan) -> hipMemcpyAsync on a copy stream, double-buffered against "
                    "voxel grid 0.2 m + load_pc_infer + fused BEV / Radon / normalise on the compute stream; frac_of_pinned_copy = copy-only time / "
                    "pipeline time for the same batches"}


def pipeline_shard_leg(device, spec_pool, gicp_res):
    """BASELINE configs[4], one GPU's share: RING++ database of 50 000 / 8 = 6 250 entries ([6][61][120] complex64,
    2.2 GB) swept by one query, one elevation-map frame (move + 120k points + fuse + features + ray tracing, row N3), and
    the GICP refinement rate of the configs[2] leg."""
    from mr_slam_amd import elevation
    n_db = 6250
    idx = torch.arange(n_db, device=device) % spec_pool.shape[0]
    db6 = torch.stack([spec_pool[idx].roll(k, 0) for k in range(6)], 1).contiguous()
    q = db6[:1].contiguous()
    ms = ev_ms(lambda: ring.corr_sweep_fft(q, db6), reps=3, warm=1)
    out = {"ringpp_db_entries": n_db, "ringpp_db_bytes": n_db * 351360, "ringpp_sweep_ms": ms,
           "ringpp_sweep_pairs_per_s": n_db / ms * 1e3, "ringpp_sweep_db_gbs": n_db * 351360 / ms / 1e6}
    del db6
    m = elevation.ElevationMap(200, 0.1)
    rng = np.random.default_rng(0)
    n = N_POINTS
    x = rng.uniform(-9, 9, n).astype(np.float32); y = rng.uniform(-9, -1.2, n).astype(np.float32)
    z = (0.1 * np.sin(x) - 0.6 + rng.normal(0, 0.02, n)).astype(np.float32)
    T = np.eye(4, dtype=np.float32); T[2, 3] = 0.9
    rv = np.diag([1e-4, 1e-4, 4e-4]).astype(np.float32)
    cr = rng.integers(0, 256, n); inten = rng.uniform(0, 1, n).astype(np.float32)

    def frame():
        m.move(np.array([0.0, 0.0, 0.9], np.float32))
        r = m.process_points(x, y, z, T, -2.0, 3.0, 0.02, 0.003, 0.01, [0, 0, 1.0], rv, np.eye(3), [0, 0, 1.0], np.zeros((3, 3)))
        m.fuse(r["map_index"], cr, cr, cr, inten, r["z_ts"], r["var"])
        m.mapvar_update(1e-4); m.map_feature(); m.raytracing()
    for _ in range(2):
        frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        frame()
    torch.cuda.synchronize()
    out["elevation_frame_ms"] = 1e3 * (time.perf_counter() - t0) / 5
    out["elevation_frame"] = "200 x 200 map, 120k points, host arrays in and out (the libgpu.so calling convention)"
    if gicp_res:
        out["gicp_iters_per_s"] = gicp_res["iters_per_s"]; out["gicp_pairs_to_convergence_per_s"] = gicp_res["natural"]["pairs_per_s"]
    return out


def dropin_latency_leg(scan):
    """One call at a time through the reference-named modules (host numpy in, host numpy out), as the ROS nodes make them."""
    from mr_slam_amd.compat import gputransform, voxelocc, pygicp
    soa = synth.to_soa(scan)
    n = scan.shape[0]

    def lat(fn, reps=10):
        fn(); fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return 1e3 * (time.perf_counter() - t0) / reps

    def cart():
        t = voxelocc.GPUTransformer(soa, n, 1, 1, 120, 120, 1, 1); t.transform(); return t.retreive()

    def polar():
        t = gputransform.GPUTransformer(soa, n, 1, 1, 40, 120, 20, 1); t.transform(); return t.retreive()
    out = {"voxelocc_120x120x1_ms": lat(cart), "gputransform_40x120x20_ms": lat(polar),
           "generate_RING_ms": lat(lambda: ring.generate_RING(scan)), "points": n}
    _, _, tir = ring.generate_RING(scan)
    out["fast_corr_ms"] = lat(lambda: ring.fast_corr(tir, tir))
    srcs, tgts = _gicp_pairs(1, 0)
    src64, tgt64 = srcs[0].astype(np.float64), tgts[0].astype(np.float64)
    out["pygicp_downsample_0.2_ms"] = lat(lambda: pygicp.downsample(src64, 0.2), reps=5)
    s, t = pygicp.downsample(src64, 0.2), pygicp.downsample(tgt64, 0.2)

    def reg():
        g = pygicp.FastGICP(); g.set_input_target(t); g.set_input_source(s); g.set_max_correspondence_distance(5.0)
        g.align(initial_guess=np.eye(4)); return g.get_fitness_score(1.0)
    out["pygicp_align_downsampled_ms"] = lat(reg, reps=5)
    out["pygicp_points"] = [int(s.shape[0]), int(t.shape[0])]
    return out



# ------------------------------------------------------------------------------------------------ the result line
COMPACT_LIMIT = 4096      # bytes: the driver keeps a bounded tail of stdout; round 5's 21 KB line could not be parsed


def _sig(x, digits=5):
    """floats to `digits` significant digits, NaN / inf to None (strict JSON), recursively"""
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    if isinstance(x, (np.floating, float)):
        x = float(x)
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{digits}g}")
    if isinstance(x, np.integer):
        return int(x)
    if isinstance(x, np.bool_):
        return bool(x)
    return x


def _strict(x):
    """the detail block as strict JSON: numpy scalars to Python, NaN / inf to None, full precision"""
    if isinstance(x, dict):
        return {str(k): _strict(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_strict(v) for v in x]
    if isinstance(x, (np.floating, float)):
        x = float(x)
        return None if (x != x or x in (float("inf"), float("-inf"))) else x
    if isinstance(x, np.integer):
        return int(x)
    if isinstance(x, np.bool_):
        return bool(x)
    return x


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d or d[k] is None:
            return None
        d = d[k]
    return d


def compact_line(d, detail_name):
    """The contract line: the driver's fields + the scalars the targets are stated in, <= COMPACT_LIMIT bytes.  Every block of the full result
    (`d`, written to `detail_name`) that is not a scalar of the contract stays in the detail file."""
    r, g, sw = d.get("roofline") or {}, d.get("gicp") or {}, d.get("sweeps") or {}
    cfg = d.get("config") or {}
    roof = {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch")}
    roof["kernel"] = (roof["kernel"] or "").split(" (")[0]
    scal = {
        "bev_scatter_frac": r.get("bev_scatter_frac"), "polar_frac": r.get("polar_frac"),
        "frac_of_max_hbm_only_march_only": r.get("frac_of_max_hbm_only_march_only"),
        "gicp_iters_per_s": g.get("iters_per_s"), "gicp_iters_per_s_natural": _get(g, "natural", "iters_per_s"),
        "gicp_iters_per_s_cold5": _get(g, "cold", "iters_per_s"), "gicp_searched_fraction": g.get("searched_fraction"),
        "gicp_natural_pairs_per_s": _get(g, "natural", "pairs_per_s"), "gicp_pairs_per_s_incl_covariances": g.get("pairs_per_s_incl_covariances"),
        "gicp_linearize_frac": r.get("gicp_linearize_frac"), "gicp_nn_certify_frac": r.get("gicp_nn_certify_frac"),
        "gicp_cov_from_knn_frac": _get(g, "roofline", "k_cov_from_knn", "frac"), "gicp_cov_from_knn_ms": r.get("gicp_cov_from_knn_ms"),
        "gicp_knn_select_ms": r.get("gicp_knn_select_ms"),
        "ring_q1_frac": _get(sw, "ring_q1", "hbm_frac"), "ringpp_q1_frac": _get(sw, "ringpp_q1", "hbm_frac"),
        "ring_q4_pairs_per_s": _get(sw, "ring_q4", "pairs_per_s"), "ring_q8_pairs_per_s": _get(sw, "ring_q8", "pairs_per_s"),
        "ringpp_q4_pairs_per_s": _get(sw, "ringpp_q4", "pairs_per_s"),
        "disco_q1_ms": _get(sw, "disco_q1", "ms"),
        "node_twin_pairs_per_s": _get(d, "node_shape", "twin_pairs_per_s"),
        "node_unchanged_loop_pairs_per_s": _get(d, "node_shape", "reference_loop_through_dropin", "pairs_per_s"),
        "pygicp_align_ms": _get(d, "dropin_latency", "pygicp_align_downsampled_ms"),
        "ringpp_build_scans_per_s": _get(d, "builds", "ringpp_build", "scans_per_s"),
        "host_fed_scans_per_s": _get(d, "builds", "ingest_from_host", "scans_per_s"),
        "host_fed_frac_of_copy": _get(d, "builds", "ingest_from_host", "frac_of_pinned_copy"),
    }
    roof.update({k: v for k, v in scal.items() if v is not None})
    cb = d.get("cpu_baseline")
    out = {k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                 "dtype", "data")}
    out["config"] = {"workload": cfg.get("workload"),
                     **{k: cfg.get(k) for k in ("pairs_per_rank_per_step", "launches_per_step", "points_per_scan", "exchange", "exchange_impl")}}
    out["timed_region_s"] = d.get("timed_region_s")
    v = d.get("verify")
    out["verify"] = {"ok": v.get("ok"), "checked": v.get("checked")} if v else None
    out["roofline"] = roof
    if cb:
        out["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "sample": (cb.get("sample") or "")[:160], "gicp_iters_per_s": _get(cb, "gicp", "iters_per_s")}
    pg = _get(d, "exchange", "process_group")
    if pg:
        out["exchange"] = {"process_group": pg, "impl": _get(d, "exchange", "impl"), "verify_ok": _get(d, "exchange", "verify", "ok")}
    out["detail"] = detail_name
    keep_exact = {"value": out["value"], "ms_per_step": out["ms_per_step"], "timed_region_s": out["timed_region_s"]}
    out = _sig(out)
    out.update({k: (None if v is None else float(v)) for k, v in keep_exact.items()})
    out["roofline"]["frac"] = None if roof.get("frac") is None else float(roof["frac"])
    out["roofline"]["achieved"] = None if roof.get("achieved") is None else float(roof["achieved"])
    out["roofline"]["traffic"] = roof.get("traffic")
    text = json.dumps(out, allow_nan=False, separators=(",", ":"))
    while len(text) >= COMPACT_LIMIT and len(out["roofline"]) > 8:      # never reached today (~2.5 KB); a later block must not break the driver's parser
        out["roofline"].popitem()
        text = json.dumps(out, allow_nan=False, separators=(",", ":"))
    assert len(text) < COMPACT_LIMIT, len(text)
    return text


def emit(detail, detail_file):
    """Full result -> `detail_file` (strict JSON, one object); the compact contract line -> the LAST line of stdout, nothing after it."""
    detail = _strict(detail)
    name = None
    if detail_file:
        # the contract line must appear whatever happens to the detail file: a read-only working directory falls back to the temporary directory
        import tempfile
        for path in (detail_file, os.path.join(tempfile.gettempdir(), os.path.basename(detail_file))):
            try:
                with open(path, "w") as f:
                    json.dump(detail, f, allow_nan=False)
                    f.write("\n")
                name = os.path.basename(path) if path == detail_file else path
                break
            except OSError as e:
                print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
    text = compact_line(detail, name)
    sys.stdout.flush()
    sys.stderr.flush()
    try:   # librccl printf()s a banner into libc's stdout buffer, which would otherwise be flushed at exit, after the JSON
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    print(text, flush=True)


# ------------------------------------------------------------------------------------------------------------ main
def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: re-exec through torch.distributed.run with N ranks on this
    node (rendezvous on 127.0.0.1, a port that is free right now), pass the command line through, return its exit code.  Fails loudly when the
    node has fewer than N GPUs (sharing one GPU between ranks is the gloo test mode only: MRS_BENCH_SHARE_GPU=1)."""
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("MRS_BENCH_SHARE_GPU") != "1":
        print(f"bench.py: --gpus {n} needs {n} visible GPUs, found {have} (one rank per GPU)", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd)


def after_timed_region(S):
    """Everything between the clock stopping and the result line, on every rank: per-launch kernel times from the recorded events, the GICP leg,
    the exchange check (--verify-exchange), the two database designs side by side (N > 1), the side stream's own time.  S: main()'s locals."""
    # per launch of B scans; entries that cover several launches (descriptor kernel, grouped correlation, a group's sweeps) carry their count.
    # With --sweep-stream side the sweep figure is stream time on the side stream (it includes waiting for compute units the other stream holds)
    kern_ms = {k: float(sum(t[0].elapsed_time(t[1]) for t in v) / sum((t[2] if len(t) > 2 else 1) for t in v)) for k, v in S.ev.items() if v}

    extra = {}
    gicp_res = None
    if S.args.gicp_pairs > 0:
        S.fence()
        gicp_res = gicp_leg(S.local_rank, S.rank, S.args.gicp_pairs, S.args.gicp_iters)
        if S.dist_on:   # whole-job GICP rate: all ranks' iterations / slowest rank's time
            t = torch.tensor([gicp_res["align_s"]], dtype=torch.float64, device=S.device)    # the cold-start forced protocol
            S.dist.all_reduce(t, op=S.dist.ReduceOp.MAX)
            gicp_res["iters_per_s"] = S.world * S.args.gicp_pairs * S.args.gicp_iters / float(t.item())
            gicp_res["pairs"] = S.world * S.args.gicp_pairs

    verify = None
    if S.dist_on and S.args.verify_exchange:
        # the last launch scored its new descriptors against what the exchange delivered (fp16 replicas / fetched fp32 rows) of the
        # descriptors every rank built DEPTH launches earlier: gather those exact fp32 entries and score the same pairs against them
        S.fence()
        c = S.CH - 1
        exact_db = shard.allgather_ragged(torch.view_as_real(S.spec32[S.db_slot(c)]).contiguous())
        exact_db = torch.view_as_complex(exact_db.contiguous())
        d_ex, a_ex = ring.corr_pairs_fft(S.spec32[c].contiguous(), exact_db[S.cand_idx[c].long()].contiguous())
        err = (d_ex - S.out_dist[c]).abs()
        verify = {"checked": int(err.numel()), "max_abs_dist_error": float(err.max()),
                  "angle_mismatches": int((a_ex != S.out_ang[c]).sum()), "remote_candidates": int((S.cand_idx[c] // S.B != S.rank).sum())}
        if S.EXCH == "fetch":
            # the fetched rows are the owners' entries bit for bit, and the sharded top-1 sweep equals a sweep over the gathered database
            verify["fetched_rows_bit_identical"] = bool(torch.equal(torch.view_as_real(S.last_fetched[0]),
                                                                    torch.view_as_real(exact_db[S.cand_idx[c].long()])))
            d_full, _ = ring.corr_sweep_fft(S.spec32[c, :1].contiguous(), exact_db)
            v_full, r_full = torch.min(d_full, 1)
            verify["sweep_value_equal"] = bool(float(v_full[0]) == float(S.sweep_val[c]))
            verify["sweep_row_equal"] = bool(int(r_full[0]) == int(S.sweep_row[c]))
        tol = 2e-3 if (S.EXCH == "allgather" and not S.REP32) else 1e-6          # fp16 replicas differ from the exact entries by < 2e-3 (re-scored near the threshold)
        verify["ok"] = bool(verify["max_abs_dist_error"] < tol and (S.EXCH == "allgather" and not S.REP32 or verify["angle_mismatches"] == 0) and
                            all(verify.get(k, True) for k in ("fetched_rows_bit_identical", "sweep_value_equal", "sweep_row_equal")))

    topk_cmp = None
    if S.dist_on:
        # the two database designs of SURVEY.md 8(e) side by side: (a) replicate (all-gather the descriptors, every rank sweeps
        # everything: what the step does) vs (b) keep the database sharded, all-gather the QUERIES, exchange top-k rows only
        q_local = S.spec32[0, :4].contiguous()
        S.fence()

        def design_a():
            full = shard.allgather_ragged(torch.view_as_real(S.spec32[0]).to(torch.float16))
            d, a = ring.corr_sweep_fft(q_local, full)
            return torch.topk(d, 4, dim=1, largest=False)

        def design_b():
            q_all = shard.allgather_ragged(torch.view_as_real(q_local).contiguous())
            return shard.sharded_topk_sweep(torch.view_as_complex(q_all), S.spec32[0], ring.corr_sweep_fft, 4)
        ms_a, ms_b = ev_ms(design_a, reps=3, warm=1), ev_ms(design_b, reps=3, warm=1)
        topk_cmp = {"replicate_db_ms": ms_a, "sharded_topk_ms": ms_b, "queries_per_rank": 4, "db_rows_per_rank": S.B,
                    "replicate_bytes_in_per_rank": (S.world - 1) * S.B * 29280, "sharded_bytes_in_per_rank": (S.world - 1) * 4 * (58560 + 4 * 16)}

    side_stream = None
    if S.SIDE_SWEEP and "sweep" in kern_ms:
        # on the side stream the interval between a batch's events is STREAM time: it contains the wait for compute units the descriptor
        # kernel holds, so it is not a kernel duration and must not be added to the others.  The sweep's own duration is measured stand-alone
        # right here (same entries), the stream time is reported apart
        S.fence()
        side_stream = {"sweeps_stream_time_ms_per_launch": kern_ms.pop("sweep"),
                       "note": "time between the events around a batch of sweeps on the side stream / launches in the batch: includes waiting for "
                               "compute units held by the descriptor kernel on the compute stream; kernel_ms.sweep_standalone is the sweep's own duration"}
        kern_ms["sweep_standalone"] = ev_ms(lambda: torch.min(ring.corr_sweep_fft(S.spec32[S.wslot(0), :1], S.spec32[S.db_slot(0)])[0], 1))

    return {"gicp_res": gicp_res, "kern_ms": kern_ms, "side_stream": side_stream, "topk_cmp": topk_cmp, "verify": verify}


def result_line(S):
    """Rank 0: the full result (every block; emit() writes it to the detail file and prints the compact contract line).  S: main()'s locals + what
    after_timed_region returned."""
    pmc = load_pmc()
    cells = 120 * 120
    bev_bytes = S.B * (12 * N_POINTS + 4 * cells)          # 12 B/point + 4 B/cell
    if S.FUSE:
        # the rasteriser no longer runs on its own in the timed region: its stand-alone roofline is measured right here (same
        # scans, same box), the timed region's dominant kernel is the fused one (12 B/point in, one normalised sinogram out)
        xyz0, offs0 = S.chunks[0]
        S.kern_ms["bev_standalone"] = ev_ms(lambda: bev.cart_bev(xyz0, offs0, 1, 1, 120, 120, 1, out=S.img.view(S.B, -1)))
        S.kern_ms["radon_standalone"] = ev_ms(lambda: S.plan.forward(S.img.view(S.B, 120, 120), raw=False, normalized=True))
        fused_bytes = S.B * (12 * N_POINTS + 4 * cells)
        achieved = fused_bytes / (S.kern_ms["bev_radon"] * 1e-3) / 1e9
    else:
        achieved = bev_bytes / (S.kern_ms["bev"] * 1e-3) / 1e9
    line = {
        "metric": "loop-candidate pairs/sec (BEV+Radon+corr), 120k-pt scans",
        "value": S.world * S.B * S.CH * S.args.steps / S.elapsed,
        "unit": "pairs/s",
        "n_gpus": S.world,
        "steps": S.args.steps,
        "warmup": S.args.warmup,
        "ms_per_step": 1e3 * S.elapsed / S.args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1] batched: 120k-pt synthetic lidar scans -> Cartesian BEV 120x120x1 -> Radon "
                               "120x120 -> normalise -> half spectrum -> FFT-domain rotation correlation vs 1 candidate of the "
                               "database (+ 1 query per launch swept over the database)",
                   "pairs_per_rank_per_step": S.B * S.CH, "pairs_per_launch": S.B, "launches_per_step": S.CH,
                   "database_rows_swept_per_launch": S.NDB,      # grows with the world size: the replicated database is world x B rows
                   "resident_scan_bytes_per_rank": S.B * S.CH * 12 * N_POINTS, "points_per_scan": N_POINTS,
                   "parallelism": f"scan-sharded x{S.world}" + ("" if not S.dist_on else
                                                              " + RCCL all-gather of fp16 descriptor replicas, candidates and sweeps read the "
                                                              "replicated database, owner re-scoring" if S.EXCH == "allgather" else
                                                              " + database kept sharded: RCCL all-to-all of the candidate rows asked for (exact "
                                                              "fp32, pre-planned), per-launch query all-gathered for a sharded top-1 sweep"),
                   "exchange": S.EXCH, "exchange_impl": (S.args.exchange_impl if S.dist_on else None), "fused_grid": S.fused_grid if S.FUSE else None},
        "timed_region_s": S.elapsed,
        "setup_s": S.setup_s,
        "kernel_ms": S.kern_ms,
        "side_stream": S.side_stream,
        "roofline": {"kernel": "k_cart_lds (BEV scatter)", "bound": "hbm", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": pmc.get("k_cart_lds", {}).get("hbm_bytes"), "algorithmic_bytes_per_launch": bev_bytes,
                     "traffic_source": PMC_NAME + " (rocprofv3 --pmc passes of tools/pmc_targets.py)" if pmc.get("k_cart_lds") else None},
        "gicp": S.gicp_res,
    }
    if S.FUSE:
        # one kernel with an HBM-bound half (rasteriser) and a VALU-bound half (Radon march) per workgroup, overlapped across
        # compute units: its time is bounded below by max(HBM time of the points, VALU time of the rays), not by either alone
        r = pmc.get("k_bev_radon3") or pmc.get("k_bev_radon2", {})
        if r.get("hbm_bytes"):        # the PMC pass profiles the kernel at 16 x 1024 scans per launch: per 1024 scans like everything else here
            per = (r.get("launch_scans") or 1024) / 1024.0
            r = dict(r, hbm_bytes=r["hbm_bytes"] / per, valu_pipe_cycles_est=(r.get("valu_pipe_cycles_est") or 0) / per or None,
                     counters={k: v / per for k, v in (r.get("counters") or {}).items()})
        sa = bev_bytes / (S.kern_ms["bev_standalone"] * 1e-3) / 1e9
        line["config"]["fused_launches"] = S.FUSE
        line["config"]["corr_launches_grouped"] = S.FUSE if S.GROUP_CORR else 1
        line["config"]["sweep_stream"] = "side" if (S.SIDE_SWEEP or S.EXCH == "fetch") else "main"
        line["config"]["sweeps_per_launch"] = S.FUSE if S.SWEEP_BATCH else 1
        if S.SIDE_SWEEP:
            line["config"]["sweep_join"] = S.args.sweep_join
        if S.args.fused_wgs > 0:
            line["config"]["fused_persistent_workgroups"] = S.args.fused_wgs
        line["config"]["database_slots"] = "two sets, alternating per step" if S.RING_DB else "one set + copies of the previous step's last entries"
        line["roofline"] = {"kernel": f"k_bev_radon3 (BEV scatter + Radon + normalise, {S.FUSE} x {S.B} scans per launch)", "bound": "hbm",
                            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                            "traffic": r.get("hbm_bytes"), "algorithmic_bytes_per_launch": bev_bytes,
                            "traffic_source": PMC_NAME if r else None,
                            "valu_issue_frac": r.get("valu_issue_frac"), "lds_busy_frac": r.get("lds_busy_frac"),
                            "valu_roofline": valu_roofline(r, S.kern_ms["bev_radon"]),
                            "note": "per 1024 scans; the kernel also carries the VALU-bound Radon march (1.47 M two-tap samples per image), "
                                    "so the HBM fraction of the fused kernel is below the stand-alone rasteriser's by construction",
                            "ms_vs_separate_kernels": {"fused": S.kern_ms["bev_radon"], "bev_standalone": S.kern_ms["bev_standalone"],
                                                       "radon_standalone": S.kern_ms["radon_standalone"]}}
        line["roofline_bev_scatter"] = {"kernel": "k_cart_lds (BEV scatter, stand-alone launch outside the timed region)", "bound": "hbm",
                                        "achieved": sa, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sa / HBM_PEAK_GBS,
                                        "ms": S.kern_ms["bev_standalone"], "traffic": pmc.get("k_cart_lds", {}).get("hbm_bytes"),
                                        "algorithmic_bytes_per_launch": bev_bytes}
    if S.gicp_res:
        # north_star's own targets as flat scalars of `roofline` (the driver's record keeps scalars of this block)
        gr = S.gicp_res["roofline"]
        line["roofline"].update({
            "gicp_iters_per_s": S.gicp_res["iters_per_s"], "gicp_iters_per_s_warm": S.gicp_res["warm"]["iters_per_s"],
            "gicp_iters_per_s_cold5": S.gicp_res["cold"]["iters_per_s"],
            "gicp_natural_pairs_per_s": S.gicp_res["natural"]["pairs_per_s"], "gicp_pairs_per_s_incl_covariances": S.gicp_res["pairs_per_s_incl_covariances"],
            "gicp_pairs_per_s_incl_covariances_shared_submaps": S.gicp_res["shared_submaps"]["pairs_per_s_incl_covariances"],
            "gicp_searched_fraction_natural": S.gicp_res["natural"]["searched_fraction"],
            "gicp_linearize_ms": gr["k_linearize"]["ms"], "gicp_linearize_gbs": gr["k_linearize"]["achieved"], "gicp_linearize_frac": gr["k_linearize"]["frac"],
            "gicp_linearize_error_only_frac": gr["k_linearize_error_only"]["frac"],
            "gicp_nn_round3_all_ms": S.gicp_res["kernel_ms"]["search_round3_all"], "gicp_nn_round4_all_ms": S.gicp_res["kernel_ms"]["search_round4_all"],
            "gicp_nn_certify_ms": S.gicp_res["kernel_ms"]["certify"], "gicp_nn_certify_frac": gr["k_nn_certify (unchanged pose)"]["frac"],
            "gicp_nn_certified_pass_1mm_ms": S.gicp_res["kernel_ms"]["certify_plus_worklist_1mm"],
            "gicp_knn_select_ms": S.gicp_res["kernel_ms"]["knn_select"], "gicp_cov_from_knn_ms": S.gicp_res["kernel_ms"]["cov_from_knn"],
            "gicp_pairs": S.gicp_res["pairs"]})
        line["roofline"]["gicp"] = gr
    if S.FUSE:
        line["roofline"]["bev_scatter_frac"] = line["roofline_bev_scatter"]["frac"]
        line["roofline"]["bev_scatter_gbs"] = line["roofline_bev_scatter"]["achieved"]
    if S.dist_on:
        # bytes a rank receives per launch under either design, and the inbound rate each would need at the measured step time
        ag_launch = (S.world - 1) * S.B * (58560 if S.REP32 else 29280)
        if S.EXCH == "fetch":
            fetch_launch = float(np.mean([pl.bytes_in(58560) for pl in S.fetch_plans]))
        else:
            fetch_launch = (S.world - 1) / S.world * S.B * 58560                    # expected for uniformly drawn candidates
        sweep_launch = (S.world - 1) * (58560 + S.world * 16)                     # the other ranks' queries + their packed top-1 answers
        step_s = 1e-3 * line["ms_per_step"]
        undecided = None
        if S.rescorer is not None:
            # proof that no loop decision rests on a replica score: after re-scoring, every (query, candidate) whose REPLICA distance was
            # within the margin of the acceptance threshold carries the owner's exact value (rescore.requested of them, in rescore.rounds
            # fixed-size rounds that only end when every rank reports none left); the rest differ from exact by < 2e-3 < margin
            undecided = {"replica_margin": S.rescorer.margin, "threshold": S.rescorer.threshold, "requested": S.rescorer.stats["requested"],
                         "rounds": S.rescorer.stats["rounds"], "left_undecided": S.rescorer.stats["still_ambiguous"],
                         "note": "measured on the last call: left_undecided = ambiguous entries of its output that were not replaced by an owner's exact "
                                 "score; the loop ends only when an all-reduce(MAX) of the per-rank remaining counts is 0"}
        line["exchange"] = {"design": S.EXCH, "impl": S.args.exchange_impl,
                            "impl_note": ("data-path collectives through the C ABI (mrs_exchange_allgather / mrs_exchange_fetch_planned, the library's own RCCL "
                                          "communicator of %d ranks) on a communication stream" % S.xch.world) if S.CABI else "torch.distributed collectives",
                            "process_group": {"backend": S.dist.get_backend(), "world_size": S.dist.get_world_size(), "gpus_flag": S.args.gpus,
                                              "devices_visible": torch.cuda.device_count()},
                            "replica": (S.args.replica if S.EXCH == "allgather" else None),
                            "decisions_on_replica_scores": (None if S.EXCH != "allgather" or S.rescorer is None else S.rescorer.stats["still_ambiguous"]),
                            "rescore_proof": undecided,
                            "allgather": {"format": ("exact fp32 half spectra, 58 560 B per descriptor" if S.REP32 else
                                                     "fp16 half spectra, 29 280 B per descriptor") + ", every descriptor to every rank",
                                          "bytes_in_per_rank_per_launch": ag_launch,
                                          "inbound_gbs_needed_at_this_rate": (ag_launch * S.CH / step_s / 1e9) if S.world > 1 else None},
                            "fetch": {"format": "exact fp32 half spectra, 58 560 B per candidate row actually asked for (pre-planned all-to-all) + one "
                                                "query per rank and launch all-gathered for the sharded top-1 sweep",
                                      "bytes_in_per_rank_per_launch": fetch_launch + sweep_launch,
                                      "rows_bytes_in_per_rank_per_launch": fetch_launch, "sweep_bytes_in_per_rank_per_launch": sweep_launch,
                                      "inbound_gbs_needed_at_this_rate": ((fetch_launch + sweep_launch) * S.CH / step_s / 1e9) if S.world > 1 else None,
                                      "launches_ahead": S.FETCH_AHEAD if S.EXCH == "fetch" else None},
                            "compute_stream_wait_ms_per_launch": S.kern_ms.get("wait"),
                            "compute_stream_wait_ms_per_step": S.kern_ms.get("wait", 0.0) * S.CH,
                            "rescore": S.rescorer.stats if S.rescorer else None, "designs": S.topk_cmp, "verify": S.verify}
    if not S.args.no_extra_legs:
        # polar BEV (the rasteriser north_star names), DiSCO layout 40 x 120 x 20, same scans
        xyz0, offs0 = S.chunks[0]
        pcells = 40 * 120 * 20
        ms = ev_ms(lambda: bev.polar_bev(xyz0, offs0, 1, 1, 40, 120, 20))
        pbytes = S.B * (12 * N_POINTS + 4 * pcells)
        line["roofline_polar"] = {"kernel": "k_polar_lds (polar BEV scatter, 40x120x20)", "bound": "hbm", "achieved": pbytes / ms / 1e6,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": pbytes / ms / 1e6 / HBM_PEAK_GBS, "ms": ms,
                                  "traffic": pmc.get("k_polar_lds", {}).get("hbm_bytes"), "algorithmic_bytes_per_launch": pbytes}
        samples = 1.47e6 * S.B                       # two-tap samples per launch (120 angles x 120 rays x ~102 steps)
        radon_ms = S.kern_ms.get("radon", S.kern_ms.get("radon_standalone"))
        r = pmc.get("k_radon2", {})
        line["roofline_radon"] = {"kernel": "k_radon2 (two images per workgroup)", "bound": "valu+lds (not HBM: 115 KB per image)",
                                  "ms": radon_ms, "samples_per_s": samples / (radon_ms * 1e-3),
                                  "valu_issue_frac": r.get("valu_issue_frac"), "lds_busy_frac": r.get("lds_busy_frac"),
                                  "lds_bank_conflict_frac_of_lds": r.get("lds_bank_conflict_frac"),
                                  "hbm_bytes": r.get("hbm_bytes"), "source": PMC_NAME if r else None}
        # the same box's device-to-device copy rate next to the nominal peak (a copy moves 2 bytes per byte copied)
        src_buf = S.chunks[0][0]
        dst_buf = torch.empty_like(src_buf)
        ms = ev_ms(lambda: dst_buf.copy_(src_buf), reps=5, warm=2)
        copy_gbs = 2 * src_buf.numel() * 4 / ms / 1e6
        ms_fill = ev_ms(lambda: dst_buf.zero_(), reps=5, warm=2)
        fill_gbs = src_buf.numel() * 4 / ms_fill / 1e6
        del dst_buf
        # the polar rasteriser writes 21 % of its bytes (384 KB of cells per scan); this memory system streams writes slower than reads, so its
        # bound is the mix of the two stream rates measured on this box: the read-mostly Cartesian rasteriser's rate for the points, a fill's
        # rate for the cells
        rp = line["roofline_polar"]
        read_gbs = line.get("roofline_bev_scatter", {}).get("achieved")
        if read_gbs:
            rd, wr = S.B * 12 * N_POINTS, S.B * 4 * pcells
            mix = (rd + wr) / (rd / read_gbs + wr / fill_gbs)
            rp.update({"read_stream_gbs": read_gbs, "write_stream_gbs": fill_gbs, "write_share": wr / (rd + wr), "mix_bound_gbs": mix,
                       "frac_of_mix_bound": rp["achieved"] / mix})
            line["roofline"]["polar_frac_of_rw_mix_bound"] = rp["achieved"] / mix
        line["roofline"]["measured_fill_gbs"] = fill_gbs
        line["roofline"]["polar_frac_of_measured_copy"] = rp["achieved"] / copy_gbs
        line["roofline"]["polar_frac"] = line["roofline_polar"]["frac"]
        line["roofline"]["polar_gbs"] = line["roofline_polar"]["achieved"]
        line["roofline"]["measured_copy_gbs"] = copy_gbs
        line["roofline"]["frac_of_measured_copy"] = achieved / copy_gbs
        line["roofline_polar"]["frac_of_measured_copy"] = line["roofline_polar"]["achieved"] / copy_gbs
        line["sweeps"] = sweep_legs(S.device, S.spec32[:S.CH].reshape(-1, 61, 120))
        line["node_shape"] = node_shape_leg(S.device, S.spec32[:S.CH].reshape(-1, 61, 120))
        line["pipeline_shard"] = pipeline_shard_leg(S.device, S.spec32[:S.CH].reshape(-1, 61, 120), S.gicp_res)
        line["builds"] = build_legs(S.device, S.chunks)
        line["dropin_latency"] = dropin_latency_leg(host_scans(S.chunks[0][0], 1)[0])
    if S.FUSE and not S.dist_on and S.args.verify > 0:
        # the outputs of the timed loop's last fused launch are still in norm_group / out_dist / out_ang
        last = ((S.CH - 1) // S.FUSE) * S.FUSE
        line["verify"] = verify_timed_outputs(S.args.verify, make_shard.whole, S.norm_group[:(S.CH - last) * S.B], last, S.out_dist, S.out_ang, S.cand_idx,
                                              lambda c: c - S.DEPTH if c >= S.DEPTH else S.CH - S.DEPTH + c)
        # the per-launch sweeps of the timed loop (possibly issued on the side stream) against fresh ones over the same entries
        bad = 0
        for c in np.random.default_rng(1).choice(S.CH, size=min(S.args.verify, S.CH), replace=False):
            d_f, _ = ring.corr_sweep_fft(S.spec32[S.wslot(int(c)), :1], S.spec32[S.db_slot(int(c))])
            v_f, r_f = torch.min(d_f, 1)
            bad += int(float(v_f[0]) != float(S.sweep_val[int(c)]) or int(r_f[0]) != int(S.sweep_row[int(c)]))
        line["verify"]["sweep_mismatches"] = bad
        line["verify"]["ok"] = bool(line["verify"]["ok"] and bad == 0)
    if S.FUSE and not S.args.no_extra_legs:
        # the other workgroup shape of the fused kernel on the same scans (N > 1 runs per_pair so that RCCL's kernels get in): measured, not assumed
        other = "per_pair" if S.fused_grid == "persistent" else "persistent"
        ng = min(S.FUSE, S.CH)
        S.set_fused_grid(other)
        ms_other = ev_ms(lambda: ring.ring_descriptors_fused(make_shard.whole[:ng].view(-1), S.group_offs[:ng * S.B + 1], raw=False, normalized=True,
                                                             out_norm=S.norm_group[:ng * S.B]), reps=3, warm=1) / ng
        S.set_fused_grid(S.fused_grid)
        line["roofline"]["fused_grid_ms_per_launch"] = {S.fused_grid: S.kern_ms["bev_radon"], other: ms_other}
        # the kernel's two phase floors in the same run (measurement option of the plan: the same kernel without its ray march / without its
        # rasteriser) and how far the kernel is from a compute unit that overlapped them perfectly
        fl = {}
        for name, skip in (("full", 0), ("hbm_only_no_march", 2), ("march_only_no_rasteriser", 1)):
            S.plan.set_option(S.plan.OPT_FUSED_SKIP, skip)
            fl[name] = ev_ms(lambda: ring.ring_descriptors_fused(make_shard.whole[:ng].view(-1), S.group_offs[:ng * S.B + 1], raw=False, normalized=True,
                                                                 out_norm=S.norm_group[:ng * S.B]), reps=3, warm=1) / ng
        S.plan.set_option(S.plan.OPT_FUSED_SKIP, 0)
        line["roofline"]["fused_phase_floors_ms_per_launch"] = fl
        line["roofline"]["frac_of_max_hbm_only_march_only"] = max(fl["hbm_only_no_march"], fl["march_only_no_rasteriser"]) / fl["full"]
    if S.world == 1 and not S.args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(host_scans(S.chunks[0][0], min(S.args.cpu_sample, S.B)))
    return line


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="scan pairs per launch")
    ap.add_argument("--chunks", type=int, default=48, help="launches per step (resident shard = batch x chunks scans)")
    ap.add_argument("--fuse", type=int, default=FUSE_DEFAULT, help="BEV + Radon + normalisation of this many launches' scans in ONE persistent "
                    "kernel (mrs_ring_descriptors_batch; same bits); 0 = the two kernels per launch")
    ap.add_argument("--gicp-pairs", type=int, default=256, help="120k-pt pairs per rank in the GICP leg (BASELINE configs[2]: 256; 0 = skip)")
    ap.add_argument("--gicp-iters", type=int, default=20)
    ap.add_argument("--cpu-sample", type=int, default=256, help="scans in the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="only the headline step (+ GICP unless --gicp-pairs 0)")
    ap.add_argument("--verify", type=int, default=8, help="after the timed region: this many random outputs of the last fused launch against the "
                    "oracle (N = 1, fused step only; 0 = skip)")
    ap.add_argument("--detail-file", default="bench_detail.json", help="every block of the result (kernel times, legs, rooflines per kernel, the CPU "
                    "baseline's settings) as one JSON object; the last stdout line is the compact contract line (< 4 KB).  '' = no file")
    ap.add_argument("--verify-exchange", action="store_true", help="N > 1: re-derive the last launch's scores from the exact remote entries")
    ap.add_argument("--exchange", choices=("fetch", "allgather"), default="fetch", help="N > 1: how candidate rows and the swept database reach a rank. "
                    "fetch (default): the database stays sharded, a pre-planned all-to-all brings exactly the candidate rows asked for (exact fp32) and "
                    "the per-launch query goes through a sharded top-k sweep; allgather: fp16 replicas of every new descriptor to every rank "
                    "(north_star's wording) + owner re-scoring")
    ap.add_argument("--exchange-impl", choices=("torch", "cabi"), default="torch", help="N > 1: who carries the data-path exchanges (the descriptor all-gather, "
                    "the candidate-row fetch, the per-launch query gather and the top-1 results).  torch: torch.distributed collectives (RCCL underneath); cabi: "
                    "the product's own exchange behind the C ABI (mrs_exchange_allgather / mrs_exchange_fetch_planned: RCCL resolved at run time), on a "
                    "communication stream.  The timing barrier / max over ranks and the owner re-scoring's small messages stay on torch.distributed.  "
                    "Default torch until a multi-GPU record exists for both (the 8-GPU runs are the driver's)")
    ap.add_argument("--corr-group", type=int, choices=(0, 1), default=1, help="N = 1 with --fuse: the half-spectrum + pair-correlation kernel of a "
                    "group's launches in ONE launch right after the group's descriptor kernel (same pairs, same databases: a launch only reads "
                    "entries that are at least one group old), like the descriptor kernel itself; 0 = one launch per 1024 pairs")
    ap.add_argument("--sweep-stream", choices=("main", "side"), default="side", help="N = 1: the per-launch database sweeps on the compute stream, or on "
                    "a second HIP stream so that they fill the tail of the next descriptor kernel (N > 1 with --exchange fetch always uses the "
                    "side stream); joined before the step ends")
    ap.add_argument("--sweep-batch", type=int, choices=(0, 1), default=1, help="N = 1 with grouped correlation: the per-launch sweeps of a group (one query per "
                    "launch, each against the database slot its launch reads) as ONE kernel launch over (query, slot) blocks instead of one launch per sweep")
    ap.add_argument("--fused-wgs", type=int, default=0, help="persistent workgroups of the descriptor kernel (0 = one per compute unit); fewer leave "
                    "compute units to the side-stream sweeps while the descriptor kernel runs")
    ap.add_argument("--sweep-join", choices=("step", "lag"), default="lag", help="--sweep-stream side: the compute stream joins the side stream at the "
                    "end of every step, or (lag) only waits for all but the step's LAST batch of sweeps, which then runs under the next step's "
                    "first descriptor kernel (it reads only the slot set the next step does not write); everything is joined before the clock stops")
    ap.add_argument("--fused-grid", choices=("auto", "persistent", "per_pair"), default="auto", help="workgroups of the fused descriptor kernel: persistent (one "
                    "per compute unit) or one per pair of scans; auto = per_pair only with --exchange allgather at N > 1 (lets RCCL's kernels in "
                    "while the descriptor kernel runs), persistent otherwise")
    ap.add_argument("--replica", choices=("f16", "f32"), default="f16", help="--exchange allgather: fp16 replicas (29 280 B, half the bytes) + exact "
                    "owner re-scoring of every candidate within 2e-3 of the acceptance threshold, or the exact fp32 entries themselves (58 560 B, no re-scoring)")
    return ap.parse_args()


def main():
    args = parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))               # bare `python bench.py --gpus N`: start the N ranks ourselves (one JSON line from rank 0)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or os.environ.get("MRS_BENCH_FORCE_DIST") == "1"   # the env var exercises the RCCL path on 1 GPU
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or run bare `python bench.py "
                 f"--gpus {args.gpus}`, which starts the ranks itself)")
    if world > 1 and torch.cuda.device_count() < world and os.environ.get("MRS_BENCH_SHARE_GPU") != "1":
        sys.exit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible (one rank per GPU; MRS_BENCH_SHARE_GPU=1 "
                 f"MRS_BENCH_BACKEND=gloo is the correctness-only test mode that shares one)")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    # test hook: MRS_BENCH_BACKEND=gloo MRS_BENCH_SHARE_GPU=1 runs the N > 1 control flow with several ranks on ONE GPU
    # (RCCL refuses two ranks per device; gloo stages the collectives through the host) -- correctness only, not a measurement
    backend = os.environ.get("MRS_BENCH_BACKEND", "nccl")
    if os.environ.get("MRS_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))      # only reached without a launcher (MRS_BENCH_FORCE_DIST at world 1)
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend)

    B, CH = args.batch, args.chunks
    t_setup = time.perf_counter()
    chunks = make_shard(B, CH, rank, device)
    plan = ring.ring_plan(local_rank)
    img = torch.empty((B, 1, 120, 120), dtype=torch.float32, device=device)
    # the rank's exact database entries: Hermitian half spectra [61][120] complex64 (58 560 B) of the normalised
    # sinograms of every resident scan; see RING_DB below for where the entries of the previous step live.
    # The database a launch reads is the one built DEPTH launches earlier: its exchange then has DEPTH launches of kernels to
    # hide behind.
    FUSE = max(0, min(args.fuse, CH))
    # with fused groups the exchanges of a group's launches are issued in a burst after its descriptor kernel: the database a
    # launch reads is then one group (+ 2 launches) old, so that the burst has the next group's descriptor kernel to hide behind
    DEPTH = min(FUSE + 2, CH) if FUSE else 2
    assert CH >= DEPTH
    # N = 1: two sets of CH slots, written alternately (step parity): launch c writes slot c of this step's set and reads the entry built DEPTH
    # launches earlier -- slot c - DEPTH of the same set or, for the step's first DEPTH launches, slot CH - DEPTH + c of the OTHER set (what the
    # previous step's last launches wrote).  Nothing is copied and no kernel reads a slot that is rewritten within the same step, whatever
    # stream it runs on.  N > 1 keeps one set + DEPTH extra slots that receive a copy of the previous step's last entries at the start of every
    # step (the exchanges read them asynchronously).
    RING_DB = not dist_on
    spec32 = torch.empty(((2 * CH) if RING_DB else (CH + DEPTH), B, 61, 120), dtype=torch.complex64, device=device)
    for c, (xyz, offs) in enumerate(chunks):
        _, _, nrm = ring.ring_descriptors(xyz, offs)
        spec32[c] = ring.half_spectrum(nrm)
    if RING_DB:
        spec32[CH:] = spec32[:CH]
    else:
        spec32[CH:] = spec32[CH - DEPTH:CH]
    parity = [0]                                       # the set the current (or, after the loop, the last) step writes

    def wslot(c):
        return write_slot(c, parity[0], CH, RING_DB)

    def db_slot(c):
        return read_slot(c, parity[0], CH, DEPTH, RING_DB)
    g = torch.Generator(device=device).manual_seed(7 + rank)
    NDB = world * B
    cand_idx = torch.randint(0, NDB, (CH, B), generator=g, device=device, dtype=torch.int32)   # pre-selected candidate rows
    out_dist = torch.empty((CH, B), dtype=torch.float32, device=device)
    out_ang = torch.empty((CH, B), dtype=torch.int32, device=device)
    sweep_val = torch.empty(CH, dtype=torch.float32, device=device)    # best distance / database row of the per-launch sweep
    sweep_row = torch.empty(CH, dtype=torch.int64, device=device)
    # N > 1: the replicated database of the previous launch's descriptors (fp16 replicas, 29 280 B each; the owner keeps
    # the exact fp32 entry) -- at > 1 M descriptors/s/GPU fp32 spectra would exceed what the xGMI links carry
    gathered = None
    REP32 = False
    EXCH = args.exchange if dist_on else None
    CABI = bool(dist_on and args.exchange_impl == "cabi")
    xch = comm_stream = None
    if CABI and world > 1 and os.environ.get("MRS_BENCH_SHARE_GPU") == "1":
        sys.exit("bench.py: --exchange-impl cabi needs one GPU per rank (RCCL refuses two ranks on one device); the shared-GPU gloo mode is for "
                 "--exchange-impl torch only")
    if CABI:
        # the product's own exchange: one RCCL communicator made by the library (its unique id travels through the process group once)
        xch = shard.Exchange(device=local_rank)
        comm_stream = torch.cuda.Stream(device=device)

    class _EvWork:                                     # what a torch.distributed Work offers the step functions: a stream-side wait
        def __init__(self, stream):
            self.ev = torch.cuda.Event(); self.ev.record(stream)

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    def gather_async(out, local, stream=None):
        """every rank's `local` into `out`, asynchronously: (torch) an async all-gather of the process group, (cabi) mrs_exchange_allgather on the
        communication stream (or `stream`) behind everything the current stream has enqueued so far; -> an object with .wait()"""
        if not CABI:
            return dist.all_gather_into_tensor(out, local, async_op=True)
        st = stream if stream is not None else comm_stream
        cur = torch.cuda.current_stream()
        if st is not cur:
            ready = torch.cuda.Event(); ready.record(cur)
            st.wait_event(ready)
        xch.allgather_into(out, local.contiguous(), stream=st)
        local.record_stream(st); out.record_stream(st)
        return _EvWork(st)
    fetch_plans = fetch_q = None
    if EXCH == "allgather":
        REP32 = args.replica == "f32"                  # exact fp32 entries to every rank (twice the bytes, nothing to re-score)
        gathered = [torch.empty((NDB, 61, 120, 2), dtype=torch.float32 if REP32 else torch.float16, device=device) for _ in range(DEPTH + 1)]
        for gbuf in gathered:
            gbuf.copy_(torch.view_as_real(spec32[CH - 1]).to(gbuf.dtype).repeat(world, 1, 1, 1))
    elif EXCH == "fetch":
        # request phase once, ahead of time (the candidates of every launch are known before the step starts): per launch ONE all-to-all
        # of exactly the rows asked for, exact fp32, issued FETCH_AHEAD launches early; the swept query of launch c travels after launch c
        # and its sharded top-1 sweep runs on a side stream while launch c + 1 computes
        shard_rows = [B] * world
        if CABI:
            fetch_plans = [xch.fetch_plan(cand_idx[c].to(torch.int64), B) for c in range(CH)]
        else:
            fetch_plans = [shard.RowFetchPlan(cand_idx[c].to(torch.int64), shard_rows) for c in range(CH)]
        fetch_q = {}
        FETCH_AHEAD = max(1, min(DEPTH, 4))
        ident_idx = torch.arange(B, dtype=torch.int32, device=device)
        side = torch.cuda.Stream(device=device)
        q_all = [torch.empty((world, 61, 120, 2), dtype=torch.float32, device=device) for _ in range(2)]
        sweep_pending = []                             # (launch, work of the query all-gather, event after the launch's kernels)
        last_fetched = [None]
    GROUP_CORR = bool(FUSE and EXCH is None and args.corr_group)
    SIDE_SWEEP = EXCH is None and args.sweep_stream == "side"
    if GROUP_CORR:
        # candidate rows of every launch as rows of ALL database slots laid end to end (slot s = rows [s * B, (s + 1) * B))
        spec_flat = spec32.view(-1, 61, 120)
        flat_cand = []                                 # per step parity
        for par in (0, 1):
            parity[0] = par
            slot_of = torch.tensor([db_slot(c) for c in range(CH)], dtype=torch.int32, device=device)
            flat_cand.append((slot_of[:, None] * B + cand_idx).contiguous())
        parity[0] = 0
    if SIDE_SWEEP:
        side = torch.cuda.Stream(device=device)
    SWEEP_BATCH = bool(GROUP_CORR and RING_DB and args.sweep_batch)
    if SWEEP_BATCH:
        # per step parity: the pool entry of every launch's query (first new descriptor of the launch) and the first entry of the slot it sweeps
        sw_q, sw_db = [], []
        for par in (0, 1):
            parity[0] = par
            sw_q.append(torch.tensor([wslot(c) * B for c in range(CH)], dtype=torch.int64, device=device))
            sw_db.append(torch.tensor([db_slot(c) * B for c in range(CH)], dtype=torch.int64, device=device))
        parity[0] = 0
        sw_d = torch.empty((FUSE, B), dtype=torch.float32, device=device)
        sw_a = torch.empty((FUSE, B), dtype=torch.int32, device=device)
    pending = []                                       # (work, source tensor) of the exchanges still in flight, oldest first
    launch_no = [0]
    rescorer = shard.OwnerRescorer(DIST_THRESHOLD, margin=2e-3, slots=64) if (EXCH == "allgather" and not REP32) else None
    setup_s = time.perf_counter() - t_setup

    ev = {k: [] for k in ("bev", "radon", "bev_radon", "corr", "sweep", "wait")}
    # auto: persistent workgroups unless the exchange needs RCCL's kernels to run WHILE the descriptor kernel does.  allgather issues a group's
    # exchanges in a burst that hides behind the next group's descriptor kernel (persistent workgroups would keep RCCL out until the launch
    # ends); fetch moves its rows between the small per-launch kernels, 4 launches ahead, so nothing waits behind the descriptor kernel
    fused_grid = args.fused_grid if args.fused_grid != "auto" else ("per_pair" if EXCH == "allgather" else "persistent")

    def set_fused_grid(mode):
        # per_pair: one workgroup per pair of scans instead of persistent ones.  A persistent workgroup holds its compute unit's whole
        # register file (4 waves x 128 VGPRs per SIMD) until the launch ends, so RCCL's kernels could not start before that;
        # workgroups that retire every ~0.2 ms let the collective's workgroups in between them
        plan.set_option(plan.OPT_FUSED_GRID, 65535 if mode == "per_pair" else max(0, args.fused_wgs))
        plan.set_option(plan.OPT_FUSED_STAGGER_US, 0 if mode == "per_pair" else 70)
    if FUSE:
        set_fused_grid(fused_grid)
        whole = make_shard.whole                                   # [CH][B][3][N], one allocation
        norm_group = torch.empty((FUSE * B, 120, 120), dtype=torch.float32, device=device)
        group_offs = torch.arange(FUSE * B + 1, dtype=torch.int64, device=device) * N_POINTS

    def run_sharded_sweep(c, qwork):
        """(side stream) the queries of launch c, one per rank, against this rank's shard of the database launch c reads; only the
        best (dist, angle, global row) per query travels back (shard.sharded_topk_sweep, static shapes, one packed collective)"""
        qwork.wait()
        qs = torch.view_as_complex(q_all[c % 2])
        dk, ak, rk = shard.sharded_topk_sweep(qs, spec32[db_slot(c)], ring.corr_sweep_fft, 1, shard_rows=shard_rows, packed=True, exchange=xch)
        sweep_val[c] = dk[rank, 0]; sweep_row[c] = rk[rank, 0]

    sweep_batches = []                                 # this step's batches of side-stream sweeps: (first launch, event after the batch)

    def issue_side_sweeps(launches, record):
        """(N = 1) one new query per launch against the database that launch reads, on the side stream: the compute stream goes on with
        the next descriptor kernel, the sweeps run where compute units are free (the tail of that kernel, between the small kernels)"""
        ready = torch.cuda.Event(); ready.record()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            s0 = torch.cuda.Event(enable_timing=True) if record else None
            if record:
                s0.record()
            if SWEEP_BATCH and len(launches) > 1:
                c0, ng = launches[0], len(launches)
                d, a = ring.corr_sweep_fft_blocks(spec_flat, sw_q[parity[0]][c0:c0 + ng], sw_db[parity[0]][c0:c0 + ng], B, out=(sw_d[:ng], sw_a[:ng]))
                torch.min(d, 1, out=(sweep_val[c0:c0 + ng], sweep_row[c0:c0 + ng]))
            else:
                for cc in launches:
                    d, a = ring.corr_sweep_fft(spec32[wslot(cc), :1], spec32[db_slot(cc)])
                    torch.min(d, 1, out=(sweep_val[cc:cc + 1], sweep_row[cc:cc + 1]))
            if record:
                s1 = torch.cuda.Event(enable_timing=True); s1.record()
                ev["sweep"].append((s0, s1, len(launches)))
            fin = torch.cuda.Event(); fin.record()
            sweep_batches.append((launches[0], fin))

    # ---- one step = every launch of the shard once.  The descriptor part is common; what follows it (correlation against the candidates,
    # the top-1 sweep, the exchange between ranks) is one function per exchange mode: step_local (N = 1), step_allgather, step_fetch.
    def mark():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def begin_step():
        if RING_DB:
            parity[0] ^= 1                             # this step writes the other set of slots
        else:
            spec32[CH:] = spec32[CH - DEPTH:CH]        # last launches of the previous step = databases of this step's first

    def descriptors(c, xyz, offs, record):
        """Normalised sinograms of launch c.  Fused kernel: the scans of the next FUSE launches are rasterised + Radon-transformed + normalised
        when the first of them comes up (N = 1 with grouped correlation: their half spectra, correlations and side-stream sweeps are issued
        there too).  Returns (norm [B,120,120], e0, e1, e2): the events around the two stand-alone kernels, None where they do not apply."""
        if FUSE:
            if c % FUSE == 0:                  # rasterise + Radon + normalise the scans of the next FUSE launches in one kernel
                ng = min(FUSE, CH - c)
                ef0 = mark() if record else None
                ring.ring_descriptors_fused(whole[c:c + ng].view(-1), group_offs[:ng * B + 1], raw=False, normalized=True,
                                            out_norm=norm_group[:ng * B])
                if record:
                    ev["bev_radon"].append((ef0, mark(), ng))
                if GROUP_CORR:                 # half spectra (kept: database entries) + correlation with the candidates, all launches of the group
                    ec0 = mark() if record else None
                    ring.spectrum_corr_pairs_db(norm_group[:ng * B], spec_flat, flat_cand[parity[0] if RING_DB else 0][c:c + ng].view(-1),
                                                out=(out_dist[c:c + ng].view(-1), out_ang[c:c + ng].view(-1)),
                                                spec_out=spec32[wslot(c):wslot(c) + ng].view(-1, 61, 120))
                    if record:
                        ev["corr"].append((ec0, mark(), ng))
                    if SIDE_SWEEP:
                        issue_side_sweeps(range(c, c + ng), record)
            return norm_group[(c % FUSE) * B:(c % FUSE + 1) * B], None, None, (mark() if record else None)
        e0 = mark() if record else None
        bev.cart_bev(xyz, offs, 1, 1, 120, 120, 1, out=img.view(B, -1))
        e1 = mark() if record else None
        _, norm = plan.forward(img.view(B, 120, 120), raw=False, normalized=True)
        return norm, e0, e1, (mark() if record else None)

    def note_launch(e0, e1, e2, e3, e4, waited=None):
        if not FUSE:
            ev["bev"].append((e0, e1)); ev["radon"].append((e1, e2))
        if not GROUP_CORR:
            ev["corr"].append((e2, e3, 1))
        if not SIDE_SWEEP:
            ev["sweep"].append((e3, e4, 1))
        if waited is not None:
            ev["wait"].append(waited)

    def step_local(record):
        """N = 1: candidates and the swept database are this GPU's own entries."""
        begin_step()
        for c, (xyz, offs) in enumerate(chunks):
            launch_no[0] += 1
            norm, e0, e1, e2 = descriptors(c, xyz, offs, record)
            db = spec32[db_slot(c)]
            if GROUP_CORR:
                spec = spec32[wslot(c)]                # written by the group's correlation launch
            else:
                spec, spec16, _, _ = ring.spectrum_corr_pairs_db(norm, db, cand_idx[c], out=(out_dist[c], out_ang[c]), spec_out=spec32[wslot(c)])
            e3 = mark() if record else None
            if SIDE_SWEEP:
                if not GROUP_CORR:
                    issue_side_sweeps((c,), record)
            else:
                d, a = ring.corr_sweep_fft(spec[:1], db)   # one new query against the whole database
                torch.min(d, 1, out=(sweep_val[c:c + 1], sweep_row[c:c + 1]))
            if record:
                note_launch(e0, e1, e2, e3, mark())
        if SIDE_SWEEP:
            # lag: the step's last batch of sweeps may still run while the next step starts.  Its launches (>= DEPTH) read only this step's
            # set of slots, which the next step does not write; the step after that waits (here) for later events of the same stream
            if args.sweep_join == "lag" and len(sweep_batches) >= 2 and sweep_batches[-1][0] >= DEPTH:
                torch.cuda.current_stream().wait_event(sweep_batches[-2][1])
            else:
                torch.cuda.current_stream().wait_stream(side)  # every sweep of the step is done before the step ends
            sweep_batches.clear()

    def step_allgather(record):
        """N > 1, replicated database: every launch's new entries go to every rank (fp16 replicas or exact fp32), DEPTH launches ahead of their use."""
        begin_step()
        for c, (xyz, offs) in enumerate(chunks):
            g = launch_no[0]; launch_no[0] += 1
            norm, e0, e1, e2 = descriptors(c, xyz, offs, record)
            ew0 = mark() if record else None
            while len(pending) >= DEPTH:           # the compute stream waits for the exchange of launch g - DEPTH
                pending.pop(0)[0].wait()           # (stream-side wait, the host does not block)
            ew1 = mark() if record else None
            db = gathered[(g - DEPTH) % (DEPTH + 1)]
            # half spectrum of the new descriptors (kept: database entries; fp16 replica for the other ranks) +
            # correlation with their candidates out of the replicated database, one launch
            if REP32:
                db = torch.view_as_complex(db)
            spec, spec16, _, _ = ring.spectrum_corr_pairs_db(norm, db, cand_idx[c], want_f16=not REP32, out=(out_dist[c], out_ang[c]),
                                                             spec_out=spec32[c])
            if REP32:
                spec16 = torch.view_as_real(spec32[c])         # the exact entry itself travels
            e3 = mark() if record else None
            d, a = ring.corr_sweep_fft(spec[:1], db)   # one new query against the whole (replicated) database
            torch.min(d, 1, out=(sweep_val[c:c + 1], sweep_row[c:c + 1]))
            e4 = mark() if record else None
            pending.append((gather_async(gathered[g % (DEPTH + 1)], spec16), spec16))
            if record:
                note_launch(e0, e1, e2, e3, e4, (ew0, ew1))
        if rescorer is not None:
            # exact re-scoring of the candidates whose replica score is within 2e-3 of the acceptance threshold: global row r
            # of launch c's database = descriptor r % B of rank r // B, built in launch c - DEPTH
            slot = torch.tensor([db_slot(c) for c in range(CH)], device=device)
            flat_rows = (slot[:, None].to(torch.int64) * NDB + cand_idx.to(torch.int64)).reshape(-1)
            e32 = spec32.view(-1, 61, 120)

            def exact(qdesc, local_rows):
                return ring.corr_pairs_fft(qdesc.contiguous(), e32[local_rows].contiguous())
            d2, a2 = rescorer.rescore(out_dist.view(-1), out_ang.view(-1), flat_rows, spec32[:CH].reshape(-1, 61, 120),
                                      lambda rows: (rows % NDB) // B, lambda rows: (rows // NDB) * B + rows % B, exact)
            out_dist.view(-1).copy_(d2); out_ang.view(-1).copy_(a2)

    def fetch_rows_async(L):
        """the candidate rows of launch L, requested ahead: -> (work, finish)"""
        if CABI:
            return fetch_plans[L].fetch(spec32[db_slot(L)], async_op=True, stream=comm_stream)
        return fetch_plans[L].fetch(spec32[db_slot(L)], async_op=True)

    def step_fetch(record):
        """N > 1, sharded database: every launch fetches exactly the candidate rows it needs (exact fp32, FETCH_AHEAD launches early) and its
        swept query visits every rank's shard on a side stream."""
        begin_step()
        for c, (xyz, offs) in enumerate(chunks):
            launch_no[0] += 1
            norm, e0, e1, e2 = descriptors(c, xyz, offs, record)
            if c == 0:                             # the first fetches of the step read the slots copied at its start
                for L in range(min(FETCH_AHEAD, CH)):
                    fetch_q[L] = fetch_rows_async(L)
            work, finish = fetch_q.pop(c)
            ew0 = mark() if record else None
            if work is not None:
                work.wait()                        # the compute stream waits for the rows requested FETCH_AHEAD launches ago
            ew1 = mark() if record else None
            rows = finish()                        # [B,61,120] complex64: the candidates' exact entries, in request order
            last_fetched[0] = rows
            spec, spec16, _, _ = ring.spectrum_corr_pairs_db(norm, rows, ident_idx, out=(out_dist[c], out_ang[c]), spec_out=spec32[c])
            e3 = mark() if record else None
            # the rows of launch c + FETCH_AHEAD: slot c + FETCH_AHEAD - DEPTH <= c is written on every owner by now
            L = c + FETCH_AHEAD
            if L < CH:
                fetch_q[L] = fetch_rows_async(L)
            # this launch's query to every rank; the sharded top-1 sweep of the PREVIOUS launch's queries on the side stream
            done = torch.cuda.Event(); done.record()
            with torch.cuda.stream(side):
                side.wait_event(done)
                qw = gather_async(q_all[c % 2], torch.view_as_real(spec32[c, :1]).contiguous(), stream=side)
                sweep_pending.append((c, qw))
                if len(sweep_pending) > 1:
                    run_sharded_sweep(*sweep_pending.pop(0))
            if record:
                note_launch(e0, e1, e2, e3, mark(), (ew0, ew1))
        with torch.cuda.stream(side):              # the last launch's sweep; the compute stream joins the side stream at the step's end
            while sweep_pending:
                run_sharded_sweep(*sweep_pending.pop(0))
        torch.cuda.current_stream().wait_stream(side)

    step = {None: step_local, "allgather": step_allgather, "fetch": step_fetch}[EXCH]

    def fence():
        if dist_on:
            while pending:
                pending.pop(0)[0].wait()
            if EXCH == "fetch":
                torch.cuda.current_stream().wait_stream(side)
            if CABI:
                torch.cuda.current_stream().wait_stream(comm_stream)
            dist.barrier()
        if SIDE_SWEEP:
            torch.cuda.current_stream().wait_stream(side)
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

    S = types.SimpleNamespace(**locals())
    S.__dict__.update(after_timed_region(S))
    line = result_line(S) if rank == 0 else None
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        emit(line, args.detail_file)


if __name__ == "__main__":
    main()
