"""One-query RING database sweep: the LDS-DMA variants against the register-staged kernel (development aid).
python tools/quick_sweep_dma.py [entries ...]   (needs MRS_DEV=1: the variant is a development switch)"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MRS_DEV"] = "1"
import torch
from mr_slam_amd import ring

dev = "cuda:0"
sizes = [int(a) for a in sys.argv[1:]] or [10000]
VARIANTS = [int(v) for v in os.environ.get("SWEEP_VARIANTS", "0,8,12,108,112,1008,1012,1108,1112").split(",")]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


res = {}
ROUNDS = int(os.environ.get("SWEEP_ROUNDS", "5"))
for N in sizes:
    g = torch.Generator(device=dev).manual_seed(N)
    db = ring.normalize(torch.randn((N, 1, 120, 120), device=dev, generator=g))
    sdb = ring.half_spectrum(db[:, 0])
    q = sdb[N // 3:N // 3 + 1].contiguous() if N > 3 else sdb[:1].contiguous()
    del db
    tiled = ring.spec_to_tiled(sdb)
    want = None
    sweeps, ok = {}, {}
    for v in VARIANTS:
        is_tiled = v >= 1000000

        def sweep(v=v, is_tiled=is_tiled):
            os.environ["MRS_SWEEP_VARIANT"] = str(v % 1000000)
            return ring.corr_sweep_fft_tiled(q, tiled) if is_tiled else ring.corr_sweep_fft(q, sdb)
        d, a = sweep()
        d, a = d.reshape(1, -1), a.reshape(1, -1)
        torch.cuda.synchronize()
        if want is None:
            want = (d.clone(), a.clone())
        ok[v] = int((d != want[0]).sum() + (a != want[1]).sum())
        sweeps[v] = sweep
    times = {v: [] for v in VARIANTS}
    if N >= 256:
        for r in range(ROUNDS):                      # variants interleaved: clock and box state are shared fairly
            for v in (VARIANTS if r % 2 == 0 else VARIANTS[::-1]):
                times[v].append(timeit(sweeps[v], 10))
    for v in VARIANTS:
        t = sorted(times[v]) or [float("nan")]
        med, best = t[len(t) // 2], t[0]
        res[f"N={N} v={v}"] = {"us_median": round(med * 1e3, 1), "us_min": round(best * 1e3, 1), "Mpairs_s": round(N / med / 1e3, 1),
                               "TB_s": round(N * 58560 / med / 1e9, 3), "mismatches": ok[v]}
        print(f"N={N:6d} variant={v:8d}  median {med*1e3:7.1f} us  min {best*1e3:7.1f} us  {N/med/1e3:7.1f} Mpairs/s  {N*58560/med/1e9:6.3f} TB/s ({N*58560/med/8e9:.3f} of 8)  mismatches={ok[v]}", flush=True)
    del sdb, tiled
print(json.dumps(res))
