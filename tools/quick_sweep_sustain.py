#!/usr/bin/env python3
"""Development: the one-query tiled sweep in bursts (3 + 10 launches after a 50 ms pause) and sustained (30 + 100 launches), 10 000 and 25 000 entries."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mr_slam_amd import ring
dev = "cuda:0"
def ev_ms(fn, reps, warm):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
g = torch.Generator(device=dev).manual_seed(1)
sino = torch.rand((2048, 120, 120), device=dev, generator=g) * (torch.rand((2048, 120, 120), device=dev, generator=g) < 0.3)
pool = ring.half_spectrum(ring.normalize(sino[:, None])[:, 0]).contiguous()
for n in (10000, 25000):
    db = pool[torch.arange(n, device=dev) % 2048].contiguous()
    tiled = ring.spec_to_tiled(db); del db
    q = pool[:1].contiguous()
    f = lambda: ring.corr_sweep_fft_tiled(q, tiled)
    out = []
    for _ in range(4):
        time.sleep(0.05)
        out.append(n / ev_ms(f, 10, 3) / 1e3)
    sus = n / ev_ms(f, 100, 30) / 1e3
    out2 = []
    for _ in range(2):
        time.sleep(0.05)
        out2.append(n / ev_ms(f, 10, 3) / 1e3)
    print(f"n={n}: bursts {[round(x,1) for x in out]} M pairs/s, sustained {sus:.1f}, bursts again {[round(x,1) for x in out2]}; entry bytes 58624 -> burst {out[-1]*58624/1e3/8000:.3f} sustained {sus*58624/1e3/8000:.3f} of 8 TB/s", flush=True)
    del tiled
