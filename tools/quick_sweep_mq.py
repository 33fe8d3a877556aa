#!/usr/bin/env python3
"""Development timing of the several-queries-per-sweep forms (round 6): 10 000-entry RING / RING++ databases, Q = 1, 2, 3, 4, 8, 16 queries per call,
DMA pipeline (tiled and row layout) against the register-staged k_ring_corr_fft / k_ring_sweep_mc (MRS_DEV=1 MRS_SWEEP_MQ_VARIANT=0).
  python tools/quick_sweep_mq.py [n_db]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mr_slam_amd import ring  # noqa: E402


def ev_ms(fn, reps=10, warm=3):
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


def main():
    n_db = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(1)
    sino = torch.rand((512, 120, 120), device=dev, generator=g) * (torch.rand((512, 120, 120), device=dev, generator=g) < 0.3)
    pool = ring.half_spectrum(ring.normalize(sino[:, None])[:, 0]).contiguous()
    db = pool[torch.arange(n_db, device=dev) % 512].contiguous()
    tiled = ring.spec_to_tiled(db)
    tag = os.environ.get("MRS_SWEEP_MQ_VARIANT", "default")
    for nq in (1, 2, 3, 4, 8, 16):
        q = pool[:nq].contiguous()
        ms_t = ev_ms(lambda: ring.corr_sweep_fft_tiled_q(q, tiled))
        ms_r = ev_ms(lambda: ring.corr_sweep_fft(q, db))
        print(f"ring   variant={tag} nq={nq:2d} tiled {ms_t * 1e3:8.1f} us {nq * n_db / ms_t / 1e3:7.1f} M pairs/s | row {ms_r * 1e3:8.1f} us {nq * n_db / ms_r / 1e3:7.1f} M pairs/s", flush=True)
    del tiled
    db6 = torch.stack([db.roll(k, 0) for k in range(6)], 1).contiguous()
    tiled6 = ring.spec_to_tiled(db6)
    for nq in (1, 2, 4, 8):
        q = db6[:nq].contiguous()
        ms_t = ev_ms(lambda: ring.corr_sweep_fft_tiled_q(q, tiled6), reps=3, warm=1)
        ms_r = ev_ms(lambda: ring.corr_sweep_fft(q, db6), reps=3, warm=1)
        print(f"ringpp variant={tag} nq={nq:2d} tiled {ms_t * 1e3:8.1f} us {nq * n_db / ms_t / 1e3:7.2f} M pairs/s | row {ms_r * 1e3:8.1f} us {nq * n_db / ms_r / 1e3:7.2f} M pairs/s", flush=True)


if __name__ == "__main__":
    main()
