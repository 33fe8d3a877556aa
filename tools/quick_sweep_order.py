import os, sys, torch
sys.path.insert(0, ".")
from mr_slam_amd import ring
dev="cuda:0"
def ev_ms(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
g = torch.Generator(device=dev).manual_seed(1)
big = torch.empty((60 << 30,), dtype=torch.uint8, device=dev) if len(sys.argv) > 1 and sys.argv[1] == "big" else None
n_pool = 12000
sino = torch.rand((n_pool, 120, 120), device=dev, generator=g) * (torch.rand((n_pool, 120, 120), device=dev, generator=g) < 0.3)
pool = ring.half_spectrum(ring.normalize(sino[:, None])[:, 0]).contiguous()
del sino
db = pool[:10000].contiguous()
tiled = ring.spec_to_tiled(db)
q = pool[10000:10004].contiguous()
for rep in range(3):
    t = ev_ms(lambda: ring.corr_sweep_fft_tiled_q(q, tiled)); r = ev_ms(lambda: ring.corr_sweep_fft(q, db))
    print(f"rep {rep}: tiled {4e4/t/1e3:.1f} M  row {4e4/r/1e3:.1f} M", flush=True)
for rep in range(2):
    r = ev_ms(lambda: ring.corr_sweep_fft(q, db)); t = ev_ms(lambda: ring.corr_sweep_fft_tiled_q(q, tiled))
    print(f"swapped {rep}: tiled {4e4/t/1e3:.1f} M  row {4e4/r/1e3:.1f} M", flush=True)
t = ev_ms(lambda: ring.corr_sweep_fft_tiled_q(q, tiled), reps=50, warm=10); r = ev_ms(lambda: ring.corr_sweep_fft(q, db), reps=50, warm=10)
print(f"50 reps: tiled {4e4/t/1e3:.1f} M  row {4e4/r/1e3:.1f} M")
