"""One-query RING++ (6-channel) database sweep: the LDS-DMA kernels (row layout / tiled) against the channel-outer register kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MRS_DEV"] = "1"
import torch
from mr_slam_amd import ring
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
g = torch.Generator(device=dev).manual_seed(1)
base = ring.half_spectrum(ring.normalize(torch.randn((512, 6, 120, 120), device=dev, generator=g)))      # [512,6,61,120]
db = base[torch.arange(N, device=dev) % 512].contiguous()
db[:, 0, 0, 0] += torch.arange(N, device=dev).to(torch.complex64) * 1e-3                                   # entries differ
q = base[5:6].contiguous()
tiled = ring.spec_to_tiled(db)
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
os.environ["MRS_SWEEP_VARIANT"] = "0"
wd, wa = ring.corr_sweep_fft(q, db)
os.environ["MRS_SWEEP_VARIANT"] = "11008"
d1, a1 = ring.corr_sweep_fft(q, db)
d2, a2 = ring.corr_sweep_fft_tiled(q, tiled)
print("row-layout DMA == channel-outer:", bool(torch.equal(d1, wd) and torch.equal(a1, wa)), " tiled == :", bool(torch.equal(d2, wd[0]) and torch.equal(a2, wa[0])))
for r in range(3):
    os.environ["MRS_SWEEP_VARIANT"] = "0"
    t0 = timeit(lambda: ring.corr_sweep_fft(q, db))
    os.environ["MRS_SWEEP_VARIANT"] = "11008"
    t1 = timeit(lambda: ring.corr_sweep_fft(q, db))
    t2 = timeit(lambda: ring.corr_sweep_fft_tiled(q, tiled))
    for name, t in (("channel-outer", t0), ("dma row", t1), ("dma tiled", t2)):
        print(f"N={N} {name:14s} {t*1e3:8.1f} us  {N/t/1e3:6.2f} Mpairs/s  {N*351360/t/1e9:6.3f} TB/s ({N*351360/t/8e9:.3f} of 8)")
