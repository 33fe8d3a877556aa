"""RING database sweep timings (development aid): python tools/quick_sweep.py [entries]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mr_slam_amd import ring

dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
g = torch.Generator(device=dev).manual_seed(0)
db = ring.normalize(torch.randn((N, 1, 120, 120), device=dev, generator=g))
sdb = ring.half_spectrum(db[:, 0]); sq = sdb[:8].contiguous()
_, sdb16 = ring.half_spectrum_f16(db[:, 0])
del db


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


import hashlib
d, a = ring.corr_sweep_fft(sq[:4], sdb)[:2]
digest = hashlib.sha1(d.cpu().numpy().tobytes() + a.cpu().numpy().tobytes()).hexdigest()[:12]      # same bits across variants?
out = [f"sha {digest}"]
for nq in (1, 2, 4, 8):
    ms = timeit(lambda: ring.corr_sweep_fft(sq[:nq], sdb))
    ms16 = timeit(lambda: ring.corr_sweep_fft(sq[:nq], sdb16))
    out.append(f"nq={nq}: f32 {ms*1e3:.0f} us {nq*N/ms/1e3:.1f} Mp/s {N*58560/ms/1e6:.0f} GB/s | f16 {ms16*1e3:.0f} us {nq*N/ms16/1e3:.1f} Mp/s {N*29280/ms16/1e6:.0f} GB/s")
print(f"variant={os.environ.get('MRS_SWEEP_VARIANT','0')} blocks={os.environ.get('MRS_SWEEP_BLOCKS','2')} N={N}  " + "  ||  ".join(out))
