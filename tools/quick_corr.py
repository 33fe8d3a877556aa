"""Quick correlation timings (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mr_slam_amd import ring

dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = torch.Generator(device=dev).manual_seed(0)
db = torch.randn((N, 1, 120, 120), device=dev, generator=g)
db = ring.normalize(db)
q = db[:8].contiguous()
sdb = ring.half_spectrum(db[:, 0]); sq = sdb[:8].contiguous()


def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for nq in (1, 8):
    ms = timeit(lambda: ring.corr_sweep(q[:nq], db))
    print(f"direct sweep  nq={nq} N={N}: {ms:.3f} ms  {nq*N/ms/1e3:.2f} Mpairs/s  {nq*N*57600/ms/1e6:.0f} GB/s")
    ms = timeit(lambda: ring.corr_sweep_fft(sq[:nq], sdb))
    print(f"fft sweep     nq={nq} N={N}: {ms:.3f} ms  {nq*N/ms/1e3:.2f} Mpairs/s  {nq*N*58560/ms/1e6:.0f} GB/s")
P = 512
ms = timeit(lambda: ring.corr_pairs(db[:P], db[P:2 * P]))
print(f"direct pairs P={P}: {ms:.3f} ms")
ms1 = timeit(lambda: ring.half_spectrum(db[:P, 0]))
ms2 = timeit(lambda: ring.corr_pairs_fft(sdb[:P], sdb[P:2 * P]))
print(f"fft pairs P={P}: spectrum {ms1:.3f} ms + corr {ms2:.3f} ms")

# RING++ (6 channels)
C = 6
Np = max(64, N // 8)
dbp = ring.normalize(torch.randn((Np, C, 120, 120), device=dev, generator=g))
sdbp = ring.half_spectrum(dbp)
for nq in (1, 4):
    ms = timeit(lambda: ring.corr_sweep(dbp[:nq].contiguous(), dbp), n=2)
    print(f"RING++ direct sweep nq={nq} N={Np}: {ms:.3f} ms  {nq*Np/ms/1e3:.3f} Mpairs/s")
    ms = timeit(lambda: ring.corr_sweep_fft(sdbp[:nq].contiguous(), sdbp))
    print(f"RING++ fft sweep    nq={nq} N={Np}: {ms:.3f} ms  {nq*Np/ms/1e3:.3f} Mpairs/s  {nq*Np*C*58560/ms/1e6:.0f} GB/s")

# fp16 replicas
_, sdb16 = ring.half_spectrum_f16(db[:, 0])
for nq in (1, 8):
    ms = timeit(lambda: ring.corr_sweep_fft(sq[:nq], sdb16))
    print(f"fft sweep f16 nq={nq} N={N}: {ms:.3f} ms  {nq*N/ms/1e3:.2f} Mpairs/s  {nq*N*29280/ms/1e6:.0f} GB/s")
