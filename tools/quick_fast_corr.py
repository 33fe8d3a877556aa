import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from mr_slam_amd import ring
g = torch.Generator(device="cuda:0").manual_seed(0)
sino = torch.rand((8,120,120), device="cuda:0", generator=g)
half = ring.half_spectrum(ring.normalize(sino[:,None])[:,0])
pool = torch.cat([half, half[:, 1:60].flip(1).conj()], 1).contiguous()
a, b = pool[0:1].contiguous(), pool[1:2].contiguous()
ha, hb = half[0:1].contiguous(), half[1:2].contiguous()
def ev(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
def wall(fn, n=500):
    for _ in range(10): fn()
    t0=time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter()-t0)/n*1e6
out = torch.empty(2, dtype=torch.float32, device="cuda:0")
print("full spectra kernel, device args, no readback: us (events)", ev(lambda: ring.fast_corr.__wrapped__(a,b) if hasattr(ring.fast_corr,"__wrapped__") else None) if False else "")
import ctypes as C
from mr_slam_amd import _lib
L=_lib.load(); ctx=_lib.ctx(0)
def k_full():
    L.mrs_ring_corr_spectra(ctx, _lib.ptr(torch.view_as_real(a)), _lib.ptr(torch.view_as_real(b)), 1, 1, 120, 120, C.c_void_p(out.data_ptr()), C.c_void_p(out.data_ptr()+4), None, _lib.current_stream(0))
def k_half():
    ring.corr_pairs_fft(ha, hb, out=(out[0:1], out[1:2].view(torch.int32)))
print("full kernel events us", ev(k_full), "wall launch-only us", wall(k_full))
print("half kernel events us", ev(k_half), "wall launch-only us", wall(k_half))
print("full + readback wall us", wall(lambda: (k_full(), out.cpu())))
print("half + readback wall us", wall(lambda: (k_half(), out.cpu())))
ah, bh = a.cpu(), b.cpu()
print("fast_corr cached wall us", wall(lambda: ring.fast_corr(ah, bh)))
