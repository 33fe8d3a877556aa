"""One-query RING sweeps for the rocprofv3 --pmc passes of tools/run_r05_b.sh: 10 000 entries, the register-staged kernel and the LDS-DMA
variants named in SWEEP_VARIANTS (three launches each; every variant is its own kernel instantiation, so its own row)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MRS_DEV"] = "1"
import torch
from mr_slam_amd import ring

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
N = 10000
db = ring.normalize(torch.randn((N, 1, 120, 120), device=dev, generator=g))
sdb = ring.half_spectrum(db[:, 0])
del db
q = sdb[7:8].contiguous()
for v in os.environ.get("SWEEP_VARIANTS", "0,1108,1112").split(","):
    os.environ["MRS_SWEEP_VARIANT"] = v
    for _ in range(3):
        ring.corr_sweep_fft(q, sdb)
    torch.cuda.synchronize()
print("sweep targets done")
