"""Several-queries-per-sweep launches for rocprofv3 --pmc passes (round 6): NQ queries (environment) against a 10 000-entry DMA-tiled RING database,
three launches of mrs_ring_corr_fft_sweep_tiled_q; MRS_SWEEP_MQ_VARIANT selects the pipeline variant (development switch)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MRS_DEV"] = "1"
import torch
from mr_slam_amd import ring

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
N = 10000
nq = int(os.environ.get("NQ", "4"))
db = ring.normalize(torch.randn((N, 1, 120, 120), device=dev, generator=g))
sdb = ring.half_spectrum(db[:, 0])
del db
tiled = ring.spec_to_tiled(sdb)
q = sdb[7:7 + nq].contiguous()
for _ in range(3):
    ring.corr_sweep_fft_tiled_q(q, tiled)
torch.cuda.synchronize()
print("mq targets done", nq)
