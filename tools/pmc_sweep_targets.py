"""Database sweeps for the rocprofv3 --pmc passes of tools/run_pmc_sweep.sh: 10 000-entry RING database with 1 query, 10 016 entries with 4 queries
(different grid sizes so that the two show up as separate rows), RING++ (6 channels) with 1 and 4 queries."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mr_slam_amd import ring

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
norm = torch.randn((256, 120, 120), generator=g, device=dev)
spec = ring.half_spectrum(norm)
for n_db, nq in ((10000, 1), (10016, 4)):
    db = spec[torch.arange(n_db, device=dev) % 256].contiguous()
    q = spec[:nq].contiguous()
    for _ in range(3):
        ring.corr_sweep_fft(q, db)
db6 = torch.stack([spec[torch.arange(2000, device=dev) % 256].roll(k, 0) for k in range(6)], 1).contiguous()
for n_db, nq in ((2000, 1), (1984, 4)):
    for _ in range(3):
        ring.corr_sweep_fft(db6[:nq].contiguous(), db6[:n_db].contiguous())
torch.cuda.synchronize()
print("sweep targets done")
