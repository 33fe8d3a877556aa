"""Blocking-call times of the loop databases at 10 000 entries (development aid): DiSCO query with device and with host arguments, RING query from
the node's host tensor."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mr_slam_amd import node
dev = "cuda:0"; n_db = 10000
g = torch.Generator(device=dev).manual_seed(3)
sig_db = torch.rand((n_db, 1024), generator=g, device=dev)
spec_db = torch.view_as_complex(torch.randn((n_db, 1, 40, 120, 2), generator=g, device=dev))
ddb = node.DiscoDatabase(capacity=n_db)
for i in range(n_db):
    ddb.append(sig_db[i], spec_db[i])
def lat(fn, reps=50):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return 1e3 * (time.perf_counter() - t0) / reps
qs, qf = (sig_db[7] + 0.01).contiguous(), spec_db[7].contiguous()
print("disco device args ms", round(lat(lambda: ddb.query(qs, qf)), 4), ddb.query(qs, qf))
hs, hf = qs.cpu(), qf.cpu()
print("disco host args ms  ", round(lat(lambda: ddb.query(hs, hf)), 4), ddb.query(hs, hf))
tir = torch.rand((n_db, 120, 120), generator=g, device=dev)
rdb = node.LoopDatabase("ring", capacity=n_db)
for i in range(n_db):
    rdb.append(tir[i])
q = tir[5].cpu()
r = rdb.query(q, 1e9)
print("ring host tensor ms ", round(lat(lambda: rdb.query(q, 0.1)), 4), "pairs/s", round(n_db / lat(lambda: rdb.query(q, 0.1)) * 1e3 / 1e6, 1), "M", [x[:2] for x in r[:3]] if isinstance(r, tuple) else type(r))
