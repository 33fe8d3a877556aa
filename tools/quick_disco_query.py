import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from mr_slam_amd import node
g = torch.Generator(device="cuda:0").manual_seed(3)
n=10000
sig = torch.rand((n, 1024), generator=g, device="cuda:0")
spec = torch.view_as_complex(torch.randn((n, 1, 40, 120, 2), generator=g, device="cuda:0"))
db = node.DiscoDatabase(capacity=n)
for i in range(n): db.append(sig[i], spec[i])
qs, qf = (sig[0]+0.01).contiguous(), spec[0].contiguous()
qs_h, qf_h = qs.cpu().numpy(), qf.cpu()
for _ in range(5): db.query(qs, qf); db.query(qs_h, qf_h)
for name,a,b in (("device",qs,qf),("host",qs_h,qf_h)):
    t0=time.perf_counter()
    for _ in range(50): db.query(a,b)
    print(name, (time.perf_counter()-t0)/50*1e3, "ms")
