#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_ring_gpu.py tests/test_fused_gpu.py tests/test_pybind_radon_backend.py tests/test_ref_pins_gpu.py -x -q -m gpu 2>&1 | tail -n 3
timeout 200 python tools/quick_radon.py 2048 2>&1 | grep -v amdgpu.ids
timeout 300 python - <<PY 2>&1 | grep -v amdgpu.ids
import torch, sys
sys.path.insert(0, '.')
from mr_slam_amd import ring
g = torch.Generator(device="cuda:0").manual_seed(0)
img = (torch.rand((2048, 120, 120), device="cuda:0", generator=g) * (torch.rand((2048, 120, 120), device="cuda:0", generator=g) < 0.12)).contiguous()
plan = ring.ring_plan(0)
res = {}
for v in (1, 0):
    plan.set_option(plan.OPT_FUSED_VARIANT, v)
    for _ in range(3): out = plan.forward(img, raw=True, normalized=True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): out = plan.forward(img, raw=False, normalized=True)
    b.record(); torch.cuda.synchronize()
    res[v] = plan.forward(img, raw=True, normalized=True)
    print("variant", v, "radon 2048 images: %.4f ms per 1024" % (a.elapsed_time(b) / 10 / 2))
print("same bits:", torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]))
plan.set_option(plan.OPT_FUSED_VARIANT, 1)
PY
