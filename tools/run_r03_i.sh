#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd $R
for P in 2 1; do
MRS_NN_P=$P timeout 600 python - <<PY 2>&1 | grep -v amdgpu.ids
import json, bench, os
r = bench.gicp_leg(0, 0, 256, 20)
print("P", os.environ["MRS_NN_P"], "forced it/s %.0f cold %.0f natural pairs/s %.0f cov_s %.4f" % (r["iters_per_s"], r["cold"]["iters_per_s"], r["natural"]["pairs_per_s"], r["covariance_s"]))
PY
done
timeout 900 python -m pytest tests/test_bev_gpu.py tests/test_ref_pins_gpu.py tests/test_pointfeat_gpu.py tests/test_gicp_gpu.py -x -q -m gpu 2>&1 | tail -n 3
