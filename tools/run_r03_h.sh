#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gicp_gpu.py tests/test_pointfeat_gpu.py tests/test_ref_pins_gpu.py -x -q -m gpu > $OUT/pytest_knn.log 2>&1; tail -n 5 $OUT/pytest_knn.log
timeout 300 python tools/quick_knn.py 128 2>&1 | grep -v amdgpu.ids | tee $OUT/quick_knn.log
timeout 600 python - <<PY 2>&1 | grep -v amdgpu.ids | tee $OUT/gicp_leg.json
import json, bench
r = bench.gicp_leg(0, 0, 256, 20)
r.pop("bound", None); r.pop("nn_search", None)
print(json.dumps(r))
PY
