#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gicp_gpu.py tests/test_pointfeat_gpu.py tests/test_ref_pins_gpu.py -x -q -m gpu > $OUT/pytest_knn.log 2>&1; tail -n 15 $OUT/pytest_knn.log
timeout 300 python tools/quick_knn.py 128 2>&1 | grep -v amdgpu.ids | tee $OUT/quick_knn.log
timeout 600 python tools/ab_fused.py --chunks 16 > $OUT/ab_fused.log 2>&1; grep -E "identical|stagger_us.: 70|separate|best" $OUT/ab_fused.log
