#!/bin/bash
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_ring_gpu.py -x -q -m gpu > $OUT/pytest_fused.log 2>&1; tail -n 4 $OUT/pytest_fused.log
timeout 600 python tools/ab_fused.py --chunks 24 > $OUT/ab_fused.log 2>&1; tail -n 45 $OUT/ab_fused.log; cp gpurun_out/ab_fused.json $OUT/ab_fused.json
