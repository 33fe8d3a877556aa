#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_node_gpu.py tests/test_ring_gpu.py tests/test_ref_pins_gpu.py -m gpu -x -q > $OUT/pytest_f.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_f.log; tail -n 30 $OUT/pytest_f.log
timeout 300 python tools/quick_sweep_mc.py 10000 > $OUT/sweep_mc_f.log 2>&1; grep -v mrslam $OUT/sweep_mc_f.log | tail -n 12
