#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
SWEEP_VARIANTS=0,1008,1001008,1011008,1002008,1003008,1004008,1005008,1001108,1011108,1001012,1011012,1001112,1011112,1004112,1005112,1001008 timeout 300 python tools/quick_sweep_dma.py 10000 25003 > $OUT/sweep_dma_d.log 2>&1; echo "rc $?" >> $OUT/sweep_dma_d.log
grep -v '^\[mrslam\]\|^{' $OUT/sweep_dma_d.log | tail -n 60
