#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
timeout 300 python tools/quick_sweep_dma.py 1 5 100 1000 10000 25003 > $OUT/sweep_dma_a.log 2>&1; echo "rc $?" >> $OUT/sweep_dma_a.log
cat $OUT/sweep_dma_a.log | grep -v '^\[mrslam\]' | tail -n 60
