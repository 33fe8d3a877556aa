#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
SWEEP_VARIANTS=0,1112,1001112,1001108,1001012,1001008,1000112,1000108,1000012,1000008 timeout 300 python tools/quick_sweep_dma.py 1 7 300 10000 25003 > $OUT/sweep_dma_c.log 2>&1; echo "rc $?" >> $OUT/sweep_dma_c.log
grep -v '^\[mrslam\]\|^{' $OUT/sweep_dma_c.log | tail -n 60
