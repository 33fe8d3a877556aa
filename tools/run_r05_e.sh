#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
SWEEP_VARIANTS=0,1008,1001008,1011008,1001108,1011108,1001012,1011012,1001112,1011112,1010112,1010008 timeout 400 python tools/quick_sweep_dma.py 10000 10240 25003 > $OUT/sweep_dma_e.log 2>&1; echo "rc $?" >> $OUT/sweep_dma_e.log
grep -v '^\[mrslam\]\|^{' $OUT/sweep_dma_e.log | tail -n 60
