#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
( export MRS_DEV=1 MRS_KNN_DBG=1 MRS_NN_TRACE_FILE=/tmp/nn_trace.bin
timeout 300 python tools/nn_trace.py 256 2>&1 | grep -v "knn dbg\|mrslam\|amdgpu.ids" | tail -n 10
timeout 300 python tools/nn_trace.py 32 2>&1 | grep -v "knn dbg\|mrslam\|amdgpu.ids" | tail -n 10 )
timeout 300 python tools/quick_nn.py 256 2>&1 | grep -E "^(1|0) " | cut -c1-600
