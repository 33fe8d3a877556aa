#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export MRS_DEV=1 MRS_KNN_DBG=1 MRS_NN_TRACE_FILE=/tmp/nn_trace.bin MRS_NN_TRACE_KERNEL=4
timeout 300 python tools/nn_trace.py 256 2>&1 | grep -v "knn dbg\|mrslam\|amdgpu.ids" | tail -n 10
timeout 300 python tools/nn_trace.py 32 2>&1 | grep -v "knn dbg\|mrslam\|amdgpu.ids" | tail -n 10
