#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export MRS_DEV=1 MRS_KNN_DBG=1 MRS_KNN_TRACE_FILE=/tmp/knn_trace.bin
timeout 200 python tools/knn_trace_single.py 2>&1 | grep -v "mrslam\|amdgpu.ids" | tail -n 6
