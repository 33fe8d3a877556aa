#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export MRS_DEV=1 MRS_KNN_DBG=1 MRS_KNN_TRACE_FILE=/tmp/knn_trace.bin
timeout 200 python tools/knn_trace.py 8 2>&1 | grep -v "knn dbg" | tail -n 16
timeout 200 python tools/knn_trace.py 64 2>&1 | grep -v "knn dbg" | tail -n 16
