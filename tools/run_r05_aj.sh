#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 300 python tools/quick_knn.py 2>&1 | tail -n 1
MRS_DEV=1 MRS_KNN_DBG=1 timeout 300 python tools/quick_knn.py --dbg 2>&1 | grep -E "knn dbg" | head -n 2
