#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gicp_gpu.py tests/test_pointfeat_gpu.py -m gpu -q -x -k "knn or covariances or pointfeat or feature or ringplusplus or cached or submap or identical" 2>&1 | tail -n 8 | cut -c1-300
timeout 300 python tools/quick_knn.py 2>&1 | tail -n 3
MRS_DEV=1 MRS_KNN_DBG=1 timeout 300 python tools/quick_knn.py --dbg 2>&1 | grep -E "knn dbg|feat_ms" | tail -n 6
