#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_gicp_gpu.py tests/test_pointfeat_gpu.py -m gpu -q -x -k "knn or covariances or pointfeat or feature or ringplusplus or cached" 2>&1 | tail -n 4 | cut -c1-300
timeout 200 python tools/quick_knn_scale.py 2>&1 | tail -n 1
timeout 300 python tools/quick_knn.py 2>&1 | tail -n 1
MRS_DEV=1 MRS_KNN_WAVES=4 timeout 300 python tools/quick_knn.py 2>&1 | tail -n 1
MRS_DEV=1 MRS_KNN_WAVES=5 timeout 300 python tools/quick_knn.py 2>&1 | tail -n 1
MRS_DEV=1 MRS_KNN_DBG=1 timeout 300 python tools/quick_knn.py --dbg 2>&1 | grep -E "knn dbg" | tail -n 2
