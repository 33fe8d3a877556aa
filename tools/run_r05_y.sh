#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_gicp_gpu.py tests/test_pointfeat_gpu.py -m gpu -q -x -k "knn or covariances or pointfeat or feature or ringplusplus or cached or degenerate" 2>&1 | tail -n 3 | cut -c1-300
bash tools/run_r05_s.sh 2>&1 | tail -n 14
