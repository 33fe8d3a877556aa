#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 400 python -m pytest tests/test_gicp_gpu.py tests/test_pointfeat_gpu.py -m gpu -q -x -k "knn or covariances or identical or fused_knn or chunks or correspondences" 2>&1 | grep -E "passed|failed" | tail -n 1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
