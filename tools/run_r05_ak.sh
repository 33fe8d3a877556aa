#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_pointfeat_gpu.py tests/test_node_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error" | tail -n 5 | cut -c1-300
timeout 300 python tools/quick_knn.py 2>&1 | tail -n 1 | cut -c1-300
