#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_node_gpu.py -m gpu -q -x 2>&1 | tail -n 3 | cut -c1-300
timeout 300 python tools/quick_loopdb.py 2>&1 | tail -n 4 | cut -c1-300
