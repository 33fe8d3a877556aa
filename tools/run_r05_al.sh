#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_bench_contract_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|rror" | tail -n 6 | cut -c1-300
