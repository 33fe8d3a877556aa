#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gicp_gpu.py tests/test_cpp_adapter.py tests/test_pybind_pygicp.py tests/test_threads_gpu.py tests/test_reference_dropin_gpu.py -m gpu -q -x 2>&1 | tail -n 3 | cut -c1-300
timeout 200 python tools/quick_single_pair.py 2>&1 | tail -n 1
