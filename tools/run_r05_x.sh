#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gicp_gpu.py tests/test_pointfeat_gpu.py tests/test_cpp_adapter.py tests/test_pybind_pygicp.py -m gpu -q -x -v > $OUT/x_tests.txt 2>&1
grep -n "PASSED\|FAILED\|Fatal\|fault\|Error\|error" $OUT/x_tests.txt | tail -n 12 | cut -c1-250
grep -n "File \"/root/repo\|File \"tests\|File \".*mr_slam" $OUT/x_tests.txt | head -n 12 | cut -c1-200
