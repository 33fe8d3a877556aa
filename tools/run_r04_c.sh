#!/bin/bash
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gicp_gpu.py tests/test_pointfeat_gpu.py -m gpu -x -q -k "not timed_protocol and not beyond_the_ordered" > $OUT/pytest_nn.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_nn.log; tail -n 6 $OUT/pytest_nn.log
MRS_DEV=1 MRS_NN_PROF=1 timeout 300 python tools/nn_prof.py 64 2>&1 | grep "nn prof" | cut -c1-600
timeout 600 python $R/tools/quick_nn.py ${2:-64} --feat > $OUT/quick_nn.log 2>&1; echo "quick_nn rc $?"
grep -E "^(0|1|feat) " $OUT/quick_nn.log | cut -c1-420
