#!/bin/bash
# round 4, first GPU call: the new search core -- parity tests, A/B timing against the round-3 core, kernel stats
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gicp_gpu.py tests/test_pointfeat_gpu.py tests/test_pybind_pygicp.py -m gpu -x -q -k "not timed_protocol" > $OUT/pytest_nn.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_nn.log; tail -n 15 $OUT/pytest_nn.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nnstats -- python $R/tools/quick_nn.py 64 --feat > $OUT/quick_nn.log 2>&1; echo "quick_nn rc $?"
for f in $(find $OUT/nnstats -name '*kernel_stats.csv'); do cp $f $OUT/nn_kernel_stats.csv; done; rm -rf $OUT/nnstats
grep -v "^\[" $OUT/quick_nn.log | tail -n 12
MRS_DEV=1 MRS_NN_CORE=0 timeout 300 python $R/tools/quick_nn.py 2 --feat 2>&1 | grep "^feat"
grep -E "k_nn_scan|k_knn|k_cov_from|k_feat_from|k_linearize|k_lm_update|k_leaf|k_group_boxes|k_boxes|RadixSort|k_fitness" $OUT/nn_kernel_stats.csv | cut -c1-60,150-400 | head -40
