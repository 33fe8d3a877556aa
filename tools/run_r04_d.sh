#!/bin/bash
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gicp_gpu.py tests/test_pointfeat_gpu.py tests/test_pybind_pygicp.py tests/test_cpp_adapter.py -m gpu -x -q > $OUT/pytest_nn.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_nn.log; tail -n 6 $OUT/pytest_nn.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nnstats -- python $R/tools/quick_nn.py 256 > $OUT/quick_nn256.log 2>&1; echo "quick_nn rc $?"
for f in $(find $OUT/nnstats -name '*kernel_stats.csv'); do cp $f $OUT/nn_kernel_stats256.csv; done; rm -rf $OUT/nnstats
grep -E "^(0|1|2|3|feat) " $OUT/quick_nn256.log | cut -c1-460
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/nn_kernel_stats256.csv")))
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("k_nn_","k_knn","k_cov_from","k_feat_from","k_linearize","k_lm_update","k_fitness")):
        print(n.split("(")[0][-44:], r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3,1), "min", round(float(r["MinNs"])/1e3,1), "max", round(float(r["MaxNs"])/1e3,1), "tot_ms", round(float(r["TotalDurationNs"])/1e6,1))
PY
