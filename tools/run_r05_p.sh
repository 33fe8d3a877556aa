#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
rm -f $OUT/pmc_*.csv
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $OUT/gpu_tests.txt; tail -n 6 $OUT/gpu_tests.txt | cut -c1-200
bash tools/run_profiles.sh r05 > $OUT/run_profiles.log 2>&1; ls -la $OUT | grep -E "pmc_|kernel_stats|bench_default"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
