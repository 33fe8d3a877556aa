#!/bin/bash
# kernel-trace stats of the default bench step at HEAD (same command as tools/run_profiles.sh's stats pass), under its own timeout
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --gicp-pairs 0 --verify 0 > $OUT/stats2.log 2>&1
for f in $(find $OUT/stats2 -name '*kernel_stats.csv'); do cp $f $OUT/kernel_stats2.csv; done; rm -rf $OUT/stats2
head -n 12 $OUT/kernel_stats2.csv; tail -n 2 $OUT/stats2.log
