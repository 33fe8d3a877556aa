#!/bin/bash
# last GPU call of round 3: the whole GPU suite and the default bench line at HEAD (every step under its own timeout)
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 420 python -m pytest tests -m gpu -x -q > $OUT/pytest_final.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_final.log
tail -n 4 $OUT/pytest_final.log
timeout 240 python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err; echo "bench rc $?"
cat $OUT/bench_final.json | cut -c1-1500
