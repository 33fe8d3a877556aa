#!/bin/bash
# One GPU call: A/B of the fused descriptor kernel (bit identity on the whole shard + timings), its tests and the BEV tests, the default
# bench, then the whole GPU suite.  Every part has its own timeout and log under gpurun_out/r02c.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02c
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
S=$OUT/status.txt
: > $S
run() {   # name, timeout, command...
    local name=$1 t=$2; shift 2
    local t0=$SECONDS
    timeout $t "$@" > $OUT/$name.log 2> $OUT/$name.err
    echo "$name rc=$? ${t0}s->${SECONDS}s" >> $S
}
run ab 60 python tools/ab_fused.py
run pytest_fused 60 python -m pytest tests/test_fused_gpu.py tests/test_bev_gpu.py -x -q -m gpu
run bench_default 90 python bench.py
[ $SECONDS -lt 90 ] && run pytest_all 130 python -m pytest tests -x -q -m gpu
cat $S
tail -3 $OUT/ab.log
tail -2 $OUT/pytest_fused.log
