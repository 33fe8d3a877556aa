#!/bin/bash
# One GPU call: A/B of the fused descriptor kernel, its tests, the bench step with and without it, then (time permitting)
# the default bench, the tests of the most recent commits and the whole GPU suite.  Every part has its own timeout and log.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02b
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
run ab 170 python tools/ab_fused.py
run pytest_fused 150 python -m pytest tests/test_fused_gpu.py -x -q
run bench_fuse8 120 python bench.py --steps 5 --warmup 2 --chunks 24 --fuse 8 --no-extra-legs --gicp-pairs 0 --no-cpu-baseline
run bench_fuse24 120 python bench.py --steps 5 --warmup 2 --chunks 24 --fuse 24 --no-extra-legs --gicp-pairs 0 --no-cpu-baseline
[ $SECONDS -lt 300 ] && run pytest_all 300 python -m pytest tests -x -q -m gpu
cat $S
tail -3 $OUT/ab.log; tail -2 $OUT/pytest_fused.log
