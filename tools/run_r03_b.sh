#!/bin/bash
# second GPU call of round 3: Radon march variants, then the whole GPU suite at HEAD
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
$R/tools/ubench/radon_march 2048 > $OUT/radon_march.json 2> $OUT/radon_march.err
cat $OUT/radon_march.json; tail -n 3 $OUT/radon_march.err
(cd $R && timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; tail -n 5 $OUT/pytest_all.log)
