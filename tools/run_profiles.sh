#!/bin/bash
# Collects the round's profiles on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the default bench run, then one --pmc pass per counter group over tools/pmc_targets.py.
# PMC passes carry no tracing flags (gpurun refuses --pmc together with trace domains).
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_valu $OUT/pmc_lds
# the plain bench run and the kernel trace come first, the counter passes last
(cd $R && python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $R/tools/pmc_targets.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $R/tools/pmc_targets.py > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT/pmc_valu -- python $R/tools/pmc_targets.py > $OUT/pmc_valu.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_lds -- python $R/tools/pmc_targets.py > $OUT/pmc_lds.log 2>&1
[ -n "$WITH_TESTS" ] && (cd $R && timeout 400 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; tail -2 $OUT/pytest_all.log)
ls $OUT; tail -2 $OUT/*.log
