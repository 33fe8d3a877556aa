#!/bin/bash
# Collects the round's profiles on the GPU box (run through gpurun from the repo root):
#   default bench, kernel-trace stats of a short bench, then one --pmc pass per counter group over tools/pmc_targets.py.
# Every step runs under its own `timeout`: one --pmc pass of round 3 hung for 37 minutes (profiler, not the kernels: the same pass took 8 s before).
# PMC passes carry no tracing flags (gpurun refuses --pmc together with trace domains).  The raw counter CSVs of a pass hold every torch
# kernel of the scan synthesis and exceed what gpurun copies back: only the rows of this library's kernels are kept (pmc_<group>.csv).
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -z "$PMC_ONLY" ]; then
(cd $R && timeout 400 python bench.py --detail-file $OUT/bench_detail.json > $OUT/bench_default.json 2> $OUT/bench_default.err)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --verify 0 > $OUT/stats.log 2>&1
for f in $(find $OUT/stats -name '*kernel_stats.csv'); do cp $f $OUT/kernel_stats.csv; done; rm -rf $OUT/stats
fi
slim() { d=$1; for f in $(find $d -name '*counter_collection.csv'); do (head -1 $f; grep -E 'k_[a-z_0-9]+[<(]' $f) > $d.csv; done; rm -rf $d; }
pass() { name=$1; shift
         for try in 1 2; do      # a pass that hangs in the profiler (seen in rounds 3 and 5) is cut at 150 s and tried once more
             timeout 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python $R/tools/pmc_targets.py > $OUT/pmc_$name.log 2>&1
             slim $OUT/pmc_$name
             [ -s $OUT/pmc_$name.csv ] && break
         done; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass valu SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass classes SQ_INSTS_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE
pass lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
ls -la $OUT; for f in $OUT/*.log; do tail -n 1 $f; done; head -c 400 $OUT/bench_default.json
