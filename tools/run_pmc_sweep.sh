#!/bin/bash
# PMC passes over the database sweeps alone (1 and 4 queries as separate rows): instruction classes and wave-state split.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
slim() { d=$1; for f in $(find $d -name '*counter_collection.csv'); do (head -1 $f; grep -E 'k_ring_[a-z_0-9]+[<(]' $f) > $d.csv; done; rm -rf $d; }
pass() { name=$1; shift; timeout 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/sweep_$name -- python $R/tools/pmc_sweep_targets.py > $OUT/sweep_$name.log 2>&1; slim $OUT/sweep_$name; }
pass classes SQ_INSTS_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE
pass wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
ls -la $OUT/sweep_*; tail -n 1 $OUT/sweep_*.log
