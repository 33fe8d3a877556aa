#!/bin/bash
# third GPU call of round 3: march ablations; calibration of the LDS / VALU-cycle counters on the microbenchmark; instruction classes of the fused kernel
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
$R/tools/ubench/radon_march 2048 > $OUT/radon_march2.json 2> $OUT/radon_march2.err
cat $OUT/radon_march2.json; tail -n 3 $OUT/radon_march2.err
slim() { d=$1; pat=$2; for f in $(find $d -name '*counter_collection.csv'); do (head -1 $f; grep -E "$pat" $f) > $d.csv; done; rm -rf $d; }
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_ubench2 -- $R/tools/ubench/valu > $OUT/pmc_ubench2.log 2>&1
slim $OUT/pmc_ubench2 'k_fma|k_add|k_pk_fma|k_cvt_i32|k_lds_b64|k_rcp|k_cndmask'
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_fused_$name -- python $R/tools/pmc_fused.py > $OUT/pmc_fused_$name.log 2>&1; slim $OUT/pmc_fused_$name k_bev_radon2; }
pass classes SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE
ls -la $OUT | tail -n 8; for f in $OUT/pmc_ubench2.log $OUT/pmc_fused_classes.log; do tail -n 2 $f; done
