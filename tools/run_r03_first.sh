#!/bin/bash
# First GPU call of round 3: issue-rate microbenchmark (tools/ubench/valu), its SQ counters (calibrates what the counters count),
# SQ / LDS / HBM counters of the fused descriptor kernel at 16 x 1024 scans per launch, default bench at HEAD.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
$R/tools/ubench/valu > $OUT/ubench_valu.json 2> $OUT/ubench_valu.err
rocprofv3 -L > $OUT/counters_list.txt 2>&1
# keep only the rows of the kernels of interest: the raw counter CSVs of a pass (every torch kernel of the scan synthesis included)
# exceed what gpurun copies back
slim() { d=$1; pat=$2; for f in $(find $d -name '*counter_collection.csv'); do (head -1 $f; grep -E "$pat" $f) > $d.csv; done; rm -rf $d; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_ubench -- $R/tools/ubench/valu > $OUT/pmc_ubench.log 2>&1
slim $OUT/pmc_ubench 'k_fma|k_pk_fma|k_cvt_i32|k_mul_lo|k_lds_b64|k_fma_f64'
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_fused_$name -- python $R/tools/pmc_fused.py > $OUT/pmc_fused_$name.log 2>&1; slim $OUT/pmc_fused_$name k_bev_radon2; }
pass valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE
pass wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
(cd $R && python bench.py > $OUT/bench_head.json 2> $OUT/bench_head.err)
ls -la $OUT; du -sh $OUT; for f in $OUT/*.log; do tail -n 2 $f; done; head -c 600 $OUT/bench_head.json
(cd $R && timeout 300 python -m pytest tests/test_ref_pins_gpu.py tests/test_ring_gpu.py tests/test_pybind_pygicp.py -x -q -m gpu > $OUT/pytest_c3.log 2>&1; tail -n 3 $OUT/pytest_c3.log)
