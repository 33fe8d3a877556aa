#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
SWEEP_VARIANTS=0,1008,1108,1112,101108,101112,11108,11112,108,112 timeout 300 python tools/quick_sweep_dma.py 10000 25003 50000 > $OUT/sweep_dma_b.log 2>&1; echo "rc $?" >> $OUT/sweep_dma_b.log
grep -v '^\[mrslam\]\|^{' $OUT/sweep_dma_b.log | tail -n 40
cd /tmp && export TMPDIR=/tmp
export SWEEP_VARIANTS=0,1108,1112
slim() { d=$1; for f in $(find $d -name '*counter_collection.csv'); do (head -1 $f; grep -E 'k_ring_[a-z_0-9]+[<(]' $f) > $d.csv; done; rm -rf $d; }
pass() { name=$1; shift; timeout 180 rocprofv3 --pmc "$@" --output-format csv -d $OUT/sweepdma_$name -- python $R/tools/pmc_sweep_dma_targets.py > $OUT/sweepdma_$name.log 2>&1; slim $OUT/sweepdma_$name; }
pass wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE
ls -la $OUT/ | tail -n 12; tail -n 2 $OUT/sweepdma_*.log
