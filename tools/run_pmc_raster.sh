#!/bin/bash
# PMC pass over the fused kernel's rasteriser alone (MRS_DEV=1 MRS_FUSED_SKIP=2) at 4 x 1024 scans: where do a wave's cycles go?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/raster_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MRS_DEV=1 MRS_FUSED_SKIP=2 PMC_FUSED_GROUP=4 PMC_FUSED_GRID=${PMC_FUSED_GRID:-256}
pass() { name=$1; shift; timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python $R/tools/pmc_fused.py > $OUT/$name.log 2>&1
         for f in $(find $OUT/$name -name '*counter_collection.csv'); do (head -1 $f; grep -E 'k_bev_radon' $f) > $OUT/$name.csv; done; rm -rf $OUT/$name; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
pass b SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES
pass c SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/*.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[-1], {k: sum(v) / len(v) for k, v in acc.items()})
PY
