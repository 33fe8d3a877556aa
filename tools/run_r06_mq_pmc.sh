#!/bin/bash
# round 6: counters of the several-queries-per-sweep DMA pipeline (HBM-side requests, L2 hits, wave wait time) for 1 / 2 / 4 / 8 queries
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_mq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
slim() { d=$1; for f in $(find $d -name '*counter_collection.csv'); do (head -1 $f; grep -E 'k_ring_[a-z_0-9]+[<(]' $f) > $d.csv; done; rm -rf $d; }
pass() { name=$1; shift; timeout 180 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python $R/tools/pmc_sweep_mq_targets.py > $OUT/$name.log 2>&1; slim $OUT/$name; }
for nq in ${NQS:-1 2 4 8}; do
  export NQ=$nq
  pass tcc_q$nq TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
  pass wait_q$nq SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
done
python3 - <<'P'
import csv, glob, os, collections
out = os.environ.get("OUT_DIR") or os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r06_mq")
for f in sorted(glob.glob(out + "/*.csv")):
    rows = list(csv.DictReader(open(f)))
    agg = collections.OrderedDict()
    for r in rows:
        key = (r.get("Dispatch_Id"), r.get("Counter_Name"))
        agg[key] = agg.get(key, 0.0) + float(r.get("Counter_Value", 0))
    disp = sorted({k[0] for k in agg}, key=lambda x: int(x))
    last = disp[-1] if disp else None
    print(os.path.basename(f), {k[1]: v for k, v in agg.items() if k[0] == last})
P
