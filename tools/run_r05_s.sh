#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/knn_stats -- python $R/tools/pmc_knn_targets.py > $OUT/knn_stats.log 2>&1
for f in $(find $OUT/knn_stats -name '*kernel_stats.csv'); do cp $f $OUT/knn_kernel_stats.csv; done; rm -rf $OUT/knn_stats
head -n 14 $OUT/knn_kernel_stats.csv | cut -c1-200
slim() { d=$1; for f in $(find $d -name '*counter_collection.csv'); do (head -1 $f; grep -E 'k_knn_cov|k_feat_from|k_cov_from' $f) > $d.csv; done; rm -rf $d; }
pass() { name=$1; shift
         timeout 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/knnpmc_$name -- python $R/tools/pmc_knn_targets.py > $OUT/knnpmc_$name.log 2>&1
         slim $OUT/knnpmc_$name; }
pass valu SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
python - <<'PY'
import csv, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05"
for name in ("valu", "wait", "fetch"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    try:
        rows = list(csv.DictReader(open(f"{out}/knnpmc_{name}.csv")))
    except Exception as e:
        print(name, "missing", e); continue
    for r in rows:
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        print(name, k, len(n[k]), {c: round(x / len(n[k])) for c, x in v.items()})
PY
