#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout 200 python $R/tools/quick_single_pair.py 2>&1 | tail -n 2
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sp_stats -- python $R/tools/quick_single_pair.py > $OUT/sp_stats.log 2>&1
for f in $(find $OUT/sp_stats -name '*kernel_stats.csv'); do cp $f $OUT/sp_kernel_stats.csv; done; rm -rf $OUT/sp_stats
python3 - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05/sp_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per registration (22 reps + downsample): %.3f ms" % (tot / 22 / 1e6))
for r in rows[:22]:
    print(r["Name"][:60].replace("\n", " "), r["Calls"], round(float(r["TotalDurationNs"]) / 22 / 1e3, 1), "us/reg", round(float(r["AverageNs"]) / 1e3, 1))
PY
