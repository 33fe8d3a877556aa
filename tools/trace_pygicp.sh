#!/bin/bash
# one pygicp registration on a timeline: rocprofv3 --kernel-trace of tools/quick_pygicp_latency.py, kernel start / end per dispatch (development)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06t
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $R/tools/quick_pygicp_latency.py 6 > $OUT/run.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last registration: walk back from the end to the last k_cloud_bbox pair boundary -- take the last 120 dispatches
names = [r["Kernel_Name"] for r in rows]
# find starts of registrations: first k_cloud_bbox after a k_fitness
idx = [i for i, n in enumerate(names) if "k_fitness" in n]
if len(idx) >= 2:
    a, b = idx[-2] + 1, idx[-1] + 1
    seg = rows[a:b]
    t0 = int(seg[0]["Start_Timestamp"])
    busy = 0
    prev_end = t0
    gaps = []
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += e - s
        if s > prev_end:
            gaps.append((s - prev_end, r["Kernel_Name"][:50]))
        prev_end = max(prev_end, e)
    span = prev_end - t0
    print(f"dispatches {len(seg)} span {span/1e3:.1f} us busy(sum) {busy/1e3:.1f} us idle {sum(g for g,_ in gaps)/1e3:.1f} us")
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f}  {r['Kernel_Name'][:70]}")
P
