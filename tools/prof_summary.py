"""Condense rocprofv3 CSV output (gpurun_out/<tag>/{stats,pmc_fetch,pmc_write}) into
profiles/<tag>_*.{md,csv,json}: per-kernel time stats and per-launch HBM traffic from the PMC
counters, corrected as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes
(FETCH_SIZE counts 64 B per 128-B request on gfx950 for wide coalesced reads -> x2; unit KB)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def newest(paths):
    """gpurun merges every call's files into gpurun_out/: keep the most recent run of each directory"""
    by_dir = {}
    for f in paths:
        d = os.path.dirname(f)
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
            by_dir[d] = f
    return sorted(by_dir.values())

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0][-60:]


rows = []
for f in newest(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))) + glob.glob(os.path.join(src, "kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        rows.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]),
                     float(r["MinNs"]), float(r["MaxNs"]), float(r["Percentage"])))
rows.sort(key=lambda r: -r[2])
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as o:
    o.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
    for r in rows:
        o.write(f"{r[0]},{r[1]},{r[2]/1e3:.1f},{r[3]/1e3:.2f},{r[4]/1e3:.2f},{r[5]/1e3:.2f},{r[6]:.3f}\n")

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc_fetch", "pmc_write"):
    for f in newest(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))) + glob.glob(os.path.join(src, sub + ".csv")):
        for r in csv.DictReader(open(f)):
            if "k_" in r["Kernel_Name"]:
                pmc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
traffic = {}
for (k, grid), c in sorted(pmc.items()):
    f = sum(c.get("FETCH_SIZE", [0])) / max(1, len(c.get("FETCH_SIZE", [0])))
    w = sum(c.get("WRITE_SIZE", [0])) / max(1, len(c.get("WRITE_SIZE", [0])))
    traffic[f"{k}@grid{grid}"] = {"FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w,
                                  "hbm_read_bytes": 2 * f * 1024, "hbm_write_bytes": w * 1024,
                                  "hbm_bytes": (2 * f + w) * 1024, "launches": len(c.get("FETCH_SIZE", []))}
json.dump(traffic, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)

with open(os.path.join(dst, f"{tag}_summary.md"), "w") as o:
    o.write(f"# rocprofv3 summary, round tag `{tag}`\n\n")
    o.write("Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --verify 0` (tools/run_profiles.sh; "
            "1 x MI355X, 48 launches x 1024 scan pairs per step, BEV + Radon + normalisation of 16 launches per fused kernel call, GICP leg 256 pairs x 120k: 5 cold + 20 forced iterations + a run to convergence).\n"
            "PMC passes (separate runs over tools/pmc_targets.py, `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, no tracing; more counters in " + tag + "_pmc.md).\n\n")
    o.write("| kernel | calls | avg us | min us | max us | % of GPU time |\n|---|---|---|---|---|---|\n")
    for r in rows[:14]:
        o.write(f"| `{r[0]}` | {r[1]} | {r[3]/1e3:.1f} | {r[4]/1e3:.1f} | {r[5]/1e3:.1f} | {r[6]:.2f} |\n")
    o.write("\n## HBM traffic per launch (PMC)\n\nFETCH_SIZE is doubled (gfx950 counts 64 B per 128-B request), units KB -> bytes.\n\n")
    o.write("| kernel @ grid | launches | read MB | write MB | total MB |\n|---|---|---|---|---|\n")
    for k, v in traffic.items():
        o.write(f"| `{k}` | {v['launches']} | {v['hbm_read_bytes']/1e6:.2f} | {v['hbm_write_bytes']/1e6:.2f} | {v['hbm_bytes']/1e6:.2f} |\n")
print(open(os.path.join(dst, f"{tag}_summary.md")).read())
