"""Condenses the rocprofv3 --pmc passes over tools/pmc_targets.py (gpurun_out/<tag>/pmc_<group>/...) into
profiles/<tag>_pmc.json + .md.  HBM bytes: FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE doubled on gfx950 (64 B counted per
128-B request) as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes.  VALU-pipe busy fraction: no SQ counter measures it on
gfx950 (profiles/r03_ubench.md), so it is built from the instruction mix: SQ_INSTS_VALU_{ADD,MUL}_F32 x 2 cycles, _FMA_F32 x 4 (the kernels here issue
their FMAs packed), _CVT x 4, _INT32 x 3, _TRANS_F32 x 8, the unclassified rest (fract / floor / max / compare / select / move / logic) x 3 -- the measured
cost of each class (tools/ubench/valu.hip) -- divided by (256 CUs x 4 SIMDs x active cycles per XCD); without the class counters the bounds
SQ_INSTS_VALU x 2 and x 4 are given instead.  LDS busy = SQ_LDS_IDX_ACTIVE / (256 CUs x active cycles per XCD); active cycles per XCD = GRBM_GUI_ACTIVE / 8."""
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

CLASS_CYCLES = {"SQ_INSTS_VALU_ADD_F32": 2.0, "SQ_INSTS_VALU_MUL_F32": 2.0, "SQ_INSTS_VALU_FMA_F32": 4.0, "SQ_INSTS_VALU_CVT": 4.0,
                "SQ_INSTS_VALU_INT32": 3.0, "SQ_INSTS_VALU_TRANS_F32": 8.0}
REST_CYCLES = 3.0


def valu_cycles(mean):
    """VALU-pipe cycles of a launch from its instruction classes (None without the class counters)"""
    if "SQ_INSTS_VALU" not in mean or not all(k in mean for k in CLASS_CYCLES):
        return None
    known = sum(mean[k] for k in CLASS_CYCLES)
    return sum(mean[k] * c for k, c in CLASS_CYCLES.items()) + max(mean["SQ_INSTS_VALU"] - known, 0.0) * REST_CYCLES


tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
KEYS = {"k_cart_lds": "k_cart_lds", "k_polar_lds": "k_polar_lds", "k_radon2": "k_radon2", "k_bev_radon2": "k_bev_radon2", "k_bev_radon3": "k_bev_radon3",
        "k_ring_spec_corr_pairs": "k_ring_spec_corr_pairs", "k_ring_corr_fft": "k_ring_corr_fft", "k_linearize": "k_linearize", "k_nn_scan": "k_nn_scan",
        "k_knn_cov": "k_knn_cov", "k_knn_features": "k_knn_features", "k_nn_scan_g": "k_nn_scan_g", "k_nn_certify": "k_nn_certify",
        "k_cov_from_knn": "k_cov_from_knn", "k_knn_select": "k_knn_select", "k_feat_from_knn": "k_feat_from_knn", "k_ring_sweep_dma": "k_ring_sweep_dma"}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
files = newest(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))) + sorted(glob.glob(os.path.join(src, "pmc_*.csv")))   # raw passes or slimmed ones
for f in files:
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        if not m or m.group(1) not in KEYS:
            continue
        k = m.group(1)
        if k == "k_ring_corr_fft":
            k += "@grid%s" % r["Grid_Size"]
        if k == "k_ring_sweep_dma":                # row layout / DMA-tiled / multi-channel are different instantiations: <WAVES, SPLIT, NT, RING, QDMA, TILED, PRIO, MC, PF>
            t = re.search(r"k_ring_sweep_dma<([^>]*)>", r["Kernel_Name"])
            a = [x.strip() for x in t.group(1).split(",")] if t else []
            if len(a) >= 8:
                k += "<%s%s, %s waves%s>" % ("tiled" if a[5] in ("true", "1") else "rows", ", 6 channels" if a[7] in ("true", "1") else "", a[0],
                                             "" if a[2] != "0" else ", several queries")      # NT = 0 (default cache policy) is the several-queries form
        if k == "k_knn_cov":                       # k = 15 covariances (16 slots) and the k = 30 point-feature selection (32 slots) are different kernels
            t = re.search(r"k_knn_cov<\s*(\d+)", r["Kernel_Name"])
            if t:
                k += "<%s>" % t.group(1)
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in sorted(acc.items()):
    mean = {n: sum(v) / len(v) for n, v in c.items()}
    e = {"launches": max(len(v) for v in c.values()), "counters": mean}
    if k.startswith("k_bev_radon"):
        e["launch_scans"] = 16384          # tools/pmc_targets.py runs the fused kernel at the bench's launch size (16 x 1024 scans)
    if "FETCH_SIZE" in mean or "WRITE_SIZE" in mean:
        e["hbm_read_bytes"] = 2 * mean.get("FETCH_SIZE", 0.0) * 1024
        e["hbm_write_bytes"] = mean.get("WRITE_SIZE", 0.0) * 1024
        e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
    if "GRBM_GUI_ACTIVE" in mean:
        cyc = mean["GRBM_GUI_ACTIVE"] / 8.0
        e["active_cycles_per_xcd"] = cyc
        if "SQ_INSTS_VALU" in mean:
            e["valu_busy_frac_bounds"] = [mean["SQ_INSTS_VALU"] * 2.0 / (256 * 4 * cyc), mean["SQ_INSTS_VALU"] * 4.0 / (256 * 4 * cyc)]
            e["valu_pipe_cycles_est"] = valu_cycles(mean)
            if e["valu_pipe_cycles_est"]:
                e["valu_issue_frac"] = e["valu_pipe_cycles_est"] / (256 * 4 * cyc)
        if "SQ_LDS_IDX_ACTIVE" in mean:
            e["lds_busy_frac"] = mean["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)
        if "SQ_LDS_BANK_CONFLICT" in mean and mean.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = mean["SQ_LDS_BANK_CONFLICT"] / mean["SQ_LDS_IDX_ACTIVE"]
    out[k] = e
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
with open(os.path.join(root, "profiles", f"{tag}_pmc.md"), "w") as o:
    o.write(f"# rocprofv3 --pmc summary `{tag}` (tools/pmc_targets.py, 1 x MI355X, separate passes per counter group, no tracing)\n\n")
    o.write("| kernel | launches | HBM read MB | HBM write MB | VALU issue | LDS busy | LDS conflict share |\n|---|---|---|---|---|---|---|\n")
    for k, e in out.items():
        f = lambda v, s=1.0, p="{:.2f}": p.format(v * s) if v is not None else "-"
        o.write(f"| `{k}` | {e['launches']} | {f(e.get('hbm_read_bytes'), 1e-6)} | {f(e.get('hbm_write_bytes'), 1e-6)} | "
                f"{f(e.get('valu_issue_frac'))} | {f(e.get('lds_busy_frac'))} | {f(e.get('lds_bank_conflict_frac'))} |\n")
print(open(os.path.join(root, "profiles", f"{tag}_pmc.md")).read())
