#!/bin/bash
# round 4: whole GPU suite + default bench line
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print(round(d["value"]), "pairs/s", round(d["ms_per_step"], 3), "ms/step", {k: round(v, 4) for k, v in d["kernel_ms"].items()}, "verify", d["verify"]["ok"])
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["roofline"].items() if not isinstance(v, (dict, str))})
print("gicp", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d["gicp"].items() if not isinstance(v, (dict, str))}, d["gicp"]["natural"], d["gicp"]["kernel_ms"], d["gicp"]["kernel_counts"])
print("cpu", {k: v for k, v in d["cpu_baseline"].items() if k in ("value", "cores", "value_at_16_threads")}, d["cpu_baseline"]["gicp"]["by_threads"])
print("builds", d["builds"]["ringpp_build"]["scans_per_s"], d["builds"]["ingest"]["scans_per_s"], "sweeps", {k: round(v["pairs_per_s"]/1e6, 1) for k, v in d["sweeps"].items()})
PY
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_all.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_all.log; tail -n 5 $OUT/pytest_all.log
