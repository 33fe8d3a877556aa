#!/bin/bash
# round 3: persistent workgroups of the descriptor kernel vs compute units left to the side-stream sweeps
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for w in 0 252 248 240; do
  timeout 60 python bench.py --no-cpu-baseline --no-extra-legs --gicp-pairs 0 --verify 2 --fused-wgs $w > $OUT/wgs_$w.json 2> $OUT/wgs_$w.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/wgs_$w.json").read().strip().splitlines()[-1])
    print("fused-wgs $w:", round(d["value"]), "pairs/s", round(d["ms_per_step"], 3), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k in ("bev_radon", "corr", "sweep")}, d["verify"]["ok"])
except Exception as e:
    print("fused-wgs $w FAILED", e)
PY
done
