#!/bin/bash
# round 3: ring database slots (no per-step copy) on top of the grouped correlation launch + side-stream sweeps; contract tests; default line
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 200 python bench.py > $OUT/bench_sched2.json 2> $OUT/bench_sched2.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open("$OUT/bench_sched2.json").read().strip().splitlines()[-1])
print(round(d["value"]), "pairs/s", round(d["ms_per_step"], 3), "ms/step", {k: round(v, 4) for k, v in d["kernel_ms"].items()}, d["config"].get("database_slots"), "verify", d["verify"]["ok"], d["verify"].get("sweep_mismatches"), "frac", round(d["roofline"]["frac"], 4))
PY
timeout 420 python -m pytest tests/test_bench_contract_gpu.py -x -q > $OUT/pytest_sched2.log 2>&1; tail -n 5 $OUT/pytest_sched2.log
