#!/bin/bash
# last GPU call of round 3: default bench line (lagged join of the side-stream sweeps), the same with a join per step, the whole GPU suite
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 200 python bench.py > $OUT/bench_final2.json 2> $OUT/bench_final2.err; echo "bench rc $?"
timeout 100 python bench.py --no-cpu-baseline --no-extra-legs --gicp-pairs 0 --sweep-join step > $OUT/bench_joinstep.json 2> $OUT/bench_joinstep.err; echo "bench(step) rc $?"
python - <<PY
import json
for f in ("bench_final2", "bench_joinstep"):
    d = json.loads(open("$OUT/" + f + ".json").read().strip().splitlines()[-1])
    print(f, round(d["value"]), "pairs/s", round(d["ms_per_step"], 3), "ms/step", {k: round(v, 4) for k, v in d["kernel_ms"].items()}, d["config"].get("sweep_join"), "verify", d["verify"]["ok"], d["verify"].get("sweep_mismatches"), "frac", round(d["roofline"]["frac"], 4))
PY
timeout 460 python -m pytest tests -m gpu -x -q > $OUT/pytest_final2.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_final2.log; tail -n 4 $OUT/pytest_final2.log
