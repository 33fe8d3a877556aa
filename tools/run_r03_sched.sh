#!/bin/bash
# round 3: schedule of the small kernels of the bench step -- grouped correlation launch and side-stream sweeps, A/B on one box
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for cg in 0 1; do for ss in main side; do
  timeout 120 python bench.py --no-cpu-baseline --no-extra-legs --gicp-pairs 0 --corr-group $cg --sweep-stream $ss > $OUT/sched_${cg}_${ss}.json 2> $OUT/sched_${cg}_${ss}.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/sched_${cg}_${ss}.json").read().strip().splitlines()[-1])
    print("corr_group $cg sweep $ss:", round(d["value"]), "pairs/s", round(d["ms_per_step"], 3), "ms/step", {k: round(v, 4) for k, v in d["kernel_ms"].items()}, "verify", d["verify"]["ok"], d["verify"].get("sweep_mismatches"))
except Exception as e:
    print("corr_group $cg sweep $ss: FAILED", e); print(open("$OUT/sched_${cg}_${ss}.err").read()[-1500:])
PY
done; done
timeout 400 python -m pytest tests/test_bench_contract_gpu.py -x -q > $OUT/pytest_sched.log 2>&1; tail -n 5 $OUT/pytest_sched.log
