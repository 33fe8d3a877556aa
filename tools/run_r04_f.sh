#!/bin/bash
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_ring_gpu.py tests/test_pointfeat_gpu.py tests/test_bench_contract_gpu.py -m gpu -x -q > $OUT/pytest_f.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_f.log; tail -n 5 $OUT/pytest_f.log
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"
timeout 100 python bench.py --no-cpu-baseline --no-extra-legs --gicp-pairs 0 --sweep-batch 0 > $OUT/bench_nobatch.json 2> $OUT/bench_nobatch.err; echo "bench(nobatch) rc $?"
python - <<PY
import json
for f in ("bench_default", "bench_nobatch"):
    d = json.loads(open("$OUT/" + f + ".json").read().strip().splitlines()[-1])
    print(f, round(d["value"]), "pairs/s", round(d["ms_per_step"], 3), "ms/step", {k: round(v, 4) for k, v in d["kernel_ms"].items()}, "verify", d["verify"]["ok"], d["verify"].get("sweep_mismatches"), d["config"].get("sweeps_per_launch"))
    if "builds" in d:
        print("  builds", d["builds"]["ringpp_build"], "\n  cpu", {k: v for k, v in d["cpu_baseline"].items() if k in ("value", "cores", "value_at_nproc_threads")})
PY
