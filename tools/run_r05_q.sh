#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
cat /sys/fs/cgroup/cpu.max; nproc
S=$SECONDS
timeout 600 python -m pytest tests/test_bench_contract_gpu.py -m gpu -q -x 2>&1 | tail -n 4 | cut -c1-300; echo "contract: $((SECONDS-S)) s"
S=$SECONDS
timeout 500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $? $((SECONDS-S)) s"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/bench_default.json"))
print(d["value"], d["ms_per_step"]); print(json.dumps(d["cpu_baseline"])[:1500])
PY
