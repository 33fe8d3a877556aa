#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_node_gpu.py -m gpu -q > $OUT/pytest_g.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_g.log; tail -n 40 $OUT/pytest_g.log | cut -c1-220
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/bench_g.json 2> $OUT/bench_g.err; echo "bench rc $?"; tail -n 5 $OUT/bench_g.err
python - <<PY
import json
d = json.loads(open("$OUT/bench_g.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"])
print(json.dumps(d.get("node_shape"), indent=1))
for k, v in d["sweeps"].items(): print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("pairs_per_s", "ms", "hbm_frac", "queries_per_s")})
PY
