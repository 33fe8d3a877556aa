#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
echo skip-tests
SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_j.json 2> $OUT/bench_j.err; echo "bench rc $? wall ${SECONDS}s"; tail -n 5 $OUT/bench_j.err
python - <<PY
import json
d = json.loads(open("$OUT/bench_j.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "verify", d.get("verify", {}).get("ok"))
r = d["roofline"]
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if not isinstance(v, dict)})
g = d["gicp"]; print("shared", json.dumps(g["shared_submaps"]))
c = d["cpu_baseline"]; print("cpu", c["value"], c["cores"], c.get("value_at_nproc_threads"), {k: round(v["value"], 1) for k, v in c["at_threads"].items()})
print("cpu gicp", {k: round(v["iters_per_s"], 1) for k, v in c["gicp"]["by_threads"].items()}, c["gicp"].get("all_cores"))
for k, v in d["sweeps"].items(): print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("pairs_per_s", "ms", "hbm_frac", "queries_per_s")})
print("node", {k: (round(v["pairs_per_s"]) if isinstance(v, dict) and "pairs_per_s" in v else v) for k, v in d["node_shape"].items() if k != "append"})
print("builds", {k: v for k, v in d["builds"].items() if not isinstance(v, dict)})
PY
