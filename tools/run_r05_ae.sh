#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $OUT/gpu_tests.txt; tail -n 4 $OUT/gpu_tests.txt | cut -c1-200
timeout 500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
g = d["gicp"]; print({k: g[k] for k in ("iters_per_s", "pairs_per_s_incl_covariances", "covariance_s")}, g["natural"]["pairs_per_s"], g["cold"]["iters_per_s"], g["shared_submaps"]["pairs_per_s_incl_covariances"])
print(g["kernel_ms"]); print(d["builds"]["ringpp_build"]["scans_per_s"], d["builds"]["ringpp_build"]["ms"]); print(d["dropin_latency"])
PY
