#!/bin/bash
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gicp_gpu.py tests/test_pybind_pygicp.py -m gpu -x -q > $OUT/pytest_g.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_g.log; tail -n 4 $OUT/pytest_g.log
timeout 300 python bench.py --no-extra-legs --no-cpu-baseline --steps 2 --warmup 1 --chunks 4 --fuse 4 --verify 0 > $OUT/bench_gicp.json 2> $OUT/bench_gicp.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open("$OUT/bench_gicp.json").read().strip().splitlines()[-1])
g = d["gicp"]
print("it/s", round(g["iters_per_s"]), "cold", round(g["cold"]["iters_per_s"]), "natural", round(g["natural"]["pairs_per_s"]), "incl cov", round(g["pairs_per_s_incl_covariances"]), "searched", g["natural"]["searched_fraction"])
print({k: round(v, 3) for k, v in g["kernel_ms"].items()}, "lin frac", round(g["roofline"]["k_linearize"]["frac"], 3), round(g["roofline"]["k_linearize_error_only"]["frac"], 3))
PY
