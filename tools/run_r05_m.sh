#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_bench_contract_gpu.py tests/test_gicp_gpu.py tests/test_pybind_pygicp.py tests/test_cpp_adapter.py -m gpu -x -q > $OUT/pytest_m.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_m.log; tail -n 12 $OUT/pytest_m.log | cut -c1-300
timeout 400 python bench.py --no-extra-legs --no-cpu-baseline --steps 2 --warmup 1 --chunks 4 --fuse 4 --verify 0 > $OUT/bench_gicp_m.json 2> $OUT/bench_gicp_m.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open("$OUT/bench_gicp_m.json").read().strip().splitlines()[-1])
g = d["gicp"]
print("cold20 it/s", round(g["iters_per_s"]), "warm", round(g["warm"]["iters_per_s"]), "cold5", round(g["cold"]["iters_per_s"]), "natural", round(g["natural"]["pairs_per_s"]), "incl cov", round(g["pairs_per_s_incl_covariances"]), "shared", round(g["shared_submaps"]["pairs_per_s_incl_covariances"]))
print({k: round(v, 3) for k, v in g["kernel_ms"].items()}, "lin frac", round(g["roofline"]["k_linearize"]["frac"], 3), round(g["roofline"]["k_linearize_error_only"]["frac"], 3), "certify frac", round(g["roofline"]["k_nn_certify (unchanged pose)"]["frac"], 3))
PY
