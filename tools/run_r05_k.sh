#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
for ch in 1 4 8 16; do
MRS_DEV=1 MRS_LIN_CHUNKS=$ch timeout 400 python bench.py --no-extra-legs --no-cpu-baseline --steps 2 --warmup 1 --chunks 4 --fuse 4 --verify 0 > $OUT/bench_gicp_k$ch.json 2> $OUT/bench_gicp_k$ch.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open("$OUT/bench_gicp_k$ch.json").read().strip().splitlines()[-1])
g = d["gicp"]
print("chunks=$ch  cold20 it/s", round(g["iters_per_s"]), "warm", round(g["warm"]["iters_per_s"]), "natural", round(g["natural"]["pairs_per_s"]), "incl cov", round(g["pairs_per_s_incl_covariances"]), "shared", round(g["shared_submaps"]["pairs_per_s_incl_covariances"]))
print({k: round(v, 3) for k, v in g["kernel_ms"].items()}, "lin frac", round(g["roofline"]["k_linearize"]["frac"], 3), round(g["roofline"]["k_linearize_error_only"]["frac"], 3), "certify frac", round(g["roofline"]["k_nn_certify (unchanged pose)"]["frac"], 3))
PY
done
timeout 600 python -m pytest tests/test_gicp_gpu.py tests/test_pybind_pygicp.py -m gpu -x -q > $OUT/pytest_k.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_k.log; tail -n 4 $OUT/pytest_k.log | cut -c1-200
