#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
for v in 1 2 3 4; do
MRS_DEV=1 MRS_LIN_VARIANT=$v timeout 400 python bench.py --no-extra-legs --no-cpu-baseline --steps 2 --warmup 1 --chunks 4 --fuse 4 --verify 0 > $OUT/bench_gicp_n$v.json 2> $OUT/bench_gicp_n$v.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open("$OUT/bench_gicp_n$v.json").read().strip().splitlines()[-1])
g = d["gicp"]
print("variant=$v  cold20 it/s", round(g["iters_per_s"]), "warm", round(g["warm"]["iters_per_s"]), "natural", round(g["natural"]["pairs_per_s"]), "shared", round(g["shared_submaps"]["pairs_per_s_incl_covariances"]), "lin", round(g["kernel_ms"]["linearize"], 3), "err-only", round(g["kernel_ms"]["linearize_error_only"], 3))
PY
done
