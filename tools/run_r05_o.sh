#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
SWEEP_VARIANTS=0,11008,1011008,1011108 timeout 300 python tools/quick_sweep_dma.py 1 3 9 300 10000 10240 25003 > $OUT/sweep_dma_o.log 2>&1; echo "rc $?" >> $OUT/sweep_dma_o.log
grep -v '^\[mrslam\]\|^{' $OUT/sweep_dma_o.log | grep -v "N=     [139] \|N=   300" | tail -n 20
grep -c "mismatches=0" $OUT/sweep_dma_o.log; grep -c "variant" $OUT/sweep_dma_o.log
timeout 300 python tools/quick_sweep_mc.py 10000 > $OUT/sweep_mc_o.log 2>&1; grep -v mrslam $OUT/sweep_mc_o.log | tail -n 4
for v in 3 5; do
MRS_DEV=1 MRS_LIN_VARIANT=$v timeout 400 python bench.py --no-extra-legs --no-cpu-baseline --steps 2 --warmup 1 --chunks 4 --fuse 4 --verify 0 > $OUT/bench_gicp_o$v.json 2> $OUT/bench_gicp_o$v.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open("$OUT/bench_gicp_o$v.json").read().strip().splitlines()[-1])
g = d["gicp"]
print("variant=$v  cold20 it/s", round(g["iters_per_s"]), "warm", round(g["warm"]["iters_per_s"]), "natural", round(g["natural"]["pairs_per_s"]), "shared", round(g["shared_submaps"]["pairs_per_s_incl_covariances"]), "lin", round(g["kernel_ms"]["linearize"], 3), "err-only", round(g["kernel_ms"]["linearize_error_only"], 3), "lin frac", round(g["roofline"]["k_linearize"]["frac"], 3))
PY
done
