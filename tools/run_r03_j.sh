#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_preprocess_gpu.py -x -q -m gpu 2>&1 | tail -n 12
timeout 300 python - <<PY 2>&1 | grep -v amdgpu.ids
import json, torch, bench
ch = bench.make_shard(1024, 1, 0, "cuda:0")
r = bench.build_legs("cuda:0", ch)
print(json.dumps(r["ingest"], indent=1)); print(json.dumps(r["ringpp_build"]["ms"])); print(r["ringpp_build"]["scans_per_s"])
PY
