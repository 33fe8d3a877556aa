#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_fused_gpu.py -x -q -m gpu > $OUT/pytest_fused.log 2>&1; tail -n 3 $OUT/pytest_fused.log
timeout 600 python tools/ab_fused.py --chunks 16 > $OUT/ab_fused.log 2>&1; grep -E "identical|'stagger_us': 70|separate|best" $OUT/ab_fused.log; cp gpurun_out/ab_fused.json $OUT/ab_fused.json
(cd /tmp && $R/tools/ubench/valu > $OUT/ubench_valu2.json 2> $OUT/ubench_valu2.err); python - <<PY
import json
d=json.load(open('$OUT/ubench_valu2.json'))
for k,v in d['valu'].items():
    print(f"{k:50s}", "  ".join(f"w{w}: {v['w%d'%w]['cyc_at_2.4GHz_wall']:.2f}" for w in (1,2,4,8)))
PY
