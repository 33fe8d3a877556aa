#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd $R
run() { MRS_FUSED_PROF=1 MRS_FUSED_SKIP=$1 MRS_PF=$2 MRS_STAG=$3 python - <<PY 2>&1 | grep -v amdgpu.ids | tail -n 2
import os, sys, torch
sys.path.insert(0, '.')
import bench
from mr_slam_amd import ring
dev='cuda:0'; B=1024; G=16
bench.make_shard(B, G, 0, dev); whole=bench.make_shard.whole
offs=torch.arange(G*B+1, dtype=torch.int64, device=dev)*bench.N_POINTS
out=torch.empty((G*B,120,120), dtype=torch.float32, device=dev)
plan=ring.ring_plan(0); plan.set_option(plan.OPT_FUSED_PREFETCH, int(os.environ['MRS_PF'])); plan.set_option(plan.OPT_FUSED_STAGGER_US, int(os.environ['MRS_STAG']))
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
for _ in range(2): ring.ring_descriptors_fused(whole.view(-1), offs, raw=False, normalized=True, out_norm=out)
a.record(); ring.ring_descriptors_fused(whole.view(-1), offs, raw=False, normalized=True, out_norm=out); b.record(); torch.cuda.synchronize()
print("skip", os.environ['MRS_FUSED_SKIP'], "pf", os.environ['MRS_PF'], "stagger", os.environ['MRS_STAG'], "ms per 1024 scans (incl. prof sync): %.4f" % (a.elapsed_time(b)/G), file=sys.stderr)
PY
}
(run 0 2 70; run 2 2 0; run 2 4 0; run 2 6 0; run 1 2 0) > $OUT/fused_prof2.log 2>&1
cat $OUT/fused_prof2.log
