#!/bin/bash
# development aid: does the GPU test suite leave the box in a state that slows the short kernels of the bench step?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/probe; rm -rf $OUT; mkdir -p $OUT; cd $R
q() { python bench.py --steps 3 --warmup 1 --gicp-pairs 0 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})"; }
snap() { echo "--- $1"; ps -eo pid,ppid,stat,etime,cmd | grep -E "python|adapter|elev_main" | grep -v grep | head -20; rocm-smi --showpids 2>/dev/null | tail -8; rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -4; }
{
q fresh
snap fresh
for t in tests/test_bench_contract_gpu.py tests/test_cpp_adapter.py tests/test_gicp_gpu.py "tests/test_bev_gpu.py tests/test_ring_gpu.py tests/test_elevation_gpu.py" ; do
  python -m pytest $t -m gpu -q -x 2>&1 | tail -1
  q "after:$t"
done
snap end
python -m pytest tests -m gpu -q -x 2>&1 | tail -1
q after:all
snap all
} > $OUT/probe.txt 2>&1
cat $OUT/probe.txt
