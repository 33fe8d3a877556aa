#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 python $R/tools/quick_knn_scale.py 2>&1 | tail -n 2
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/knn_stats2 -- python $R/tools/quick_knn_scale.py > $OUT/knn_stats2.log 2>&1
for f in $(find $OUT/knn_stats2 -name '*kernel_trace.csv'); do grep -E "k_knn_cov|k_feat_from" $f | awk -F, '{print $0}' | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin):
    nums=[x for x in r if x.isdigit()]
    name=[x for x in r if 'k_' in x][0][:40]
    # start/end timestamps are the two largest numbers
    big=sorted(int(x) for x in nums)[-2:]
    print(name, (big[1]-big[0])/1e3, [x for x in nums if int(x)<100000][-6:])
" | head -40; done
rm -rf $OUT/knn_stats2
