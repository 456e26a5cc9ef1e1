#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-f64prof}
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/bench_f64.py > $O/log.txt 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
python3 tools/trace_summary.py $f ${STEPS:-1} 16 > $O/kernels.txt
tail -3 $O/log.txt; cat $O/kernels.txt
rm -rf $O/kt
