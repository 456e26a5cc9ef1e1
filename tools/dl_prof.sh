#!/bin/bash
# per-step kernel table of the dictionary step: tools/dl_prof.sh <tag>  (env passed through to tools/dl_trace.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-dl}
O=gpurun_out/$TAG
rm -rf $O; mkdir -p $O
export STEPS=${STEPS:-8}
python3 tools/dl_trace.py > $O/plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/dl_trace.py > $O/log.txt 2>&1
cat $O/plain.log | tail -1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
python3 tools/trace_summary.py $f $STEPS 30 > $O/kernels.txt
cat $O/kernels.txt
