#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-e2edump}
rm -rf $O; mkdir -p $O
REPS=${REPS:-4} rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/dl_e2e.py > $O/log.txt 2>&1
grep "^rep" $O/log.txt
python3 tools/trace_dump.py $(find $O/kt -name "*kernel_trace.csv" | head -1) 170 200 > $O/dump.txt
rm -rf $O/kt
