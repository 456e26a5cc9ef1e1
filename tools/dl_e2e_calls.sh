#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/e2ecalls
rm -rf $O; mkdir -p $O
REPS=3 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/dl_e2e.py > $O/log.txt 2>&1
grep "^rep" $O/log.txt
python3 tools/e2e_calls.py $(find $O/kt -name "*kernel_trace.csv" | head -1) > $O/calls.txt
cat $O/calls.txt
rm -rf $O/kt
