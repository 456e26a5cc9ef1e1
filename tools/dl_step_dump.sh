#!/bin/bash
# timeline of the last steps of tools/dl_trace.py: bash tools/dl_step_dump.sh <outdir> [count] [skip]   (env as dl_trace.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-stepdump}
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/dl_trace.py > $O/log.txt 2>&1
tail -1 $O/log.txt
python3 tools/trace_dump.py $(find $O/kt -name "*kernel_trace.csv" | head -1) ${2:-80} ${3:-0} > $O/dump.txt
rm -rf $O/kt
