#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-e2ehip}
rm -rf $O; mkdir -p $O
REPS=${REPS:-5} rocprofv3 --kernel-trace --hip-trace --output-format csv -d $O/kt -- python3 tools/dl_e2e.py > $O/log.txt 2>&1
grep -v amdgpu.ids $O/log.txt | grep "^rep"
python3 tools/hip_api_slow.py $(find $O/kt -name "*hip_api_trace.csv" | head -1) 1000 $(find $O/kt -name "*kernel_trace.csv" | head -1) > $O/hip_slow.txt
cat $O/hip_slow.txt
rm -rf $O/kt
