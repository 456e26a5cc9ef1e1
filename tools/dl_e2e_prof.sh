#!/bin/bash
# kernel-trace timeline of decomp_amd.dictionary_learning.solve end to end: tools/dl_e2e_prof.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-e2e}
O=gpurun_out/$TAG
rm -rf $O; mkdir -p $O
R=${REPS:-8}
REPS=$R rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/dl_e2e.py > $O/log.txt 2>&1
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 tools/e2e_timeline.py $f ${FRAC:-0.08} 24 > $O/timeline.txt
grep -v amdgpu.ids $O/log.txt | tail -12; head -60 $O/timeline.txt
rm -rf $O/kt
