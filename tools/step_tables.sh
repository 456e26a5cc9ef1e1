#!/bin/bash
# per-step rocprofv3 kernel tables of the secondary steps (run through gpurun): tools/step_tables.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r03}
STEPS=8 tools/dl_prof.sh ${TAG}_dl_f32_ista > /dev/null 2>&1
STEPS=8 METHOD=cd tools/dl_prof.sh ${TAG}_dl_f32_cd > /dev/null 2>&1
STEPS=6 CPLX=1 tools/dl_prof.sh ${TAG}_dl_c64_ista > /dev/null 2>&1
O=gpurun_out/${TAG}_shard; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/shard_trace.py > $O/log.txt 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
python3 tools/trace_summary.py $f 104 14 > $O/kernels.txt
for d in ${TAG}_dl_f32_ista ${TAG}_dl_f32_cd ${TAG}_dl_c64_ista ${TAG}_shard; do echo "== $d"; tail -1 gpurun_out/$d/plain.log 2>/dev/null; head -24 gpurun_out/$d/kernels.txt; done
