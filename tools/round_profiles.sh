#!/bin/bash
# Everything the round's committed profiles come from, in one gpurun call: tools/round_profiles.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r04}
bash tools/profile_round.sh $TAG > gpurun_out/prof_${TAG}.log 2>&1
bash tools/step_tables.sh $TAG > gpurun_out/steps_${TAG}.log 2>&1
STEPS=4 bash tools/dl_pmc.sh ${TAG}_dlpmc > gpurun_out/dlpmc_${TAG}.log 2>&1
bash tools/dl_e2e_dump.sh ${TAG}_e2e > gpurun_out/e2e_${TAG}.log 2>&1
tail -3 gpurun_out/prof_${TAG}.log; tail -30 gpurun_out/steps_${TAG}.log
