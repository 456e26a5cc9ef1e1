#!/bin/bash
# the three PMC sections of profiles/<round>_pmc_dictionary_step.json: tools/dl_pmc_all.sh <tag>
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r04}
STEPS=4 bash tools/dl_pmc.sh ${TAG}_pmc_f32_ista > gpurun_out/${TAG}_pmc_f32_ista.log 2>&1
STEPS=4 METHOD=cd bash tools/dl_pmc.sh ${TAG}_pmc_f32_cd > gpurun_out/${TAG}_pmc_f32_cd.log 2>&1
STEPS=3 CPLX=1 bash tools/dl_pmc.sh ${TAG}_pmc_c64_ista > gpurun_out/${TAG}_pmc_c64_ista.log 2>&1
python3 - <<PY
import json
out = {}
for k in ('f32_ista', 'f32_cd', 'c64_ista'):
    out[k] = json.load(open('gpurun_out/${TAG}_pmc_%s/summary.json' % k))
json.dump(out, open('gpurun_out/${TAG}_pmc_dictionary_step.json', 'w'), indent=1, sort_keys=True)
print({k: len(v) for k, v in out.items()})
PY
