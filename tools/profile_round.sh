#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + PMC passes of bench.py.
# Usage: tools/profile_round.sh <tag>   -> gpurun_out/prof_<tag>/...
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the kernel trace profiles the DEFAULT bench command (the one the driver runs)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py > $OUT/bench_kt.log 2>&1
echo "kernel-trace rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES \
   --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_pmc_sq.log 2>&1
echo "pmc sq rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_pmc_fetch.log 2>&1
echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_pmc_write.log 2>&1
echo "pmc write rc=$?"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_LDS --output-format csv -d $OUT/pmc_lds -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_pmc_lds.log 2>&1
echo "pmc lds rc=$?"
find $OUT -name "*.csv" | head -20
