#!/bin/bash
# PMC passes (separate runs, kernel counters only) of the dictionary step: tools/dl_pmc.sh <tag> ; env passed to tools/dl_trace.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-dlpmc}
O=gpurun_out/$TAG
rm -rf $O; mkdir -p $O
export STEPS=${STEPS:-4}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES \
   --output-format csv -d $O/pmc_sq -- python3 tools/dl_trace.py > $O/pmc_sq.log 2>&1
echo "pmc sq rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/dl_trace.py > $O/pmc_fetch.log 2>&1
echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/dl_trace.py > $O/pmc_write.log 2>&1
echo "pmc write rc=$?"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD \
   --output-format csv -d $O/pmc_lds -- python3 tools/dl_trace.py > $O/pmc_lds.log 2>&1
echo "pmc lds rc=$?"
python3 tools/make_pmc_summary.py $O $O/summary.json
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/pmc_lds
