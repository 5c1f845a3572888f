#!/bin/bash
# Matrix-core busy + stall counters of the top conv GEMMs (forward / dgrad / wgrad) at the bench shapes.
# Separate --pmc passes, no trace domain besides --kernel-trace, the program itself after `--`
# (MI355X_MICROARCH.md, rocprofv3 PMC slots: SQ 8 per pass, GRBM 2 independent).
#   bash tools/pmc_mfma.sh   -> gpurun_out/pmc_mfma/pmc_mfma.json
export TMPDIR=/tmp
OUT=gpurun_out/pmc_mfma
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 tools/pmc_layers.py $OUT/plan.json 3 > $OUT/sq.log 2>&1
echo "sq pass exit $?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/sq2 -- python3 tools/pmc_layers.py $OUT/plan.json 3 > $OUT/sq2.log 2>&1
echo "sq2 pass exit $?"
python3 tools/pmc_mfma.py $OUT "${VS_BUILD_TAG:-untagged}"
find $OUT -name "*kernel_trace*.csv" -delete
find $OUT -name "*counter_collection*.csv" -size +30M -delete
