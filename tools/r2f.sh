#!/bin/bash
export TMPDIR=/tmp
bash tools/pmc_mfma.sh > gpurun_out/pmc_mfma_halo.log 2>&1; tail -30 gpurun_out/pmc_mfma_halo.log
rm -rf gpurun_out/pmc_mfma_halo; mv gpurun_out/pmc_mfma gpurun_out/pmc_mfma_halo
