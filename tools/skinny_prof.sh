#!/bin/bash
# kernel-trace the decode-shape GEMM microbench (VS_LSK_DBG ablations: 1 no loads, 2 no MFMA, 3 no x loads)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in ${LSK_DBGS:-0}; do
  export VS_LSK_DBG=$d
  rocprofv3 --kernel-trace -d gpurun_out/lsk$d -o lsk -- python3 tools/skinny_gemm_bench.py 50 > /dev/null 2>&1
done
