#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3x; mkdir -p $OUT
for cfg in "- -" "1024 4" "4096 4" "8192 4" "2048 1" "2048 2" "2048 8" "4096 2" "8192 1"; do
  set -- $cfg
  ( [ "$1" != "-" ] && export VS_BN_TARGET=$1; [ "$2" != "-" ] && export VS_BN_NBMAX=$2; timeout 300 python tools/bn_time.py ) >> $OUT/bn_time.log 2>&1
done
grep -E "VS_BN|sum x" $OUT/bn_time.log
