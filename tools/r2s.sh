#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r2s; rm -f gpurun_out/r2s/*
for cfg in "VS_PW_DBG=2" "VS_PW_DBG=6" "VS_PW_DBG=22" "VS_PW_DBG=54" "VS_PW_DBG=14" "VS_PW_DBG=62" "VS_PW_DBG=2 VS_PW_NSLOT=3" "VS_PW_DBG=2 VS_PW_NSLOT=5"; do
  echo "== $cfg" >> gpurun_out/r2s/pw_ab.txt
  env VS_PW_OCC=1 $cfg timeout 300 python tools/pw_ab.py 2>&1 | grep -v "amdgpu.ids\|^tiles" >> gpurun_out/r2s/pw_ab.txt
done
