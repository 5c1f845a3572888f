#!/bin/bash
cd /root/repo
for cfg in "VS_PW_EDBG=0" "VS_PW_EDBG=1" "VS_PW_EDBG=2" "VS_PW_EDBG=3" "VS_PW_DBG=1" "VS_PW_DBG=2"; do
  echo "== $cfg"; env VS_PW_OCC=1 $cfg timeout 300 python tools/pw_ab.py 2>&1 | grep -E "^s2.c   train|^s3.c   train|^s2.sc  train"
done
