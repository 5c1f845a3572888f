#!/bin/bash
# Diagnostic build for tools/launch_anatomy.py: the convolution kernels with s_memrealtime phase stamps (-DVS_STAMP,
# csrc/conv_tile.h).  ConvP grows by one pointer, so every translation unit that sees it is rebuilt; the rest of the
# library is linked from the shipped objects.  Output: tmp/stamp/libvidsitu_hip.so (never the product path).
set -e
cd "$(dirname "$0")/../vidsitu_amd/csrc"
make -j4 >/dev/null
OUT=../../tmp/stamp; mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize -DVS_STAMP"
for f in conv_pair conv_halo conv_pw conv_deep; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $OUT/$f.o & done; wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OUT/conv_pair.o $OUT/conv_halo.o $OUT/conv_pw.o $OUT/conv_deep.o \
  bn_pool.o conv_stem.o gpt2_ops.o resize_u8.o txenc_ops.o -o $OUT/libvidsitu_hip.so
ls -la $OUT/libvidsitu_hip.so
