#!/bin/bash
# A variant of the library with extra compiler flags for the convolution translation units (build-time experiments):
#   tools/build_variant_lib.sh NAME "-DVS_YST=1"   -> tmp/NAME/libvidsitu_hip.so   (use with VS_LIB_PATH / tools/ab_lib.sh)
set -e
NAME=$1; EXTRA=$2
cd "$(dirname "$0")/../vidsitu_amd/csrc"
make -j4 >/dev/null
OUT=../../tmp/$NAME; mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize $EXTRA"
for f in conv_pair conv_halo conv_pw conv_deep; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $OUT/$f.o & done; wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OUT/conv_pair.o $OUT/conv_halo.o $OUT/conv_pw.o $OUT/conv_deep.o \
  bn_pool.o conv_stem.o gpt2_ops.o resize_u8.o txenc_ops.o -o $OUT/libvidsitu_hip.so
rm -f $OUT/*.o
ls -la $OUT/libvidsitu_hip.so
