// Shared device/host helpers for libvidsitu_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "vidsitu_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

void vs_set_error(const char* fmt, ...);

// Every kernel launch of the library goes through this macro: one relaxed atomic add per launch feeds
// vs_launch_count() (bench.py reports kernel launches per step from it; tests count the glue launches).
void vs_count_launch(void);
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, ...)                    \
  do {                                                         \
    vs_count_launch();                                         \
    hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__);     \
  } while (0)

#define VS_CHECK_ARG(cond, msg)                   \
  do {                                            \
    if (!(cond)) {                                \
      vs_set_error("%s: %s", __func__, msg);      \
      return VS_ERR_BAD_ARG;                      \
    }                                             \
  } while (0)

#define VS_CHECK_LAUNCH()                                                   \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      vs_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
      return VS_ERR_LAUNCH;                                                 \
    }                                                                       \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}
// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and
// keeps NaN a NaN (MI355X_MICROARCH.md, correctness boundaries).
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}
__device__ __forceinline__ void unpack8_bf16(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16);
  f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16);
  f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16);
  f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16);
  f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8_bf16(const float* f) {
  uint4 v;
  v.x = pack2_bf16(f[0], f[1]);
  v.y = pack2_bf16(f[2], f[3]);
  v.z = pack2_bf16(f[4], f[5]);
  v.w = pack2_bf16(f[6], f[7]);
  return v;
}

// One operand fragment (8 bf16) through the BN + ReLU of its producer: the arithmetic of bn_apply_cols_kernel<false,
// true> (fma, max, round to nearest even), same bits -- here as fma, ONE v_cvt_pk_bf16_f32 per pair and the ReLU on the
// rounded pair as a packed signed-16-bit max with 0 (rounding keeps the sign, so max-then-round == round-then-max;
// -0 becomes +0 either way).  _k: the 8 elements are 8 consecutive channels (sc / sh point at their constants);
// _n: one channel, 8 positions.  (No v_pk_fma_f32: see profiles/r03_linear_fused_neighbor.txt.)
typedef __attribute__((ext_vector_type(2))) float vs_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 vs_bf16x2;
typedef __attribute__((ext_vector_type(2))) short vs_s16x2;
__device__ __forceinline__ uint32_t aol_pair(float lo, float hi) {
  const vs_bf16x2 h = __builtin_convertvector((vs_f32x2){lo, hi}, vs_bf16x2);
  vs_s16x2 v = __builtin_bit_cast(vs_s16x2, h);
  v = __builtin_elementwise_max(v, (vs_s16x2){0, 0});
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ bf16x8 aol_frag_k(const bf16x8 v, const float* sc, const float* sh) {
  float f[8];
  unpack8_bf16(__builtin_bit_cast(uint4, v), f);
  const float4 s0 = *(const float4*)sc, s1 = *(const float4*)(sc + 4);
  const float4 h0 = *(const float4*)sh, h1 = *(const float4*)(sh + 4);
  uint4 o;
  o.x = aol_pair(__fmaf_rn(f[0], s0.x, h0.x), __fmaf_rn(f[1], s0.y, h0.y));
  o.y = aol_pair(__fmaf_rn(f[2], s0.z, h0.z), __fmaf_rn(f[3], s0.w, h0.w));
  o.z = aol_pair(__fmaf_rn(f[4], s1.x, h1.x), __fmaf_rn(f[5], s1.y, h1.y));
  o.w = aol_pair(__fmaf_rn(f[6], s1.z, h1.z), __fmaf_rn(f[7], s1.w, h1.w));
  return __builtin_bit_cast(bf16x8, o);
}
__device__ __forceinline__ bf16x8 aol_frag_n(const bf16x8 v, const float sc, const float sh) {
  float f[8];
  unpack8_bf16(__builtin_bit_cast(uint4, v), f);
  uint4 o;
  o.x = aol_pair(__fmaf_rn(f[0], sc, sh), __fmaf_rn(f[1], sc, sh));
  o.y = aol_pair(__fmaf_rn(f[2], sc, sh), __fmaf_rn(f[3], sc, sh));
  o.z = aol_pair(__fmaf_rn(f[4], sc, sh), __fmaf_rn(f[5], sc, sh));
  o.w = aol_pair(__fmaf_rn(f[6], sc, sh), __fmaf_rn(f[7], sc, sh));
  return __builtin_bit_cast(bf16x8, o);
}

// x = q * d + r for 0 <= x < 2^24, rcp = 1.0f / d (float quotient, two fix-ups): ~8 instructions where a runtime
// 32-bit integer division is ~40 and a 64-bit one ~100+
__device__ __forceinline__ void fast_divmod(int x, int d, float rcp, int& q, int& r) {
  q = (int)((float)x * rcp);
  r = x - q * d;
  if (r < 0) { r += d; --q; }
  if (r >= d) { r -= d; ++q; }
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
