// Bodies of the two ~5 us glue kernels of a unit's backward that can share ONE launch (round 3): the slab reduce
// behind a weight gradient and the BN-backward finalize of the NEXT unit are neighbours on the stream and independent
// of each other (both depend only on the pair launch in front of them).  The stand-alone kernels and the merged one
// inline these bodies: same code, same bits.  gfx950 only.
#pragma once
#include "common.h"

// dw[i] = sum_s slab[s][i], bitwise reproducible: a (virtual) 256-thread block = 16 float4 columns x 16 slab slices,
// each slice summed in order, the 16 slice sums combined in order through LDS.  t: thread within the virtual block.
__device__ __forceinline__ void wgrad_reduce_body(const float* slabs, float* dw, long long n, int S, float4 (*part)[17],
                                                  long long vb, int t) {
  const long long n4 = n >> 2;
  const int col = t & 15, sl = t >> 4;
  const long long i = vb * 16 + col;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const int per = (S + 15) / 16;
    const int s0 = sl * per, s1 = min(S, s0 + per);
    // 8 slab rows in flight per thread (a one-load-at-a-time loop is a chain of memory
    // latencies: 20 us for S = 1024); the sum order stays s0, s0+1, ...
    int s = s0;
    for (; s + 8 <= s1; s += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(slabs + (long long)(s + u) * n + i * 4);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc.x += v[u].x;
        acc.y += v[u].y;
        acc.z += v[u].z;
        acc.w += v[u].w;
      }
    }
    for (; s < s1; ++s) {
      const float4 v = *(const float4*)(slabs + (long long)s * n + i * 4);
      acc.x += v.x;
      acc.y += v.y;
      acc.z += v.z;
      acc.w += v.w;
    }
  }
  part[sl][col] = acc;
  __syncthreads();
  if (sl == 0 && i < n4) {
    float4 tt = part[0][col];
    for (int k = 1; k < 16; ++k) {
      const float4 v = part[k][col];
      tt.x += v.x;
      tt.y += v.y;
      tt.z += v.z;
      tt.w += v.w;
    }
    *(float4*)(dw + i * 4) = tt;
  }
}

// A slab reduce that waits for the next BN-backward finalize on its stream (conv_wgrad.hip owns the state).
struct VsPendingReduce {
  const float* slabs;
  float* dw;
  long long n;
  int S;
  hipStream_t st;
};
// true + *out filled when a reduce is pending on `st` (the caller then owns it and must launch it)
bool vs_pending_reduce_take(hipStream_t st, VsPendingReduce* out);
