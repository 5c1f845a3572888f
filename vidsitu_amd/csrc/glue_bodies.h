// Bodies of the two ~5 us glue kernels of a unit's backward that can share ONE launch (round 3): the slab reduce
// behind a weight gradient and the BN-backward finalize of the NEXT unit are neighbours on the stream and independent
// of each other (both depend only on the pair launch in front of them).  The stand-alone kernels and the merged one
// inline these bodies: same code, same bits.  gfx950 only.
#pragma once
#include "common.h"

// Virtual 256-thread blocks of the reduce below (every launch site sizes its grid with this).
static inline long long wgrad_reduce_vblocks(long long n, int S) {
  const long long n4 = n >> 2;
  return S <= 16 ? (n4 + 255) / 256 : (n4 + 15) / 16;
}

// dw[i] = sum_s slab[s][i], bitwise reproducible: a (virtual) 256-thread block = 16 float4 columns x 16 slab slices,
// each slice summed in order, the 16 slice sums combined in order through LDS.  t: thread within the virtual block.
// S <= 16 (every convolution layer's position split; only the stems split deeper): one slab per slice, so the sum is
// ((0 + s0) + (0 + s1)) + ... -- formed by ONE thread per float4 column, 256 columns per virtual block, all S loads in
// flight, no LDS: the same additions in the same order (bit for bit the slice form, signs of zero included), at 16
// times the bytes per block.  (Round 4: the slice form left 256 - 16 S threads of a block idle and fetched 256 bytes
// per slab per block: 99 launches per step at 1.5 TB/s.)
__device__ __forceinline__ void wgrad_reduce_body(const float* slabs, float* dw, long long n, int S, float4 (*part)[17],
                                                  long long vb, int t) {
  const long long n4 = n >> 2;
  if (S <= 16) {
    const long long i = vb * 256 + t;
    if (i >= n4) return;
    const float* src = slabs + i * 4;
    float4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (u < S) v[u] = *(const float4*)(src + (long long)u * n);
    float4 tt = make_float4(0.f + v[0].x, 0.f + v[0].y, 0.f + v[0].z, 0.f + v[0].w);
#pragma unroll
    for (int u = 1; u < 16; ++u)
      if (u < S) {
        tt.x += 0.f + v[u].x;
        tt.y += 0.f + v[u].y;
        tt.z += 0.f + v[u].z;
        tt.w += 0.f + v[u].w;
      }
    if (S < 16) {  // the slice form adds the empty slices' +0.0 (turns a -0.0 sum into +0.0)
      tt.x += 0.f;
      tt.y += 0.f;
      tt.z += 0.f;
      tt.w += 0.f;
    }
    *(float4*)(dw + i * 4) = tt;
    return;
  }
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
