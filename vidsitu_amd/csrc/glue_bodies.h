// Bodies of the two ~5 us glue kernels of a unit's backward that can share ONE launch (round 3): the slab reduce
// behind a weight gradient and the BN-backward finalize of the NEXT unit are neighbours on the stream and independent
// of each other (both depend only on the pair launch in front of them).  The stand-alone kernels and the merged one
// inline these bodies: same code, same bits.  gfx950 only.
#pragma once
#include "common.h"

// A slab row is read exactly once (by this reduce) after the weight-gradient kernel wrote it: non-temporal loads keep
// the 20-30 MB of a wide layer's slabs from evicting what the kernels behind the reduce read (profiles/r04_nontemporal.txt).
__device__ __forceinline__ float4 ld_slab4(const float* p) {
  typedef __attribute__((ext_vector_type(4))) float f32x4_t;
  const f32x4_t v = __builtin_nontemporal_load((const f32x4_t*)p);
  return make_float4(v[0], v[1], v[2], v[3]);
}

// dW is read next by the optimizer at the end of the step: a non-temporal store.
__device__ __forceinline__ void st_dw4(float* p, float4 v) {
  typedef __attribute__((ext_vector_type(4))) float f32x4_t;
  const f32x4_t t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, (f32x4_t*)p);
}

// The reduce below has two thread mappings with the SAME arithmetic; this picks one (host and device agree through it).
// Column form: one thread per float4 column walks all S slabs -- whole 4-KiB rows per wave, no LDS.  It needs columns
// to spread over the chip, so: every split of <= 16 (each convolution layer of the slow pathway) and splits up to 64
// whose dW has >= 32 768 float4 columns (slow s4's c layers: S = 22, 23 MB of slabs).
__host__ __device__ __forceinline__ bool wgrad_reduce_cols(long long n, int S) {
  return S <= 16 || (S <= 64 && (n >> 2) >= 32768);
}
// Virtual 256-thread blocks of the reduce (every launch site sizes its grid with this).
static inline long long wgrad_reduce_vblocks(long long n, int S) {
  const long long n4 = n >> 2;
  return wgrad_reduce_cols(n, S) ? (n4 + 255) / 256 : (n4 + 15) / 16;
}

// dw[i] = sum_s slab[s][i], bitwise reproducible.  The sum is defined by the slice form: 16 slices of per = ceil(S / 16)
// consecutive slabs, each slice summed in order starting from +0.0, the 16 slice sums added in order.
//   slice form  (few columns, many slabs: the fast pathway's S = 79 .. 349 over 1 .. 12 K floats): a (virtual) 256-thread
//               block = 16 float4 columns x 16 slices, combined through LDS; a slice's loads go out 16 at a time, the
//               tail predicated (round 4: the tail was a loop of up to 7 dependent load -> add round trips);
//   column form (wgrad_reduce_cols): one thread forms the 16 slice sums of its column one after the other and adds
//               them -- the same additions in the same order, signs of zero included (the batched slice-form kernel of
//               conv_wgrad.hip is the tests' bitwise reference), 16 loads in flight, no LDS, no barrier.
// Round 4, measured: the slice form at S <= 22 left most of a block idle and moved 256 bytes per slab per block --
// 99 launches per step at 1.5 TB/s, the largest in-situ kernel time of the train step.
// t: thread within the virtual block.
__device__ __forceinline__ void wgrad_reduce_body(const float* slabs, float* dw, long long n, int S, float4 (*part)[17],
                                                  long long vb, int t) {
  const long long n4 = n >> 2;
  const int per = (S + 15) / 16;
  if (wgrad_reduce_cols(n, S)) {
    const long long i = vb * 256 + t;
    if (i >= n4) return;
    const float* src = slabs + i * 4;
    float4 tt = make_float4(0.f, 0.f, 0.f, 0.f), acc = make_float4(0.f, 0.f, 0.f, 0.f);
    bool have = false;  // tt holds the sum of the finished slices
    int left = per;     // slabs the current slice still takes
    for (int s0 = 0; s0 < S; s0 += 16) {
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (s0 + u < S) v[u] = ld_slab4(src + (long long)(s0 + u) * n);
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (s0 + u < S) {
          if (left == 0) {  // the previous slice is complete
            if (have) { tt.x += acc.x; tt.y += acc.y; tt.z += acc.z; tt.w += acc.w; }
            else tt = acc;
            have = true;
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
            left = per;
          }
          acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
          --left;
        }
    }
    if (have) { tt.x += acc.x; tt.y += acc.y; tt.z += acc.z; tt.w += acc.w; }
    else tt = acc;
    if ((S + per - 1) / per < 16) {  // the slice form adds the empty slices' +0.0 (turns a -0.0 sum into +0.0)
      tt.x += 0.f; tt.y += 0.f; tt.z += 0.f; tt.w += 0.f;
    }
    st_dw4(dw + i * 4, tt);
    return;
  }
  const int col = t & 15, sl = t >> 4;
  const long long i = vb * 16 + col;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const int s0 = sl * per, s1 = min(S, s0 + per);
    const float* src = slabs + i * 4;
    for (int s = s0; s < s1; s += 16) {
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {  // branch-free: a clamped address, the value dropped below
        const int su = min(s + u, s1 - 1);
        v[u] = ld_slab4(src + (long long)su * n);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (s + u < s1) {
          acc.x += v[u].x;
          acc.y += v[u].y;
          acc.z += v[u].z;
          acc.w += v[u].w;
        }
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
    st_dw4(dw + i * 4, tt);
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
