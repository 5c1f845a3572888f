// Probe: what does one CU take in through `buffer_load_dwordx4 ... lds` (the conv kernels' fill path)?
// Every block runs a ring of SLOTS 1-KiB pieces per wave and keeps DEPTH of them in flight (counted vmcnt);
// the source is a region of `span` bytes that every block walks from a block-dependent start, so span picks the
// level that serves it (1 MiB: the XCD's L2, 64 MiB: Infinity Cache, 2 GiB: HBM).  Row shape: a piece is eight
// 128-byte rows `pitch` bytes apart (pitch 128 = one contiguous KiB; 2048 = the gather of a 1024-channel tensor).
// Prints GB/s per CU and chip-wide for waves per block x depth.
// build: hipcc -O2 --offload-arch=gfx950 lds_dma_rate.hip -o lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int DEPTH>
__global__ void k(const char* x, unsigned span_mask, unsigned pitch, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int nw = blockDim.x >> 6;
  typedef __attribute__((address_space(3))) char* lds_ptr_t;
  const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem + (unsigned)wave * (DEPTH + 1) * 1024u;
  const unsigned long a = (unsigned long)x;
  const i32x4 desc = {(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)(span_mask + 1u), 0x00020000};
  // lane l fetches 16-byte unit (l & 7) of row (l >> 3) of the piece
  const unsigned lane_off = (unsigned)(lane >> 3) * pitch + (unsigned)(lane & 7) * 16u;
  // consecutive pieces of a wave: the next 64-channel chunk of the same 8 rows (pitch > 128) or the next KiB
  const unsigned step = pitch > 128 ? 128u : 1024u;
  const unsigned row_block = pitch * 8u;                     // bytes covered by 8 rows
  const unsigned per_row_steps = pitch > 128 ? pitch / 128u : 1u;
  unsigned base = ((unsigned)blockIdx.x * (unsigned)nw + (unsigned)wave) * 8u * (pitch > 128 ? pitch : 128u);
  unsigned sub = 0;
  auto issue = [&](int slot) {
    const unsigned off = (base + sub * step + lane_off) & span_mask;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :
                 : "s"(lds0 + (unsigned)slot * 1024u), "v"(off), "s"(desc)
                 : "memory");
    if (++sub == per_row_steps) {
      sub = 0;
      base += row_block * (unsigned)nw * gridDim.x;
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d);
  int slot = DEPTH;
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(DEPTH - 1) : "memory");
    issue(slot);
    slot = slot == DEPTH ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = *(unsigned*)smem;
}

template <int DEPTH>
static float run(const char* x, unsigned span, unsigned pitch, int waves, int blocks, int iters, unsigned* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t lds = (size_t)waves * (DEPTH + 1) * 1024;
  hipFuncSetAttribute((const void*)k<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  float best = 1e30f;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<DEPTH>, dim3(blocks), dim3(waves * 64), lds, 0, x, span - 1, pitch, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  return best;
}

int main() {
  const size_t big = 2ull << 30;
  char* x;
  unsigned* sink;
  if (hipMalloc(&x, big) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&sink, 4096 * 4);
  hipMemset(x, 1, big);
  const int iters = 2000;
  printf("%-10s %-6s %-6s %-6s %10s %10s\n", "span", "pitch", "waves", "depth", "GB/s/CU", "TB/s chip");
  const unsigned spans[3] = {1u << 20, 64u << 20, 1u << 31};
  const unsigned pitches[2] = {128u, 2048u};
  for (unsigned span : spans)
    for (unsigned pitch : pitches)
      for (int waves : {4, 8, 16})
        for (int depth : {2, 4, 8, 16}) {
          if ((size_t)waves * (depth + 1) * 1024 > 160 * 1024) continue;
          const int blocks = 256;
          float ms = 0.f;
          if (depth == 2) ms = run<2>(x, span, pitch, waves, blocks, iters, sink);
          if (depth == 4) ms = run<4>(x, span, pitch, waves, blocks, iters, sink);
          if (depth == 8) ms = run<8>(x, span, pitch, waves, blocks, iters, sink);
          if (depth == 16) ms = run<16>(x, span, pitch, waves, blocks, iters, sink);
          const double bytes = (double)blocks * waves * (iters + depth) * 1024.0;
          printf("%-10u %-6u %-6d %-6d %10.1f %10.2f\n", span, pitch, waves, depth, bytes / ms / 1e6 / blocks,
                 bytes / ms / 1e9);
        }
  return 0;
}
