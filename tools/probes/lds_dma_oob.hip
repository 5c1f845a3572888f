// Probe: does `buffer_load_dwordx4 ... lds` write ZEROS into LDS for lanes whose offset fails the
// buffer range check?  (The conv gather relies on it for the zero padding.)
// build: hipcc -O2 --offload-arch=gfx950 lds_dma_oob.hip -o lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k(const unsigned* x, unsigned bytes, u32x4* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)bytes, 0x00020000);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  *(u32x4*)(smem + threadIdx.x * 16) = (u32x4){0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  __syncthreads();
  // odd lanes: out of range
  const unsigned off = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  out[threadIdx.x] = *(u32x4*)(smem + threadIdx.x * 16);
}
int main() {
  const int n = 256 * 4;
  std::vector<unsigned> h(n);
  for (int i = 0; i < n; ++i) h[i] = i + 1;
  unsigned* d; u32x4* o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 256 * 16);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, d, (unsigned)(n * 4), o);
  std::vector<unsigned> r(n);
  hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
  int zeros = 0, stale = 0, good = 0, other = 0;
  for (int t = 0; t < 256; ++t) {
    const unsigned v = r[t * 4];
    if (t & 1) { if (v == 0) ++zeros; else if (v == 0xffffffffu) ++stale; else ++other; }
    else { if (v == (unsigned)(t * 4 + 1)) ++good; else ++other; }
  }
  printf("LDS_DMA_OOB zeros=%d stale=%d good=%d other=%d\n", zeros, stale, good, other);
  return 0;
}
