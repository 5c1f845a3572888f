// v_dot2c_f32_bf16 / v_fma_f32 issue rate on gfx950: one wave per SIMD and four, independent accumulators.
// hipcc -O3 --offload-arch=gfx950 tools/probes/dot2_rate.hip -o /tmp/dot2_rate && /tmp/dot2_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
template <int KIND>
__global__ __launch_bounds__(256) void k(const unsigned* a, float* out, int iters) {
  unsigned x0 = a[threadIdx.x], x1 = a[threadIdx.x + 64], x2 = a[threadIdx.x + 128], x3 = a[threadIdx.x + 192];
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (float)i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const unsigned xa = (i & 1) ? x0 : x1, xb = (i & 2) ? x2 : x3;
      if (KIND == 0) acc[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, xa), __builtin_bit_cast(bf16x2, xb), acc[i], false);
      else acc[i] = __builtin_fmaf(__uint_as_float(xa), __uint_as_float(xb), acc[i]);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  unsigned* a; float* o;
  hipMalloc(&a, 4096); hipMemset(a, 0x3f, 4096); hipMalloc(&o, 4 * 256 * 4096);
  for (int kind = 0; kind < 2; ++kind)
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
      const int grid = 256 * bpc, iters = 4096;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, a, o, iters);
        else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, a, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double inst_per_simd = (double)iters * 16 * bpc;  // wave-instructions per SIMD (one wave per SIMD per block)
      printf("%s  %d block(s)/CU: %.3f ms, %.2f ns per wave-instruction per SIMD (~%.1f cycles at 2.4 GHz)\n",
             kind == 0 ? "v_dot2c_f32_bf16" : "v_fma_f32       ", bpc, ms, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    }
  return 0;
}
