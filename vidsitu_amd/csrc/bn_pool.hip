// BatchNorm3d (train + eval), stem max-pools, global average pool and the input
// pack for the channels-last bf16 trunk.  All HBM-bound: 16-byte vectors of 8
// consecutive channels per lane, fp32 math, wave-shuffle / LDS reductions.
// Reference call sites: vidsitu_code/mdl_sf_base.py:22-33 (trunk modules),
// :97-113 (AdaptiveAvgPool3d + cat), dat_loader.py:454-501 (input contract).
#include <stdlib.h>

#include "common.h"
#include "glue_bodies.h"

// ----------------------------------------------------------------------------
// NCDHW (f32 | bf16) -> NDHWC bf16, channels zero-padded to Cpad (multiple of 8)
// ----------------------------------------------------------------------------
template <bool IN_BF16>
__global__ void pack_input_kernel(const void* xin, uint16_t* y, int N, int C, long long THW,
                                  int Cpad) {
  const long long pos = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // n*THW + s
  if (pos >= (long long)N * THW) return;
  const long long n = pos / THW, s = pos - n * THW;
  if (Cpad == 4) {  // stem layout: 8 bytes per pixel
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C && c < 4; ++c) {
      const long long off = (n * C + c) * THW + s;
      v[c] = IN_BF16 ? bf16_to_f32(((const uint16_t*)xin)[off]) : ((const float*)xin)[off];
    }
    *(uint2*)(y + pos * 4) = make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]));
    return;
  }
  for (int c0 = 0; c0 < Cpad; c0 += 8) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + e;
      float f = 0.f;
      if (c < C) {
        const long long off = (n * C + c) * THW + s;
        f = IN_BF16 ? bf16_to_f32(((const uint16_t*)xin)[off]) : ((const float*)xin)[off];
      }
      v[e] = f;
    }
    *(uint4*)(y + pos * Cpad + c0) = pack8_bf16(v);
  }
}

// Stem layout (Cpad = 4), 8 consecutive pixels per thread: one 16-byte (bf16) or two 16-byte (f32) loads
// per channel plane and four 16-byte stores -- the one-pixel form moved 2 bytes per lane and load
// (0.84 TB/s on the 32 x 224 x 224 fast-pathway input).
template <bool IN_BF16>
__global__ void pack_input_c4x8_kernel(const void* xin, uint16_t* y, int N, int C, long long THW8) {
  // grid = (chunks of a clip, clips): no division (a 64-bit one per thread was a third of this kernel's instructions)
  const long long n = blockIdx.y, s8 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s8 >= THW8) return;
  const long long i = n * THW8 + s8;
  float v[3][8];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (c < C) {
      const long long off = ((n * C + c) * THW8 + s8) * 8;
      if (IN_BF16) {
        unpack8_bf16(*(const uint4*)((const uint16_t*)xin + off), v[c]);
      } else {
        const float4 a = *(const float4*)((const float*)xin + off), b = *(const float4*)((const float*)xin + off + 4);
        v[c][0] = a.x; v[c][1] = a.y; v[c][2] = a.z; v[c][3] = a.w;
        v[c][4] = b.x; v[c][5] = b.y; v[c][6] = b.z; v[c][7] = b.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
    }
  }
  uint4* o = (uint4*)(y + i * 32);
#pragma unroll
  for (int e = 0; e < 8; e += 2)
    o[e >> 1] = make_uint4(pack2_bf16(v[0][e], v[1][e]), pack2_bf16(v[2][e], 0.f),
                           pack2_bf16(v[0][e + 1], v[1][e + 1]), pack2_bf16(v[2][e + 1], 0.f));
}

extern "C" int vs_pack_input(const void* x, int x_is_bf16, void* y, int N, int C, int T, int H,
                             int W, int Cpad, void* stream) {
  VS_CHECK_ARG(x && y, "null tensor");
  VS_CHECK_ARG((Cpad % 8 == 0 || Cpad == 4) && Cpad >= C, "Cpad must be 4 or a multiple of 8, >= C");
  const long long THW = (long long)T * H * W, total = THW * N;
  if (Cpad == 4 && C <= 3 && (THW & 7) == 0 && N <= 65535 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
    const dim3 g8((unsigned)((THW / 8 + 255) / 256), (unsigned)N);
    if (x_is_bf16)
      hipLaunchKernelGGL(pack_input_c4x8_kernel<true>, g8, dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)y, N,
                         C, THW / 8);
    else
      hipLaunchKernelGGL(pack_input_c4x8_kernel<false>, g8, dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)y,
                         N, C, THW / 8);
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (x_is_bf16)
    hipLaunchKernelGGL(pack_input_kernel<true>, grid, block, 0, (hipStream_t)stream, x,
                       (uint16_t*)y, N, C, THW, Cpad);
  else
    hipLaunchKernelGGL(pack_input_kernel<false>, grid, block, 0, (hipStream_t)stream, x,
                       (uint16_t*)y, N, C, THW, Cpad);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ----------------------------------------------------------------------------
// uint8 frames -> normalised bf16 channels-last activations (second half).
// The reference builds its input on the CPU (vidsitu_code/dat_loader.py:183-191,454-501):
// PIL RGB 224x224 uint8 [T,H,W,3] -> float()/255 -> (x - mean)/std (utils/video_utils.py:147-164)
// -> permute to C,T,H,W -> slow pathway = index_select(T, linspace(0,T-1,T/alpha).long())
// (:59-65) -> .float() -> H2D of fp32 (96 MB per video).  Here the uint8 frames are uploaded
// (4x fewer PCIe bytes) and ONE kernel per pathway produces the stem's packed layout: same fp32
// operation order ((x/255 - mean) / std, IEEE division), then the bf16 rounding every activation
// gets anyway -- bit-identical to vs_pack_input of the reference's fp32 tensor.
// ----------------------------------------------------------------------------
__global__ void frames_u8_pack_kernel(const uint8_t* fr, const int* tidx, uint16_t* y, int N, int Tin,
                                      int Tout, long long HW, int Cpad, float m0, float m1, float m2,
                                      float s0, float s1, float s2, int reverse) {
  const long long pos = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (n, to, s)
  const long long per_n = (long long)Tout * HW;
  if (pos >= (long long)N * per_n) return;
  const long long n = pos / per_n, r = pos - n * per_n;
  const int to = (int)(r / HW);
  const long long sp = r - (long long)to * HW;
  const int ti = tidx ? tidx[to] : to;
  const uint8_t* px = fr + (((long long)n * Tin + ti) * HW + sp) * 3;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float f = (float)px[reverse ? 2 - c : c] / 255.0f;  // tensor.float() / 255.0
    v[c] = (f - mean[c]) / sd[c];                              // (tensor - mean) / std
  }
  if (Cpad == 4) {
    *(uint2*)(y + pos * 4) = make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]));
  } else {
    *(uint4*)(y + pos * Cpad) = pack8_bf16(v);
    for (int c0 = 8; c0 < Cpad; c0 += 8) *(uint4*)(y + pos * Cpad + c0) = make_uint4(0u, 0u, 0u, 0u);
  }
}

extern "C" int vs_frames_u8_pack(const uint8_t* frames, const int* t_index, void* y, int N, int Tin,
                                 int Tout, int H, int W, int Cpad, const float* mean3,
                                 const float* std3, int reverse_channels, void* stream) {
  VS_CHECK_ARG(frames && y && mean3 && std3, "null argument (mean3 / std3 are host pointers)");
  VS_CHECK_ARG((Cpad % 8 == 0 || Cpad == 4) && Cpad >= 3, "Cpad must be 4 or a multiple of 8");
  VS_CHECK_ARG(N > 0 && Tin > 0 && Tout > 0 && (t_index || Tout == Tin), "Tout != Tin needs t_index");
  const long long HW = (long long)H * W, total = (long long)N * Tout * HW;
  hipLaunchKernelGGL(frames_u8_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, frames, t_index, (uint16_t*)y, N, Tin, Tout, HW, Cpad,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], reverse_channels);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ----------------------------------------------------------------------------
// BN finalize: partial[nparts][2][C] -> mean / biased var -> scale, shift
// block = 32 channels x 32 slices; fp64 cross-partial accumulation.
// ----------------------------------------------------------------------------
// agent-scope relaxed 8-byte stores / loads (`global_store_dwordx2 ... sc1` / `global_load_dwordx2 ... sc1`): cross-block
// hand-off inside one launch without a cache-wide fence (MI355X_MICROARCH.md, "hand-offs measured with sc1 loads")
__device__ __forceinline__ void st_part64(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_part64(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The arithmetic from (sum, sum of squares) to the affine form, shared by the finalize kernel and by the apply
// kernels that finalize for themselves (below).  Every multiply-add is spelled out: left to the compiler's
// contraction the same source rounds differently in different kernels, and the two paths are meant to be bitwise.
__device__ __forceinline__ void bn_affine_from_sums(double ts, double tq, double count, float ga, float be, float eps,
                                                    float& sc, float& sf, float& mean_f, float& invstd, double& var) {
  const double mean = ts / count;
  var = fma(-mean, mean, tq / count);
  if (var < 0.0) var = 0.0;
  invstd = (float)(1.0 / sqrt(var + (double)eps));
  sc = __fmul_rn(ga, invstd);
  mean_f = (float)mean;
  sf = __fmaf_rn(-mean_f, sc, be);
}
__device__ __forceinline__ void bn_running_update(float rm, float rv, float mean_f, double var, double count,
                                                  float momentum, float& rm_new, float& rv_new) {
  const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
  rm_new = __fmaf_rn(momentum, mean_f, __fmul_rn(1.f - momentum, rm));
  rv_new = __fmaf_rn(momentum, (float)unb, __fmul_rn(1.f - momentum, rv));
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(
    const float* partials, int nparts, double count, const float* gamma, const float* beta,
    float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
    float* mean_out, float* invstd_out, int C) {
  __shared__ double sh_s[32][33];
  __shared__ double sh_q[32][33];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  // per-channel parameters requested up front (branch-free, clamped), so that their latency runs beside
  // the partial sums' instead of after the reduction
  const int cc = c < C ? c : C - 1;
  const float ga_c = gamma[cc], be_c = beta[cc];
  const float rm_c = (running_mean ? running_mean : gamma)[cc];
  const float rv_c = (running_var ? running_var : gamma)[cc];
  double s = 0.0, q = 0.0;
  if (c < C) {
    int p = sl;
    for (; p + 96 < nparts; p += 128) {  // 8 loads in flight, same summation order
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = partials[(long long)(p + 32 * u) * 2 * C + c];
        b[u] = partials[(long long)(p + 32 * u) * 2 * C + C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s += (double)a[u];
        q += (double)b[u];
      }
    }
    for (; p < nparts; p += 32) {
      s += (double)partials[(long long)p * 2 * C + c];
      q += (double)partials[(long long)p * 2 * C + C + c];
    }
  }
  sh_s[sl][cl] = s;
  sh_q[sl][cl] = q;
  __syncthreads();
  if (sl == 0 && c < C) {
    float sc, sf;
    if (nparts > 0) {
      double ts = 0.0, tq = 0.0;
      for (int i = 0; i < 32; ++i) {
        ts += sh_s[i][cl];
        tq += sh_q[i][cl];
      }
      float mean_f, invstd;
      double var;
      bn_affine_from_sums(ts, tq, count, ga_c, be_c, eps, sc, sf, mean_f, invstd, var);
      if (mean_out) mean_out[c] = mean_f;
      if (invstd_out) invstd_out[c] = invstd;
      if (running_mean) {
        float rm_n, rv_n;
        bn_running_update(rm_c, rv_c, mean_f, var, count, momentum, rm_n, rv_n);
        running_mean[c] = rm_n;
        running_var[c] = rv_n;
      }
    } else {  // eval: fold the running statistics
      const float invstd = 1.0f / sqrtf(rv_c + eps);
      sc = ga_c * invstd;
      sf = be_c - rm_c * sc;
      if (mean_out) mean_out[c] = rm_c;
      if (invstd_out) invstd_out[c] = invstd;
    }
    scale[c] = sc;
    shift[c] = sf;
  }
}

// Level-1 reduction of many per-tile partials: out[g][2][C] = sum over the g-th slice of
// parts (fp64 accumulation, fixed order).  Keeps bn_finalize short when a small-channel
// layer produced thousands of partial rows.
__global__ __launch_bounds__(1024) void bn_partials_reduce_kernel(const float* partials, int nparts,
                                                                 float* out, int C, int G) {
  __shared__ double sh_s[32][33];
  __shared__ double sh_q[32][33];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int g = blockIdx.y;
  const int per = (nparts + G - 1) / G;
  const int p0 = g * per, p1 = min(nparts, p0 + per);
  double s = 0.0, q = 0.0;
  if (c < C) {
    for (int p = p0 + sl; p < p1; p += 32) {
      s += (double)partials[(long long)p * 2 * C + c];
      q += (double)partials[(long long)p * 2 * C + C + c];
    }
  }
  sh_s[sl][cl] = s;
  sh_q[sl][cl] = q;
  __syncthreads();
  if (sl == 0 && c < C) {
    double ts = 0.0, tq = 0.0;
    for (int i = 0; i < 32; ++i) {
      ts += sh_s[i][cl];
      tq += sh_q[i][cl];
    }
    out[(long long)g * 2 * C + c] = (float)ts;
    out[(long long)g * 2 * C + C + c] = (float)tq;
  }
}

// ----------------------------------------------------------------------------
// Two-level finalize in ONE launch (round 5).  Layers whose producer wrote hundreds to thousands of partial rows (s2 / s3
// and the whole fast pathway: 392 .. 3136 rows) took a level-1 reduce launch + the finalize launch in the forward pass
// (53 + 110 launches per SlowFast-R50 step) and, in the backward pass, one finalize launch whose 32 slices walked up to
// 98 rows each, 8 loads at a time (9.8 us per launch in the replayed step).  Here grid = (C / 32) x G: block (bx, g)
// sums rows [g * rpg, (g + 1) * rpg) of its 32 channels -- thread (slice, channel) takes rows slice + 32 u, all of a
// batch of 8 in flight -- and, with G > 1, stores its sums (agent-scope stores), draws a ticket from the channel
// group's counter, and the block that draws the last ticket adds the G group sums IN GROUP ORDER and closes the
// channels.  A ticket costs every block one store drain + one atomic round trip (~1.5 us): cheap in a kernel whose
// blocks do nothing else -- unlike at the end of every convolution tile (profiles/r05_fin_inlaunch.txt).  The sums
// are bitwise stable (fixed order whoever arrives last); counters are zero between launches.
// ----------------------------------------------------------------------------
struct Fin2P {
  const float* part;
  int nparts, C, G, rpg;  // rows, channels, groups (<= 32), rows per group (multiple of 32)
  double* lvl1;           // [G][2][C]
  int* cnt;               // [ceil(C / 32)] arrival counters
  int kind;               // 1: BN forward, 2: BN backward
  double count;
  float momentum, eps;
  const float *gamma, *beta;
  float *rmean, *rvar, *scale, *shift, *mean, *invstd;
  float *dgamma, *dbeta;
};

#define VS_FIN2_CNT_BYTES 4096
#define VS_FIN2_LVL1_BYTES (32 * 2 * 4096 * 8)

extern "C" size_t vs_bn_finalize_workspace_bytes(void) { return VS_FIN2_CNT_BYTES + VS_FIN2_LVL1_BYTES; }

__device__ __forceinline__ void fin2_close(const Fin2P& f, int c, double ts, double tq, float ga_c, float be_c, float rm_c,
                                           float rv_c) {
  if (f.kind == 1) {
    float sc, sf, mean_f, invstd;
    double var;
    bn_affine_from_sums(ts, tq, f.count, ga_c, be_c, f.eps, sc, sf, mean_f, invstd, var);
    if (f.mean) f.mean[c] = mean_f;
    if (f.invstd) f.invstd[c] = invstd;
    if (f.rmean) {
      float rm_n, rv_n;
      bn_running_update(rm_c, rv_c, mean_f, var, f.count, f.momentum, rm_n, rv_n);
      f.rmean[c] = rm_n;
      f.rvar[c] = rv_n;
    }
    f.scale[c] = sc;
    f.shift[c] = sf;
  } else {
    f.dbeta[c] = (float)ts;
    f.dgamma[c] = (float)tq;
  }
}

// one 1024-thread block: channel group bx, row group g.  lds: sh_s / sh_q [32][33] doubles + one int.
__device__ __forceinline__ void fin2_body(const Fin2P& f, int bx, int g, double (*sh_s)[33], double (*sh_q)[33],
                                          int* sh_t) {
  const int tid = threadIdx.x;
  const int cl = tid & 31, sl = tid >> 5;
  const int c = bx * 32 + cl;
  const int C = f.C;
  const int cc = c < C ? c : C - 1;
  // the closing block's per-channel parameters, requested up front (their latency runs beside the rows')
  float ga_c = 0.f, be_c = 0.f, rm_c = 0.f, rv_c = 0.f;
  if (f.kind == 1) {
    ga_c = f.gamma[cc];
    be_c = f.beta[cc];
    rm_c = (f.rmean ? f.rmean : f.gamma)[cc];
    rv_c = (f.rvar ? f.rvar : f.gamma)[cc];
  }
  const int r0 = g * f.rpg, r1 = min(f.nparts, r0 + f.rpg);
  double s = 0.0, q = 0.0;
  if (c < C) {
    const float* base = f.part + c;
    for (int p = r0 + sl; p < r1; p += 256) {  // 8 rows (16 loads) in flight; rows past r1 re-read row r1 - 1, dropped
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int pr = min(p + 32 * u, r1 - 1);
        a[u] = base[(long long)pr * 2 * C];
        b[u] = base[(long long)pr * 2 * C + C];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (p + 32 * u < r1) {
          s += (double)a[u];
          q += (double)b[u];
        }
    }
  }
  sh_s[sl][cl] = s;
  sh_q[sl][cl] = q;
  __syncthreads();
  double ts = 0.0, tq = 0.0;
  if (sl == 0 && c < C) {
    for (int i = 0; i < 32; ++i) {
      ts += sh_s[i][cl];
      tq += sh_q[i][cl];
    }
    if (f.G == 1) {
      fin2_close(f, c, ts, tq, ga_c, be_c, rm_c, rv_c);
    } else {
      st_part64(f.lvl1 + ((long long)g * 2) * C + c, ts);
      st_part64(f.lvl1 + ((long long)g * 2 + 1) * C + c, tq);
    }
  }
  if (f.G == 1) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the group sums are written through before the ticket
  __syncthreads();
  if (tid == 0) {
    const int t = atomicAdd(f.cnt + bx, 1);
    if (t == f.G - 1) __hip_atomic_store(f.cnt + bx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *sh_t = t;
  }
  __syncthreads();
  if (*sh_t != f.G - 1) return;
  double ls = 0.0, lq = 0.0;
  if (sl < f.G && c < C) {
    ls = ld_part64(f.lvl1 + ((long long)sl * 2) * C + c);
    lq = ld_part64(f.lvl1 + ((long long)sl * 2 + 1) * C + c);
  }
  __syncthreads();  // (sh_s / sh_q were read above by slice 0)
  sh_s[sl][cl] = ls;
  sh_q[sl][cl] = lq;
  __syncthreads();
  if (sl == 0 && c < C) {
    ts = 0.0;
    tq = 0.0;
    for (int i = 0; i < f.G; ++i) {
      ts += sh_s[i][cl];
      tq += sh_q[i][cl];
    }
    fin2_close(f, c, ts, tq, ga_c, be_c, rm_c, rv_c);
  }
}

__global__ __launch_bounds__(1024) void bn_finalize2_kernel(Fin2P f, int g1) {
  __shared__ double sh_s[32][33];
  __shared__ double sh_q[32][33];
  __shared__ int sh_t;
  fin2_body(f, blockIdx.x % g1, blockIdx.x / g1, sh_s, sh_q, &sh_t);
}

// + the pending slab reduce of the previous unit's weight gradient (see bn_bwd_finalize_wgrad_reduce_kernel below)
__global__ __launch_bounds__(1024) void bn_finalize2_wgrad_reduce_kernel(Fin2P f, int g1, int nfin, const float* slabs,
                                                                        float* dw, long long n, int S) {
  // the larger of: sh_s + sh_q (2 x 8 448 B) + the ticket word | four part[16][17] float4 tiles of the slice-form reduce
  __shared__ __attribute__((aligned(16))) char lds[4 * 16 * 17 * 16 + 16];
  static_assert(2 * 32 * 33 * 8 <= 4 * 16 * 17 * 16, "finalize rows fit under the reduce's tiles");
  if ((int)blockIdx.x < nfin) {
    double (*sh_s)[33] = (double (*)[33])lds;
    double (*sh_q)[33] = (double (*)[33])(lds + 32 * 33 * 8);
    fin2_body(f, blockIdx.x % g1, blockIdx.x / g1, sh_s, sh_q, (int*)(lds + 4 * 16 * 17 * 16));
  } else if (wgrad_reduce_cols(n, S)) {
    if (threadIdx.x < 256) wgrad_reduce_body(slabs, dw, n, S, nullptr, (long long)(blockIdx.x - nfin), threadIdx.x);
  } else {
    float4 (*part)[17] = (float4 (*)[17])(lds + (threadIdx.x >> 8) * 16 * 17 * 16);
    wgrad_reduce_body(slabs, dw, n, S, part, (long long)(blockIdx.x - nfin) * 4 + (threadIdx.x >> 8), threadIdx.x & 255);
  }
}

// groups of the plan: one up to 256 rows (every slice-thread's rows in ONE batch of 8), else groups of 256 rows, at most 32
static bool fin2_plan(Fin2P& f, int nparts, int C, void* ws, size_t ws_bytes) {
  static const int on = [] { const char* e = getenv("VS_BN_FIN2"); return e ? atoi(e) : 1; }();
  if (!on || ws == nullptr || ws_bytes < vs_bn_finalize_workspace_bytes() || C > 4096) return false;
  int rpg = 256;
  while ((nparts + rpg - 1) / rpg > 32) rpg += 256;
  f.nparts = nparts;
  f.C = C;
  f.rpg = rpg;
  f.G = (nparts + rpg - 1) / rpg;
  if (f.G < 1) f.G = 1;
  f.cnt = (int*)ws;
  f.lvl1 = (double*)((char*)ws + VS_FIN2_CNT_BYTES);
  return true;
}

extern "C" int vs_bn_partials_reduce(const float* partials, int nparts, float* out, int C, int G,
                                     void* stream) {
  VS_CHECK_ARG(partials && out && nparts > 0 && C > 0 && G > 0, "bad args");
  hipLaunchKernelGGL(bn_partials_reduce_kernel, dim3((C + 31) / 32, G), dim3(1024), 0,
                     (hipStream_t)stream, partials, nparts, out, C, G);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

extern "C" int vs_bn_finalize(const float* partials, int nparts, double count, const float* gamma,
                              const float* beta, float* running_mean, float* running_var,
                              float momentum, float eps, float* scale, float* shift, float* mean,
                              float* invstd, int C, void* stream) {
  VS_CHECK_ARG(gamma && beta && scale && shift && C > 0, "bad args");
  VS_CHECK_ARG(nparts == 0 || (partials && count > 0), "train mode needs partials and count");
  VS_CHECK_ARG(nparts > 0 || (running_mean && running_var), "eval mode needs running stats");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream,
                     partials, nparts, count, gamma, beta, running_mean, running_var, momentum, eps,
                     scale, shift, mean, invstd, C);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// vs_bn_finalize (train mode) with a workspace: any number of partial rows in ONE launch (two levels inside it).
extern "C" int vs_bn_finalize_ws(const float* partials, int nparts, double count, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float eps, float* scale,
                                 float* shift, float* mean, float* invstd, int C, void* workspace, size_t ws_bytes,
                                 void* stream) {
  VS_CHECK_ARG(gamma && beta && scale && shift && C > 0, "bad args");
  VS_CHECK_ARG(partials && nparts > 0 && count > 0, "train mode needs partials and count");
  Fin2P f;
  if (!fin2_plan(f, nparts, C, workspace, ws_bytes)) {
    // the two-launch form (VS_BN_FIN2=0, or no workspace)
    if (nparts > 512 && workspace && ws_bytes >= (size_t)32 * 2 * C * sizeof(float) + VS_FIN2_CNT_BYTES) {
      float* lvl1 = (float*)((char*)workspace + VS_FIN2_CNT_BYTES);
      int rc = vs_bn_partials_reduce(partials, nparts, lvl1, C, 32, stream);
      if (rc) return rc;
      return vs_bn_finalize(lvl1, 32, count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean,
                            invstd, C, stream);
    }
    return vs_bn_finalize(partials, nparts, count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift,
                          mean, invstd, C, stream);
  }
  f.part = partials;
  f.kind = 1;
  f.count = count;
  f.momentum = momentum;
  f.eps = eps;
  f.gamma = gamma;
  f.beta = beta;
  f.rmean = running_mean;
  f.rvar = running_var;
  f.scale = scale;
  f.shift = shift;
  f.mean = mean;
  f.invstd = invstd;
  f.dgamma = f.dbeta = nullptr;
  const int g1 = (C + 31) / 32;
  hipLaunchKernelGGL(bn_finalize2_kernel, dim3(g1 * f.G), dim3(1024), 0, (hipStream_t)stream, f, g1);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ----------------------------------------------------------------------------
// out = relu?(y*scale + shift (+res)), rows x C bf16 with row pitches
// ----------------------------------------------------------------------------
__global__ void bn_apply_kernel(const uint16_t* y, const float* scale, const float* shift,
                                const uint16_t* res, uint16_t* out, long long rows, int C, int y_ld,
                                int res_ld, int out_ld, int relu) {
  const int cpr = C >> 3;
  const long long total = rows * cpr;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long row = idx / cpr;
    const int c = (int)(idx - row * cpr) * 8;
    float v[8], r[8];
    unpack8_bf16(*(const uint4*)(y + row * y_ld + c), v);
    const float4 s0 = *(const float4*)(scale + c), s1 = *(const float4*)(scale + c + 4);
    const float4 h0 = *(const float4*)(shift + c), h1 = *(const float4*)(shift + c + 4);
    v[0] = v[0] * s0.x + h0.x; v[1] = v[1] * s0.y + h0.y;
    v[2] = v[2] * s0.z + h0.z; v[3] = v[3] * s0.w + h0.w;
    v[4] = v[4] * s1.x + h1.x; v[5] = v[5] * s1.y + h1.y;
    v[6] = v[6] * s1.z + h1.z; v[7] = v[7] * s1.w + h1.w;
    if (res) {
      unpack8_bf16(*(const uint4*)(res + row * res_ld + c), r);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += r[e];
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    *(uint4*)(out + row * out_ld + c) = pack8_bf16(v);
  }
}

static inline int ew_grid(long long total) {
  long long g = (total + 255) / 256;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

// 16-byte load of a tensor an APPLY pass reads for the last time in a while (the raw convolution output y, the incoming
// gradient dz): a non-temporal load, so that the stream does not push the pass's OUTPUT -- which the next kernel reads --
// out of L2 / Infinity Cache.  (-DVS_BN_NT=0 restores plain loads: A/B.)  The REDUCE passes keep plain loads: the apply
// pass behind them reads the same tensors again.
#ifndef VS_BN_NT
#define VS_BN_NT 1
#endif
__device__ __forceinline__ uint4 ld_stream16(const uint16_t* p) {
#if VS_BN_NT
  const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
  return make_uint4(v[0], v[1], v[2], v[3]);
#else
  return *(const uint4*)p;
#endif
}

// Column-owner form (C/8 a power of two): a thread keeps the scale / shift of ITS 8 channels in
// registers and walks rows in batches of 4 with every load of a batch in flight before the first
// use; no per-element 64-bit division, no per-element parameter reloads.
#define BNA_BATCH 4

static int bn_rows_batches(long long rows, int C, int target_blocks) {
  // VS_BN_TARGET / VS_BN_NBMAX: sweep knobs (tools/bn_time.py); the defaults are what the sweep kept
  static const int env_target = getenv("VS_BN_TARGET") ? atoi(getenv("VS_BN_TARGET")) : 0;
  static const int env_nbmax = getenv("VS_BN_NBMAX") ? atoi(getenv("VS_BN_NBMAX")) : 0;
  if (env_target > 0) target_blocks = env_target;
  const int nbmax = env_nbmax > 0 ? env_nbmax : 4;
  const int cpr = C / 8;
  const int ncol = cpr < 256 ? cpr : 256;
  const long long rl = 256 / ncol;
  long long nb = (rows + rl * BNA_BATCH * target_blocks - 1) / (rl * BNA_BATCH * target_blocks);
  if (nb < 1) nb = 1;
  if (nb > nbmax) nb = nbmax;
  return (int)nb;
}

static bool cpr_pow2(int C) {
  const int cpr = C / 8;
  return C % 8 == 0 && cpr > 0 && (cpr & (cpr - 1)) == 0;
}

// RES_AFF (with RES): `res` is the RAW convolution output of a second unit (a ResBlock's shortcut) and the residual is
// bf16(res * scale2 + shift2) -- the tensor that unit's own apply pass would have stored, formed here instead (same
// fma, same rounding: bitwise the two-pass result; the shortcut's normalised output is neither written nor re-read).
template <bool RES, bool RELU, bool RES_AFF = false>
__global__ __launch_bounds__(256) void bn_apply_cols_kernel(const uint16_t* y, const float* scale,
                                                            const float* shift, const uint16_t* res,
                                                            uint16_t* out, uint8_t* bits,
                                                            long long rows, int C, int y_ld,
                                                            int res_ld, int out_ld, int nbatch,
                                                            const float* scale2 = nullptr,
                                                            const float* shift2 = nullptr) {
  const int cpr = C >> 3;
  const int ncol = cpr < 256 ? cpr : 256;
  const int rl = 256 / ncol;
  const int col = threadIdx.x % ncol, lane_r = threadIdx.x / ncol;
  const long long r0 = (long long)blockIdx.x * rl * BNA_BATCH * nbatch;
  for (int cb = col; cb < cpr; cb += ncol) {
    const int c = cb * 8;
    float sc[8], sh[8];
    *(float4*)(sc) = *(const float4*)(scale + c);
    *(float4*)(sc + 4) = *(const float4*)(scale + c + 4);
    *(float4*)(sh) = *(const float4*)(shift + c);
    *(float4*)(sh + 4) = *(const float4*)(shift + c + 4);
    float sc2[8], sh2[8];
    if (RES_AFF) {
      *(float4*)(sc2) = *(const float4*)(scale2 + c);
      *(float4*)(sc2 + 4) = *(const float4*)(scale2 + c + 4);
      *(float4*)(sh2) = *(const float4*)(shift2 + c);
      *(float4*)(sh2 + 4) = *(const float4*)(shift2 + c + 4);
    }
    for (int b = 0; b < nbatch; ++b) {
      uint4 vy[BNA_BATCH], vr[BNA_BATCH];
      long long row[BNA_BATCH];
#pragma unroll
      for (int u = 0; u < BNA_BATCH; ++u) {
        row[u] = r0 + (long long)(b * BNA_BATCH + u) * rl + lane_r;
        const long long rr = row[u] < rows ? row[u] : 0;
        vy[u] = ld_stream16(y + rr * y_ld + c);
        if (RES) vr[u] = ld_stream16(res + rr * res_ld + c);  // (the block's input: its last reader of the forward pass)
      }
#pragma unroll
      for (int u = 0; u < BNA_BATCH; ++u) {
        float v[8], r[8];
        unpack8_bf16(vy[u], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = __fmaf_rn(v[e], sc[e], sh[e]);
        if (RES) {
          unpack8_bf16(vr[u], r);
          if (RES_AFF) {
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = bf16_to_f32(f32_to_bf16(__fmaf_rn(r[e], sc2[e], sh2[e])));
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        unsigned mbits = 0u;
        if (RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            mbits |= (v[e] > 0.f ? 1u : 0u) << e;
            v[e] = fmaxf(v[e], 0.f);
          }
        }
        if (row[u] < rows) {
          *(uint4*)(out + row[u] * out_ld + c) = pack8_bf16(v);
          // ReLU mask as one bit per element: the backward passes read it instead of the 16x
          // larger output tensor
          if (RELU && bits) __builtin_nontemporal_store((uint8_t)mbits, bits + row[u] * cpr + cb);  // (read in the backward pass)
        }
      }
    }
  }
}

static int bn_apply_impl(const void* y, const float* scale, const float* shift, const void* residual,
                         void* out, uint8_t* relu_bits, int64_t rows, int C, int y_ld, int res_ld,
                         int out_ld, int relu, void* stream) {
  VS_CHECK_ARG(y && scale && shift && out, "null tensor");
  VS_CHECK_ARG(C % 8 == 0 && y_ld % 8 == 0 && out_ld % 8 == 0 && (!residual || res_ld % 8 == 0),
               "channels / pitches must be multiples of 8");
  hipStream_t st = (hipStream_t)stream;
  if (cpr_pow2(C)) {
    const int nb = bn_rows_batches(rows, C, 2048);
    const int cpr = C / 8, ncol = cpr < 256 ? cpr : 256;
    const long long rpb = (long long)(256 / ncol) * BNA_BATCH * nb;
    const dim3 grid((unsigned)((rows + rpb - 1) / rpb)), block(256);
#define BNA_ARGS (const uint16_t*)y, scale, shift, (const uint16_t*)residual, (uint16_t*)out, relu_bits, (long long)rows, C, y_ld, res_ld, out_ld, nb
    if (residual && relu) hipLaunchKernelGGL((bn_apply_cols_kernel<true, true>), grid, block, 0, st, BNA_ARGS);
    else if (residual) hipLaunchKernelGGL((bn_apply_cols_kernel<true, false>), grid, block, 0, st, BNA_ARGS);
    else if (relu) hipLaunchKernelGGL((bn_apply_cols_kernel<false, true>), grid, block, 0, st, BNA_ARGS);
    else hipLaunchKernelGGL((bn_apply_cols_kernel<false, false>), grid, block, 0, st, BNA_ARGS);
#undef BNA_ARGS
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  VS_CHECK_ARG(!relu_bits, "the ReLU bit mask needs C/8 to be a power of two");
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(rows * (C / 8))), dim3(256), 0, st,
                     (const uint16_t*)y, scale, shift, (const uint16_t*)residual, (uint16_t*)out,
                     (long long)rows, C, y_ld, res_ld, out_ld, relu);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

extern "C" int vs_bn_apply(const void* y, const float* scale, const float* shift,
                           const void* residual, void* out, int64_t rows, int C, int y_ld,
                           int res_ld, int out_ld, int relu, void* stream) {
  return bn_apply_impl(y, scale, shift, residual, out, nullptr, rows, C, y_ld, res_ld, out_ld, relu,
                       stream);
}

extern "C" int vs_bn_apply_mask(const void* y, const float* scale, const float* shift,
                                const void* residual, void* out, uint8_t* relu_bits, int64_t rows,
                                int C, int y_ld, int res_ld, int out_ld, void* stream) {
  VS_CHECK_ARG(relu_bits, "null mask");
  return bn_apply_impl(y, scale, shift, residual, out, relu_bits, rows, C, y_ld, res_ld, out_ld, 1,
                       stream);
}

// out = relu(y * scale + shift + bf16(y2 * scale2 + shift2)) (+ the ReLU bit mask): the c unit's apply pass of a
// ResBlock whose shortcut unit hands over its RAW convolution output y2 and BN constants instead of a normalised tensor
// (slowfast ResBlock.forward: `x = self.branch1_bn(self.branch1(x)) + self.branch2(x); x = self.relu(x)`).
extern "C" int vs_bn_apply2(const void* y, const float* scale, const float* shift, const void* y2, const float* scale2,
                            const float* shift2, void* out, uint8_t* relu_bits, int64_t rows, int C, int y_ld, int y2_ld,
                            int out_ld, void* stream) {
  VS_CHECK_ARG(y && scale && shift && y2 && scale2 && shift2 && out, "null tensor");
  VS_CHECK_ARG(C % 8 == 0 && y_ld % 8 == 0 && out_ld % 8 == 0 && y2_ld % 8 == 0, "channels / pitches must be multiples of 8");
  VS_CHECK_ARG(cpr_pow2(C), "C/8 must be a power of two");
  const int nb = bn_rows_batches(rows, C, 2048);
  const int cpr = C / 8, ncol = cpr < 256 ? cpr : 256;
  const long long rpb = (long long)(256 / ncol) * BNA_BATCH * nb;
  hipLaunchKernelGGL((bn_apply_cols_kernel<true, true, true>), dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0,
                     (hipStream_t)stream, (const uint16_t*)y, scale, shift, (const uint16_t*)y2, (uint16_t*)out,
                     relu_bits, (long long)rows, C, y_ld, y2_ld, out_ld, nb, scale2, shift2);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ----------------------------------------------------------------------------
// The stems: BN + ReLU + max-pool [1,3,3] / [1,2,2] / pad [0,1,1] as ONE forward pass over the conv output (the
// normalised full-resolution tensor is never written: its only reader was the pool) and, in the backward passes,
// the pool's gradient gathered on the fly from the pooled gradient and the argmax bytes instead of read from a
// dense tensor a pool-backward launch wrote.  The arithmetic is the separate kernels' (values rounded to bf16
// where they were stored as bf16, same comparison and summation order): bitwise the unfused path.
// ----------------------------------------------------------------------------
struct PoolSrc {
  const uint16_t* dp;  // gradient of the pooled tensor [NT, Ho, Wo, C]
  const uint8_t* idx;  // argmax tap (dh * 3 + dw) per pooled element
  int H, W, Ho, Wo, dp_ld;
  float rcp_w, rcp_h;
};

// dz of one full-resolution row (< 2^24 rows) and 8 channels, as the bf16 vector vs_maxpool_hw3s2_bwd would have stored
__device__ __forceinline__ uint4 pool_grad8(const PoolSrc& ps, int row, int c, int C) {
  int p1, w, nt, h;
  fast_divmod(row, ps.W, ps.rcp_w, p1, w);
  fast_divmod(p1, ps.H, ps.rcp_h, nt, h);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const int ho_lo = h >> 1, ho_hi = (h + 1) >> 1;
  const int wo_lo = w >> 1, wo_hi = (w + 1) >> 1;
  for (int ho = ho_lo; ho <= ho_hi; ++ho) {
    if (ho >= ps.Ho) continue;
    const int dh = h - (2 * ho - 1);
    for (int wo = wo_lo; wo <= wo_hi; ++wo) {
      if (wo >= ps.Wo) continue;
      const int tap = dh * 3 + (w - (2 * wo - 1));
      const long long opos = ((long long)nt * ps.Ho + ho) * ps.Wo + wo;
      const uint2 pk = *(const uint2*)(ps.idx + opos * C + c);
      float g[8];
      unpack8_bf16(*(const uint4*)(ps.dp + opos * ps.dp_ld + c), g);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned b = ((e < 4 ? pk.x : pk.y) >> ((e & 3) * 8)) & 0xff;
        if ((int)b == tap) acc[e] += g[e];
      }
    }
  }
  return pack8_bf16(acc);
}

__global__ void bn_apply_maxpool_kernel(const uint16_t* y, const float* scale, const float* shift, uint16_t* out,
                                        uint8_t* idx, int NT, int H, int W, int Ho, int Wo, int C, int y_ld,
                                        int out_ld) {
  const int cpr = C >> 3;
  const int total = NT * Ho * Wo * cpr;  // < 2^24 (checked by the launcher)
  const float rcp_c = 1.0f / (float)cpr, rcp_w = 1.0f / (float)Wo, rcp_h = 1.0f / (float)Ho;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int p1, p2, nt, cb, wo, ho;
    fast_divmod(i, cpr, rcp_c, p1, cb);
    fast_divmod(p1, Wo, rcp_w, p2, wo);
    fast_divmod(p2, Ho, rcp_h, nt, ho);
    const int c = cb * 8;
    float sc[8], sh[8], best[8];
    int bi[8];
    *(float4*)(sc) = *(const float4*)(scale + c);
    *(float4*)(sc + 4) = *(const float4*)(scale + c + 4);
    *(float4*)(sh) = *(const float4*)(shift + c);
    *(float4*)(sh + 4) = *(const float4*)(shift + c + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      best[e] = -INFINITY;
      bi[e] = 0;
    }
    for (int dh = 0; dh < 3; ++dh) {
      const int h = 2 * ho - 1 + dh;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int dw = 0; dw < 3; ++dw) {
        const int w = 2 * wo - 1 + dw;
        if ((unsigned)w >= (unsigned)W) continue;
        float v[8];
        unpack8_bf16(*(const uint4*)(y + (((long long)nt * H + h) * W + w) * y_ld + c), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], 0.f);
        unpack8_bf16(pack8_bf16(v), v);  // what bn_apply stores and the pool reads
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (v[e] > best[e] || (v[e] != v[e] && best[e] == best[e])) {
            best[e] = v[e];
            bi[e] = dh * 3 + dw;
          }
      }
    }
    const long long opos = ((long long)nt * Ho + ho) * Wo + wo;
    *(uint4*)(out + opos * out_ld + c) = pack8_bf16(best);
    if (idx) {
      uint2 pk;
      pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
      pk.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
      *(uint2*)(idx + opos * C + c) = pk;
    }
  }
}

extern "C" int vs_bn_apply_maxpool(const void* y, const float* scale, const float* shift, void* out, uint8_t* idx,
                                   int N, int T, int H, int W, int C, int y_ld, int out_ld, void* stream) {
  VS_CHECK_ARG(y && scale && shift && out, "null tensor");
  VS_CHECK_ARG(C % 8 == 0 && y_ld % 8 == 0 && out_ld % 8 == 0, "C / pitches multiple of 8");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long total = (long long)N * T * Ho * Wo * (C / 8);
  if (total >= (1ll << 24) || (long long)N * T * H * W >= (1ll << 24)) {
    vs_set_error("vs_bn_apply_maxpool: fewer than 2^24 rows (use vs_bn_apply + vs_maxpool_hw3s2_fwd)");
    return VS_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(bn_apply_maxpool_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)y, scale, shift, (uint16_t*)out, idx, N * T, H, W, Ho, Wo, C, y_ld, out_ld);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ----------------------------------------------------------------------------
// BN backward.  g = dz * [z > 0] (relu) ; xhat = (y - mean) * invstd
//   reduce : partial[blk][2][C] = (sum g, sum g*xhat) over the block's row slab
//   apply  : dy = gamma*invstd*(g - dbeta/M - xhat*dgamma/M) ; dres = g
// block = 256 threads = cpr chunk-columns x (256/cpr) row lanes, cpr = C/8 | 256
// ----------------------------------------------------------------------------
// Rows per thread come in batches of 4 whose loads are all issued before the first use (a
// one-row-at-a-time loop is a chain of HBM latencies: 16 us floor however small the tensor).
// The batch count adapts to the tensor so that small layers still fill the chip.
#define BNB_BATCH 4

static int bnb_batches(long long rows, int C) {
  const int cpr = C / 8;
  const int ncol = cpr < 256 ? cpr : 256;
  const long long rl = 256 / ncol;
  long long nb = (rows + rl * BNB_BATCH * 1024 - 1) / (rl * BNB_BATCH * 1024);  // aim at ~1024 blocks
  if (nb < 1) nb = 1;
  if (nb > 4) nb = 4;
  return (int)nb;
}

// MASK 0: no relu; 1: relu mask from z; 2: mask recomputed as gamma*xhat + beta > 0 (units
// without a residual input), which drops one of the three reads of each pass; 3: `z` is the
// bit mask written by vs_bn_apply_mask ([rows][C/8] bytes): one byte instead of 16 per load.
template <int MASK, bool POOL = false>  // POOL: dz gathered from (pooled gradient, argmax bytes) -- see PoolSrc
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const uint16_t* dz, const uint16_t* z, const uint16_t* y, const float* mean,
    const float* invstd, const float* gamma, const float* beta, float* partial, long long rows,
    int C, int dz_ld, int z_ld, int y_ld, int nbatch, PoolSrc ps = PoolSrc()) {
  __shared__ float red[256 * 16];
  const int cpr = C >> 3;
  const int ncol = cpr < 256 ? cpr : 256;  // chunk columns handled per pass
  const int rl = 256 / ncol;               // row lanes
  const int col = threadIdx.x % ncol, lane_r = threadIdx.x / ncol;
  const long long rows_per_blk = (long long)rl * BNB_BATCH * nbatch;
  const long long r0 = (long long)blockIdx.x * rows_per_blk;
  for (int cb = col; cb < cpr; cb += ncol) {
    const int c = cb * 8;
    float mu[8], is[8], sg[8], sx[8], ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mu[e] = mean[c + e];
      is[e] = invstd[c + e];
      ga[e] = (MASK == 2) ? gamma[c + e] : 0.f;
      be[e] = (MASK == 2) ? beta[c + e] : 0.f;
      sg[e] = 0.f;
      sx[e] = 0.f;
    }
    for (int b = 0; b < nbatch; ++b) {
      uint4 vg[BNB_BATCH], vy[BNB_BATCH], vz[BNB_BATCH];
#pragma unroll
      for (int u = 0; u < BNB_BATCH; ++u) {  // branch-free: a row past the end re-reads row 0, g := 0
        const long long row = r0 + (long long)(b * BNB_BATCH + u) * rl + lane_r;
        const bool ok = row < rows;
        const long long rr = ok ? row : 0;
        vg[u] = POOL ? pool_grad8(ps, (int)rr, c, C) : *(const uint4*)(dz + rr * dz_ld + c);
        vy[u] = *(const uint4*)(y + rr * y_ld + c);
        if (MASK == 1) vz[u] = *(const uint4*)(z + rr * z_ld + c);
        if (MASK == 3) vz[u].x = ((const uint8_t*)z)[rr * cpr + cb];
        if (!ok) vg[u] = make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < BNB_BATCH; ++u) {
        float g[8], yv[8];
        unpack8_bf16(vg[u], g);
        unpack8_bf16(vy[u], yv);
        if (MASK == 1) {
          float zv[8];
          unpack8_bf16(vz[u], zv);
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = zv[e] > 0.f ? g[e] : 0.f;
        } else if (MASK == 2) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            g[e] = ((yv[e] - mu[e]) * is[e] * ga[e] + be[e]) > 0.f ? g[e] : 0.f;
        } else if (MASK == 3) {
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = ((vz[u].x >> e) & 1u) ? g[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          sg[e] += g[e];
          sx[e] += g[e] * (yv[e] - mu[e]) * is[e];
        }
      }
    }
    // fixed-order tree over the rl row lanes through LDS
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[threadIdx.x * 16 + e] = sg[e];
      red[threadIdx.x * 16 + 8 + e] = sx[e];
    }
    __syncthreads();
    for (int st = rl >> 1; st >= 1; st >>= 1) {
      if (lane_r < st) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          red[threadIdx.x * 16 + e] += red[(threadIdx.x + st * ncol) * 16 + e];
      }
      __syncthreads();
    }
    if (lane_r == 0) {
      float* dst = partial + (long long)blockIdx.x * 2 * C;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        dst[c + e] = red[threadIdx.x * 16 + e];
        dst[C + c + e] = red[threadIdx.x * 16 + 8 + e];
      }
    }
  }
}

static int bnb_check(int C) {
  const int cpr = C / 8;
  if (C % 8 != 0 || (cpr & (cpr - 1)) != 0) return 0;
  return 1;
}

extern "C" int vs_bn_bwd_reduce_rows(int64_t rows, int C) {
  if (!bnb_check(C)) return -1;
  const int cpr = C / 8;
  const int ncol = cpr < 256 ? cpr : 256;
  const long long rpb = (long long)(256 / ncol) * BNB_BATCH * bnb_batches(rows, C);
  return (int)((rows + rpb - 1) / rpb);
}

extern "C" int vs_bn_bwd_reduce(const void* dz, const void* z, const void* y, const float* mean,
                                const float* invstd, const float* gamma, const float* beta,
                                float* partial, int64_t rows, int C, int dz_ld, int z_ld, int y_ld,
                                int relu, void* stream) {
  VS_CHECK_ARG(dz && y && mean && invstd && partial, "null tensor");
  VS_CHECK_ARG(!relu || z || (gamma && beta), "relu needs z, or gamma/beta to recompute the mask");
  VS_CHECK_ARG(relu != 2 || z, "relu = 2 needs the bit mask in z");
  VS_CHECK_ARG(bnb_check(C), "C/8 must be a power of two");
  const int nblk = vs_bn_bwd_reduce_rows(rows, C);
  const int nb = bnb_batches(rows, C);
#define VS_BNB_LAUNCH(MASK)                                                                        \
  hipLaunchKernelGGL(bn_bwd_reduce_kernel<MASK>, dim3(nblk), dim3(256), 0, (hipStream_t)stream,    \
                     (const uint16_t*)dz, (const uint16_t*)z, (const uint16_t*)y, mean, invstd,    \
                     gamma, beta, partial, (long long)rows, C, dz_ld, z_ld, y_ld, nb)
  if (!relu) VS_BNB_LAUNCH(0);
  else if (relu == 2) VS_BNB_LAUNCH(3);
  else if (z) VS_BNB_LAUNCH(1);
  else VS_BNB_LAUNCH(2);
#undef VS_BNB_LAUNCH
  VS_CHECK_LAUNCH();
  return VS_OK;
}

__device__ __forceinline__ void bn_bwd_finalize_body(const float* partial, int nparts, float* dgamma, float* dbeta,
                                                     int C, int bid, double (*sh_s)[33], double (*sh_q)[33]) {
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = bid * 32 + cl;
  double s = 0.0, q = 0.0;
  if (c < C) {
    int p = sl;
    for (; p + 96 < nparts; p += 128) {  // 8 loads in flight, same summation order
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = partial[(long long)(p + 32 * u) * 2 * C + c];
        b[u] = partial[(long long)(p + 32 * u) * 2 * C + C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s += (double)a[u];
        q += (double)b[u];
      }
    }
    for (; p < nparts; p += 32) {
      s += (double)partial[(long long)p * 2 * C + c];
      q += (double)partial[(long long)p * 2 * C + C + c];
    }
  }
  sh_s[sl][cl] = s;
  sh_q[sl][cl] = q;
  __syncthreads();
  if (sl == 0 && c < C) {
    double ts = 0.0, tq = 0.0;
    for (int i = 0; i < 32; ++i) {
      ts += sh_s[i][cl];
      tq += sh_q[i][cl];
    }
    dbeta[c] = (float)ts;
    dgamma[c] = (float)tq;
  }
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* partial, int nparts,
                                                              float* dgamma, float* dbeta, int C) {
  __shared__ double sh_s[32][33];
  __shared__ double sh_q[32][33];
  bn_bwd_finalize_body(partial, nparts, dgamma, dbeta, C, blockIdx.x, sh_s, sh_q);
}

// The finalize and the slab reduce of the previous unit's weight gradient (pending on the stream: conv_wgrad.hip,
// vs_wgrad_reduce_defer) as ONE grid: blocks [0, g1) finalize, the rest hold four 256-thread virtual blocks of the
// reduce each.  Same bodies as the two stand-alone kernels: bitwise their results; one launch fewer on the chain.
__global__ __launch_bounds__(1024) void bn_bwd_finalize_wgrad_reduce_kernel(const float* partial, int nparts,
                                                                           float* dgamma, float* dbeta, int C, int g1,
                                                                           const float* slabs, float* dw, long long n,
                                                                           int S) {
  __shared__ __attribute__((aligned(16))) char lds[4 * 16 * 17 * 16];  // 4 x part[16][17] >= sh_s + sh_q (2 x 8 448 B)
  if ((int)blockIdx.x < g1) {
    double (*sh_s)[33] = (double (*)[33])lds;
    double (*sh_q)[33] = (double (*)[33])(lds + 32 * 33 * 8);
    bn_bwd_finalize_body(partial, nparts, dgamma, dbeta, C, blockIdx.x, sh_s, sh_q);
  } else if (wgrad_reduce_cols(n, S)) {
    // column form: ONE 256-column virtual block per block (the other 12 waves leave at once) -- four of them packed
    // into a 1024-thread block put a 590 K-element, 14-slab reduce on 144 CUs, and a CU takes in ~30 GB/s from beyond L2
    if (threadIdx.x < 256) wgrad_reduce_body(slabs, dw, n, S, nullptr, (long long)(blockIdx.x - g1), threadIdx.x);
  } else {
    float4 (*part)[17] = (float4 (*)[17])(lds + (threadIdx.x >> 8) * 16 * 17 * 16);
    wgrad_reduce_body(slabs, dw, n, S, part, (long long)(blockIdx.x - g1) * 4 + (threadIdx.x >> 8), threadIdx.x & 255);
  }
}

extern "C" int vs_bn_bwd_finalize(const float* partial, int nparts, float* dgamma, float* dbeta,
                                  int C, void* stream) {
  VS_CHECK_ARG(partial && dgamma && dbeta && nparts > 0, "bad args");
  VsPendingReduce pr;
  const int g1 = (C + 31) / 32;
  if (vs_pending_reduce_take((hipStream_t)stream, &pr)) {
    const long long rb = wgrad_reduce_vblocks(pr.n, pr.S);  // 256-thread virtual blocks of the reduce
    const long long rblocks = wgrad_reduce_cols(pr.n, pr.S) ? rb : (rb + 3) / 4;
    hipLaunchKernelGGL(bn_bwd_finalize_wgrad_reduce_kernel, dim3((unsigned)(g1 + rblocks)), dim3(1024), 0,
                       (hipStream_t)stream, partial, nparts, dgamma, dbeta, C, g1, pr.slabs, pr.dw, pr.n, pr.S);
  } else {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(g1), dim3(1024), 0, (hipStream_t)stream, partial, nparts, dgamma,
                       dbeta, C);
  }
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// vs_bn_bwd_finalize with a workspace: the rows summed by (C / 32) x G blocks in one launch (bn_finalize2_kernel), the
// pending slab reduce riding in the same grid as before.
extern "C" int vs_bn_bwd_finalize_ws(const float* partial, int nparts, float* dgamma, float* dbeta, int C, void* workspace,
                                     size_t ws_bytes, void* stream) {
  VS_CHECK_ARG(partial && dgamma && dbeta && nparts > 0, "bad args");
  Fin2P f;
  if (!fin2_plan(f, nparts, C, workspace, ws_bytes)) return vs_bn_bwd_finalize(partial, nparts, dgamma, dbeta, C, stream);
  f.part = partial;
  f.kind = 2;
  f.count = 1.0;
  f.momentum = f.eps = 0.f;
  f.gamma = f.beta = nullptr;
  f.rmean = f.rvar = f.scale = f.shift = f.mean = f.invstd = nullptr;
  f.dgamma = dgamma;
  f.dbeta = dbeta;
  const int g1 = (C + 31) / 32;
  const int nfin = g1 * f.G;
  VsPendingReduce pr;
  if (vs_pending_reduce_take((hipStream_t)stream, &pr)) {
    const long long rb = wgrad_reduce_vblocks(pr.n, pr.S);
    const long long rblocks = wgrad_reduce_cols(pr.n, pr.S) ? rb : (rb + 3) / 4;
    hipLaunchKernelGGL(bn_finalize2_wgrad_reduce_kernel, dim3((unsigned)(nfin + rblocks)), dim3(1024), 0,
                       (hipStream_t)stream, f, g1, nfin, pr.slabs, pr.dw, pr.n, pr.S);
  } else {
    hipLaunchKernelGGL(bn_finalize2_kernel, dim3(nfin), dim3(1024), 0, (hipStream_t)stream, f, g1);
  }
  VS_CHECK_LAUNCH();
  return VS_OK;
}

template <int MASK>  // 0: no relu, 1: mask from z, 2: mask recomputed from y
__global__ void bn_bwd_apply_kernel(const uint16_t* dz, const uint16_t* z, const uint16_t* y,
                                    const float* mean, const float* invstd, const float* gamma,
                                    const float* beta, const float* dgamma, const float* dbeta,
                                    uint16_t* dy, uint16_t* dres, long long rows, int C, int dz_ld,
                                    int z_ld, int y_ld, int dy_ld, int dres_ld) {
  const int cpr = C >> 3;
  const long long total = rows * cpr;
  const float invM = 1.0f / (float)rows;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long row = idx / cpr;
    const int c = (int)(idx - row * cpr) * 8;
    float g[8], yv[8], o[8], mu[8], is[8], ga[8], be[8], dg[8], db[8];
    unpack8_bf16(*(const uint4*)(dz + row * dz_ld + c), g);
    unpack8_bf16(*(const uint4*)(y + row * y_ld + c), yv);
    *(float4*)(mu) = *(const float4*)(mean + c);
    *(float4*)(mu + 4) = *(const float4*)(mean + c + 4);
    *(float4*)(is) = *(const float4*)(invstd + c);
    *(float4*)(is + 4) = *(const float4*)(invstd + c + 4);
    *(float4*)(ga) = *(const float4*)(gamma + c);
    *(float4*)(ga + 4) = *(const float4*)(gamma + c + 4);
    *(float4*)(dg) = *(const float4*)(dgamma + c);
    *(float4*)(dg + 4) = *(const float4*)(dgamma + c + 4);
    *(float4*)(db) = *(const float4*)(dbeta + c);
    *(float4*)(db + 4) = *(const float4*)(dbeta + c + 4);
    if (MASK == 1) {
      float zv[8];
      unpack8_bf16(*(const uint4*)(z + row * z_ld + c), zv);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = zv[e] > 0.f ? g[e] : 0.f;
    }
    if (MASK == 2) {
      *(float4*)(be) = *(const float4*)(beta + c);
      *(float4*)(be + 4) = *(const float4*)(beta + c + 4);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (yv[e] - mu[e]) * is[e];
      if (MASK == 2) g[e] = (xh * ga[e] + be[e]) > 0.f ? g[e] : 0.f;
      o[e] = ga[e] * is[e] * (g[e] - db[e] * invM - xh * dg[e] * invM);
    }
    *(uint4*)(dy + row * dy_ld + c) = pack8_bf16(o);
    if (dres) *(uint4*)(dres + row * dres_ld + c) = pack8_bf16(g);
  }
}

// One element of the backward apply, every multiply-add spelled out (shared by the column-owner kernel and the one
// that finalizes for itself: bitwise the same dy).  RECOMPUTE: the ReLU mask is gamma * xhat + beta > 0.
template <bool RECOMPUTE>
__device__ __forceinline__ float bn_bwd_elem(float& g, float yv, float mu, float is, float ga, float be, float b1,
                                             float b2) {
  const float xh = __fmul_rn(yv - mu, is);
  if (RECOMPUTE) g = __fmaf_rn(xh, ga, be) > 0.f ? g : 0.f;
  return __fmul_rn(__fmul_rn(ga, is), __fmaf_rn(-xh, b2, g - b1));
}

// Column-owner form of the backward apply (see bn_apply_cols_kernel): per-channel constants
// a = gamma*invstd, b1 = dbeta/M, b2 = dgamma/M live in registers.
template <int MASK, bool DRES, bool POOL = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_cols_kernel(
    const uint16_t* dz, const uint16_t* z, const uint16_t* y, const float* mean, const float* invstd,
    const float* gamma, const float* beta, const float* dgamma, const float* dbeta, uint16_t* dy,
    uint16_t* dres, long long rows, int C, int dz_ld, int z_ld, int y_ld, int dy_ld, int dres_ld,
    int nbatch, PoolSrc ps = PoolSrc()) {
  const int cpr = C >> 3;
  const int ncol = cpr < 256 ? cpr : 256;
  const int rl = 256 / ncol;
  const int col = threadIdx.x % ncol, lane_r = threadIdx.x / ncol;
  const long long r0 = (long long)blockIdx.x * rl * BNA_BATCH * nbatch;
  const float invM = 1.0f / (float)rows;
  for (int cb = col; cb < cpr; cb += ncol) {
    const int c = cb * 8;
    float mu[8], is[8], ga[8], be[8], b1[8], b2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mu[e] = mean[c + e];
      is[e] = invstd[c + e];
      ga[e] = gamma[c + e];
      be[e] = (MASK == 2) ? beta[c + e] : 0.f;
      b1[e] = dbeta[c + e] * invM;
      b2[e] = dgamma[c + e] * invM;
    }
    for (int b = 0; b < nbatch; ++b) {
      uint4 vg[BNA_BATCH], vy[BNA_BATCH], vz[BNA_BATCH];
      long long row[BNA_BATCH];
#pragma unroll
      for (int u = 0; u < BNA_BATCH; ++u) {
        row[u] = r0 + (long long)(b * BNA_BATCH + u) * rl + lane_r;
        const long long rr = row[u] < rows ? row[u] : 0;
        vg[u] = POOL ? pool_grad8(ps, (int)rr, c, C) : ld_stream16(dz + rr * dz_ld + c);
        vy[u] = ld_stream16(y + rr * y_ld + c);
        if (MASK == 1) vz[u] = ld_stream16(z + rr * z_ld + c);
        if (MASK == 3) vz[u].x = __builtin_nontemporal_load((const uint8_t*)z + rr * cpr + cb);
      }
#pragma unroll
      for (int u = 0; u < BNA_BATCH; ++u) {
        float g[8], yv[8], o[8];
        unpack8_bf16(vg[u], g);
        unpack8_bf16(vy[u], yv);
        if (MASK == 1) {
          float zv[8];
          unpack8_bf16(vz[u], zv);
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = zv[e] > 0.f ? g[e] : 0.f;
        }
        if (MASK == 3) {
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = ((vz[u].x >> e) & 1u) ? g[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          // (dbeta/M and dgamma/M are folded per channel: last-bit differences vs the generic kernel)
          o[e] = bn_bwd_elem<MASK == 2>(g[e], yv[e], mu[e], is[e], ga[e], be[e], b1[e], b2[e]);
        }
        if (row[u] < rows) {
          *(uint4*)(dy + row[u] * dy_ld + c) = pack8_bf16(o);
          if (DRES) *(uint4*)(dres + row[u] * dres_ld + c) = pack8_bf16(g);
        }
      }
    }
  }
}

// Two units fed by the SAME masked gradient (a ResBlock's c unit and its shortcut unit: g = dz under the block's ReLU
// bits) in one pass: dz and the bits are read once, both dy written.  Per element the arithmetic of
// bn_bwd_apply_cols_kernel<3> (bn_bwd_elem): bitwise the two passes.
struct BnbUnit {
  const uint16_t* y;
  const float *mean, *invstd, *gamma, *dgamma, *dbeta;
  uint16_t* dy;
  int y_ld, dy_ld;
};
__global__ __launch_bounds__(256) void bn_bwd_apply2_cols_kernel(const uint16_t* dz, const uint8_t* bits, BnbUnit A, BnbUnit B,
                                                                 long long rows, int C, int dz_ld, int nbatch) {
  const int cpr = C >> 3;
  const int ncol = cpr < 256 ? cpr : 256;
  const int rl = 256 / ncol;
  const int col = threadIdx.x % ncol, lane_r = threadIdx.x / ncol;
  const long long r0 = (long long)blockIdx.x * rl * BNA_BATCH * nbatch;
  const float invM = 1.0f / (float)rows;
  for (int cb = col; cb < cpr; cb += ncol) {
    const int c = cb * 8;
    float muA[8], isA[8], gaA[8], b1A[8], b2A[8], muB[8], isB[8], gaB[8], b1B[8], b2B[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      muA[e] = A.mean[c + e]; isA[e] = A.invstd[c + e]; gaA[e] = A.gamma[c + e];
      b1A[e] = A.dbeta[c + e] * invM; b2A[e] = A.dgamma[c + e] * invM;
      muB[e] = B.mean[c + e]; isB[e] = B.invstd[c + e]; gaB[e] = B.gamma[c + e];
      b1B[e] = B.dbeta[c + e] * invM; b2B[e] = B.dgamma[c + e] * invM;
    }
    for (int b = 0; b < nbatch; ++b) {
      uint4 vg[BNA_BATCH], vA[BNA_BATCH], vB[BNA_BATCH];
      unsigned vm[BNA_BATCH];
      long long row[BNA_BATCH];
#pragma unroll
      for (int u = 0; u < BNA_BATCH; ++u) {
        row[u] = r0 + (long long)(b * BNA_BATCH + u) * rl + lane_r;
        const long long rr = row[u] < rows ? row[u] : 0;
        vg[u] = ld_stream16(dz + rr * dz_ld + c);
        vA[u] = ld_stream16(A.y + rr * A.y_ld + c);
        vB[u] = ld_stream16(B.y + rr * B.y_ld + c);
        vm[u] = __builtin_nontemporal_load(bits + rr * cpr + cb);
      }
#pragma unroll
      for (int u = 0; u < BNA_BATCH; ++u) {
        float g[8], ya[8], yb[8], oa[8], ob[8];
        unpack8_bf16(vg[u], g);
        unpack8_bf16(vA[u], ya);
        unpack8_bf16(vB[u], yb);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = ((vm[u] >> e) & 1u) ? g[e] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          oa[e] = bn_bwd_elem<false>(g[e], ya[e], muA[e], isA[e], gaA[e], 0.f, b1A[e], b2A[e]);
          ob[e] = bn_bwd_elem<false>(g[e], yb[e], muB[e], isB[e], gaB[e], 0.f, b1B[e], b2B[e]);
        }
        if (row[u] < rows) {
          *(uint4*)(A.dy + row[u] * A.dy_ld + c) = pack8_bf16(oa);
          *(uint4*)(B.dy + row[u] * B.dy_ld + c) = pack8_bf16(ob);
        }
      }
    }
  }
}

extern "C" int vs_bn_bwd_apply2(const void* dz, const uint8_t* relu_bits, const void* y_a, const float* mean_a,
                                const float* invstd_a, const float* gamma_a, const float* dgamma_a, const float* dbeta_a,
                                void* dy_a, const void* y_b, const float* mean_b, const float* invstd_b,
                                const float* gamma_b, const float* dgamma_b, const float* dbeta_b, void* dy_b, int64_t rows,
                                int C, int dz_ld, int ya_ld, int dya_ld, int yb_ld, int dyb_ld, void* stream) {
  VS_CHECK_ARG(dz && relu_bits && y_a && y_b && dy_a && dy_b, "null tensor");
  VS_CHECK_ARG(mean_a && invstd_a && gamma_a && dgamma_a && dbeta_a && mean_b && invstd_b && gamma_b && dgamma_b && dbeta_b,
               "null statistics");
  VS_CHECK_ARG(C % 8 == 0 && cpr_pow2(C), "C/8 must be a power of two");
  BnbUnit A{(const uint16_t*)y_a, mean_a, invstd_a, gamma_a, dgamma_a, dbeta_a, (uint16_t*)dy_a, ya_ld, dya_ld};
  BnbUnit B{(const uint16_t*)y_b, mean_b, invstd_b, gamma_b, dgamma_b, dbeta_b, (uint16_t*)dy_b, yb_ld, dyb_ld};
  const int nb = bn_rows_batches(rows, C, 2048);
  const int cpr = C / 8, ncol = cpr < 256 ? cpr : 256;
  const long long rpb = (long long)(256 / ncol) * BNA_BATCH * nb;
  hipLaunchKernelGGL(bn_bwd_apply2_cols_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0,
                     (hipStream_t)stream, (const uint16_t*)dz, relu_bits, A, B, (long long)rows, C, dz_ld, nb);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

extern "C" int vs_bn_bwd_apply(const void* dz, const void* z, const void* y, const float* mean,
                               const float* invstd, const float* gamma, const float* beta,
                               const float* dgamma, const float* dbeta, void* dy, void* dres,
                               int64_t rows, int C, int dz_ld, int z_ld, int y_ld, int dy_ld,
                               int dres_ld, int relu, void* stream) {
  VS_CHECK_ARG(dz && y && mean && invstd && gamma && dgamma && dbeta && dy, "null tensor");
  VS_CHECK_ARG(!relu || z || beta, "relu needs z, or beta to recompute the mask");
  VS_CHECK_ARG(C % 8 == 0, "C must be a multiple of 8");
  hipStream_t st = (hipStream_t)stream;
#define BNB_ARGS (const uint16_t*)dz, (const uint16_t*)z, (const uint16_t*)y, mean, invstd, gamma, beta, \
                 dgamma, dbeta, (uint16_t*)dy, (uint16_t*)dres, (long long)rows, C, dz_ld, z_ld, y_ld, \
                 dy_ld, dres_ld
  if (cpr_pow2(C)) {
    const int nb = bn_rows_batches(rows, C, 2048);
    const int cpr = C / 8, ncol = cpr < 256 ? cpr : 256;
    const long long rpb = (long long)(256 / ncol) * BNA_BATCH * nb;
    const dim3 grid((unsigned)((rows + rpb - 1) / rpb)), block(256);
    const int mask = !relu ? 0 : (relu == 2 ? 3 : (z ? 1 : 2));
#define BNB_L(M, D) hipLaunchKernelGGL((bn_bwd_apply_cols_kernel<M, D>), grid, block, 0, st, BNB_ARGS, nb)
    if (dres) {
      if (mask == 0) BNB_L(0, true); else if (mask == 1) BNB_L(1, true);
      else if (mask == 2) BNB_L(2, true); else BNB_L(3, true);
    } else {
      if (mask == 0) BNB_L(0, false); else if (mask == 1) BNB_L(1, false);
      else if (mask == 2) BNB_L(2, false); else BNB_L(3, false);
    }
#undef BNB_L
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  VS_CHECK_ARG(relu != 2, "the ReLU bit mask needs C/8 to be a power of two");
  const dim3 grid(ew_grid(rows * (C / 8))), block(256);
  if (!relu) hipLaunchKernelGGL(bn_bwd_apply_kernel<0>, grid, block, 0, st, BNB_ARGS);
  else if (z) hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, grid, block, 0, st, BNB_ARGS);
  else hipLaunchKernelGGL(bn_bwd_apply_kernel<2>, grid, block, 0, st, BNB_ARGS);
#undef BNB_ARGS
  VS_CHECK_LAUNCH();
  return VS_OK;
}

static int pool_src(PoolSrc& ps, const void* dp, const uint8_t* idx, int N, int T, int H, int W, int C, int dp_ld) {
  if (!dp || !idx || C % 8 != 0 || dp_ld % 8 != 0 || !bnb_check(C) || (long long)N * T * H * W >= (1ll << 24))
    return 0;
  ps.dp = (const uint16_t*)dp;
  ps.idx = idx;
  ps.H = H;
  ps.W = W;
  ps.Ho = (H + 2 - 3) / 2 + 1;
  ps.Wo = (W + 2 - 3) / 2 + 1;
  ps.dp_ld = dp_ld;
  ps.rcp_w = 1.0f / (float)W;
  ps.rcp_h = 1.0f / (float)H;
  return 1;
}

// vs_bn_bwd_reduce / vs_bn_bwd_apply of a unit whose output went through the [1,3,3] / [1,2,2] max-pool and nothing
// else (the stems; ReLU mask recomputed from y): the output gradient is (d_pooled, idx) instead of a dense tensor.
extern "C" int vs_bn_bwd_reduce_pool(const void* d_pooled, const uint8_t* idx, const void* y, const float* mean,
                                     const float* invstd, const float* gamma, const float* beta, float* partial,
                                     int N, int T, int H, int W, int C, int dp_ld, int y_ld, void* stream) {
  VS_CHECK_ARG(y && mean && invstd && gamma && beta && partial, "null tensor");
  PoolSrc ps;
  VS_CHECK_ARG(pool_src(ps, d_pooled, idx, N, T, H, W, C, dp_ld), "C/8 a power of two, fewer than 2^24 rows");
  const long long rows = (long long)N * T * H * W;
  const int nblk = vs_bn_bwd_reduce_rows(rows, C);
  hipLaunchKernelGGL((bn_bwd_reduce_kernel<2, true>), dim3(nblk), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)nullptr, (const uint16_t*)nullptr, (const uint16_t*)y, mean, invstd, gamma,
                     beta, partial, rows, C, 0, 0, y_ld, bnb_batches(rows, C), ps);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

extern "C" int vs_bn_bwd_apply_pool(const void* d_pooled, const uint8_t* idx, const void* y, const float* mean,
                                    const float* invstd, const float* gamma, const float* beta, const float* dgamma,
                                    const float* dbeta, void* dy, int N, int T, int H, int W, int C, int dp_ld,
                                    int y_ld, int dy_ld, void* stream) {
  VS_CHECK_ARG(y && mean && invstd && gamma && beta && dgamma && dbeta && dy, "null tensor");
  PoolSrc ps;
  VS_CHECK_ARG(pool_src(ps, d_pooled, idx, N, T, H, W, C, dp_ld), "C/8 a power of two, fewer than 2^24 rows");
  const long long rows = (long long)N * T * H * W;
  const int nb = bn_rows_batches(rows, C, 2048);
  const int cpr = C / 8, ncol = cpr < 256 ? cpr : 256;
  const long long rpb = (long long)(256 / ncol) * BNA_BATCH * nb;
  hipLaunchKernelGGL((bn_bwd_apply_cols_kernel<2, false, true>), dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256),
                     0, (hipStream_t)stream, (const uint16_t*)nullptr, (const uint16_t*)nullptr, (const uint16_t*)y,
                     mean, invstd, gamma, beta, dgamma, dbeta, (uint16_t*)dy, (uint16_t*)nullptr, rows, C, 0, 0, y_ld,
                     dy_ld, 0, nb, ps);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ----------------------------------------------------------------------------
// fp32 residual stream (eval only, VS_RESIDUAL_FP32): out32 = relu?(residual + branch), out16 = bf16(out32).
// `branch` is the c unit's folded-BN output (bf16); the residual is the previous block's fp32 output, or -- first
// block of a stage -- the bf16 output of the shortcut unit.  The identity chain of a stage then never passes
// through bf16: what the next convolution reads is one rounding of the exact stream, not the 16th.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void residual_add_f32_kernel(const uint16_t* branch, const float* res32,
                                                               const uint16_t* res16, float* out32, uint16_t* out16,
                                                               long long rows, int C, int b_ld, int r_ld, int o32_ld,
                                                               int o16_ld, int relu) {
  const int cpr = C >> 3;
  const long long total = rows * cpr;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long row = idx / cpr;
    const int c = (int)(idx - row * cpr) * 8;
    float v[8], r[8];
    unpack8_bf16(*(const uint4*)(branch + row * b_ld + c), v);
    if (res32) {
      *(float4*)(r) = *(const float4*)(res32 + row * r_ld + c);
      *(float4*)(r + 4) = *(const float4*)(res32 + row * r_ld + c + 4);
    } else {
      unpack8_bf16(*(const uint4*)(res16 + row * r_ld + c), r);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] += r[e];
      if (relu) v[e] = fmaxf(v[e], 0.f);
    }
    *(float4*)(out32 + row * o32_ld + c) = *(const float4*)(v);
    *(float4*)(out32 + row * o32_ld + c + 4) = *(const float4*)(v + 4);
    *(uint4*)(out16 + row * o16_ld + c) = pack8_bf16(v);
  }
}

extern "C" int vs_residual_add_f32(const void* branch, const float* res32, const void* res16, float* out32,
                                   void* out16, int64_t rows, int C, int b_ld, int r_ld, int o32_ld, int o16_ld,
                                   int relu, void* stream) {
  VS_CHECK_ARG(branch && out32 && out16 && ((res32 != nullptr) != (res16 != nullptr)), "one residual, two outputs");
  VS_CHECK_ARG(C % 8 == 0 && b_ld % 8 == 0 && r_ld % 8 == 0 && o32_ld % 4 == 0 && o16_ld % 8 == 0,
               "channels / pitches must be multiples of 8");
  hipLaunchKernelGGL(residual_add_f32_kernel, dim3(ew_grid(rows * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)branch, res32, (const uint16_t*)res16, out32, (uint16_t*)out16, (long long)rows,
                     C, b_ld, r_ld, o32_ld, o16_ld, relu);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ----------------------------------------------------------------------------
// MaxPool3d([1,3,3], s[1,2,2], p[0,1,1]); idx = first max in (kh,kw) scan order
// ----------------------------------------------------------------------------
// (SMALL: fewer than 2^24 elements -- every tensor of the bench step: the element index is split with float-reciprocal
//  divisions; three 64-bit integer divisions per 16-byte element made these kernels instruction bound)
template <bool SMALL>
__global__ void maxpool_hw_fwd_kernel(const uint16_t* x, uint16_t* y, uint8_t* idx, int NT, int H,
                                      int W, int Ho, int Wo, int C, int x_ld, int y_ld) {
  const int cpr = C >> 3;
  const long long total = (long long)NT * Ho * Wo * cpr;
  const float rcp_c = 1.0f / (float)cpr, rcp_w = 1.0f / (float)Wo, rcp_h = 1.0f / (float)Ho;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int c, wo, ho;
    long long nt;
    if (SMALL) {
      int p1, p2, p3, cb;
      fast_divmod((int)i, cpr, rcp_c, p1, cb);
      fast_divmod(p1, Wo, rcp_w, p2, wo);
      fast_divmod(p2, Ho, rcp_h, p3, ho);
      c = cb * 8;
      nt = p3;
    } else {
      c = (int)(i % cpr) * 8;
      long long pos = i / cpr;
      wo = (int)(pos % Wo);
      pos /= Wo;
      ho = (int)(pos % Ho);
      nt = pos / Ho;
    }
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      best[e] = -INFINITY;
      bi[e] = 0;
    }
    for (int dh = 0; dh < 3; ++dh) {
      const int h = 2 * ho - 1 + dh;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int dw = 0; dw < 3; ++dw) {
        const int w = 2 * wo - 1 + dw;
        if ((unsigned)w >= (unsigned)W) continue;
        float v[8];
        unpack8_bf16(*(const uint4*)(x + ((nt * H + h) * W + w) * x_ld + c), v);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (v[e] > best[e] || (v[e] != v[e] && best[e] == best[e])) {  // NaN propagates as torch
            best[e] = v[e];
            bi[e] = dh * 3 + dw;
          }
      }
    }
    const long long opos = (nt * Ho + ho) * Wo + wo;
    *(uint4*)(y + opos * y_ld + c) = pack8_bf16(best);
    if (idx) {
      uint2 pk;
      pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
      pk.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
      *(uint2*)(idx + opos * C + c) = pk;
    }
  }
}

extern "C" int vs_maxpool_hw3s2_fwd(const void* x, void* y, uint8_t* idx, int N, int T, int H,
                                    int W, int C, int x_ld, int y_ld, void* stream) {
  VS_CHECK_ARG(x && y, "null tensor");
  VS_CHECK_ARG(C % 8 == 0 && x_ld % 8 == 0 && y_ld % 8 == 0, "C / pitches multiple of 8");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long total = (long long)N * T * Ho * Wo * (C / 8);
  if (total < (1ll << 24))
    hipLaunchKernelGGL(maxpool_hw_fwd_kernel<true>, dim3(ew_grid(total)), dim3(256), 0,
                       (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, idx, N * T, H, W, Ho,
                       Wo, C, x_ld, y_ld);
  else
    hipLaunchKernelGGL(maxpool_hw_fwd_kernel<false>, dim3(ew_grid(total)), dim3(256), 0,
                       (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, idx, N * T, H, W, Ho,
                       Wo, C, x_ld, y_ld);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// gather form: every input element sums dy of the (<= 4) windows whose argmax it is
template <bool SMALL>
__global__ void maxpool_hw_bwd_kernel(const uint16_t* dy, const uint8_t* idx, uint16_t* dx, int NT,
                                      int H, int W, int Ho, int Wo, int C, int dy_ld, int dx_ld) {
  const int cpr = C >> 3;
  const long long total = (long long)NT * H * W * cpr;
  const float rcp_c = 1.0f / (float)cpr, rcp_w = 1.0f / (float)W, rcp_h = 1.0f / (float)H;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int c, w, h;
    long long nt;
    if (SMALL) {
      int p1, p2, p3, cb;
      fast_divmod((int)i, cpr, rcp_c, p1, cb);
      fast_divmod(p1, W, rcp_w, p2, w);
      fast_divmod(p2, H, rcp_h, p3, h);
      c = cb * 8;
      nt = p3;
    } else {
      c = (int)(i % cpr) * 8;
      long long pos = i / cpr;
      w = (int)(pos % W);
      pos /= W;
      h = (int)(pos % H);
      nt = pos / H;
    }
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    // windows ho with 2ho-1 <= h <= 2ho+1
    const int ho_lo = h >> 1, ho_hi = (h + 1) >> 1;
    const int wo_lo = w >> 1, wo_hi = (w + 1) >> 1;
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
      if (ho >= Ho) continue;
      const int dh = h - (2 * ho - 1);
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        if (wo >= Wo) continue;
        const int dw = w - (2 * wo - 1);
        const int tap = dh * 3 + dw;
        const long long opos = (nt * Ho + ho) * Wo + wo;
        const uint2 pk = *(const uint2*)(idx + opos * C + c);
        float g[8];
        unpack8_bf16(*(const uint4*)(dy + opos * dy_ld + c), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned b = ((e < 4 ? pk.x : pk.y) >> ((e & 3) * 8)) & 0xff;
          if ((int)b == tap) acc[e] += g[e];
        }
      }
    }
    *(uint4*)(dx + ((nt * H + h) * W + w) * dx_ld + c) = pack8_bf16(acc);
  }
}

extern "C" int vs_maxpool_hw3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int T,
                                    int H, int W, int C, int dy_ld, int dx_ld, void* stream) {
  VS_CHECK_ARG(dy && idx && dx, "null tensor");
  VS_CHECK_ARG(C % 8 == 0 && dy_ld % 8 == 0 && dx_ld % 8 == 0, "C / pitches multiple of 8");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long total = (long long)N * T * H * W * (C / 8);
  if (total < (1ll << 24))
    hipLaunchKernelGGL(maxpool_hw_bwd_kernel<true>, dim3(ew_grid(total)), dim3(256), 0,
                       (hipStream_t)stream, (const uint16_t*)dy, idx, (uint16_t*)dx, N * T, H, W, Ho,
                       Wo, C, dy_ld, dx_ld);
  else
    hipLaunchKernelGGL(maxpool_hw_bwd_kernel<false>, dim3(ew_grid(total)), dim3(256), 0,
                       (hipStream_t)stream, (const uint16_t*)dy, idx, (uint16_t*)dx, N * T, H, W, Ho,
                       Wo, C, dy_ld, dx_ld);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// MaxPool3d k = s = [kt,1,1]  (dense tensors, T % kt == 0)
__global__ void maxpool_t_fwd_kernel(const uint16_t* x, uint16_t* y, uint8_t* idx, int N, int T,
                                     long long HW, int C, int kt) {
  const int cpr = C >> 3, To = T / kt;
  const long long total = (long long)N * To * HW * cpr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpr) * 8;
    long long pos = i / cpr;
    const long long s = pos % HW;
    pos /= HW;
    const int to = (int)(pos % To);
    const long long n = pos / To;
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      best[e] = -INFINITY;
      bi[e] = 0;
    }
    for (int d = 0; d < kt; ++d) {
      float v[8];
      unpack8_bf16(*(const uint4*)(x + (((n * T + to * kt + d) * HW) + s) * C + c), v);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (v[e] > best[e] || (v[e] != v[e] && best[e] == best[e])) {
          best[e] = v[e];
          bi[e] = d;
        }
    }
    const long long opos = (n * To + to) * HW + s;
    *(uint4*)(y + opos * C + c) = pack8_bf16(best);
    if (idx) {
      uint2 pk;
      pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
      pk.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
      *(uint2*)(idx + opos * C + c) = pk;
    }
  }
}

extern "C" int vs_maxpool_t_fwd(const void* x, void* y, uint8_t* idx, int N, int T, int HW, int C,
                                int kt, void* stream) {
  VS_CHECK_ARG(x && y && C % 8 == 0 && kt > 0 && T % kt == 0, "bad args");
  const long long total = (long long)N * (T / kt) * HW * (C / 8);
  hipLaunchKernelGGL(maxpool_t_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)x, (uint16_t*)y, idx, N, T, (long long)HW, C, kt);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

__global__ void maxpool_t_bwd_kernel(const uint16_t* dy, const uint8_t* idx, uint16_t* dx, int N,
                                     int T, long long HW, int C, int kt) {
  const int cpr = C >> 3, To = T / kt;
  const long long total = (long long)N * T * HW * cpr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpr) * 8;
    long long pos = i / cpr;
    const long long s = pos % HW;
    pos /= HW;
    const int t = (int)(pos % T);
    const long long n = pos / T;
    const int to = t / kt, d = t - to * kt;
    const long long opos = (n * To + to) * HW + s;
    const uint2 pk = *(const uint2*)(idx + opos * C + c);
    float g[8], o[8];
    unpack8_bf16(*(const uint4*)(dy + opos * C + c), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned b = ((e < 4 ? pk.x : pk.y) >> ((e & 3) * 8)) & 0xff;
      o[e] = ((int)b == d) ? g[e] : 0.f;
    }
    *(uint4*)(dx + ((n * T + t) * HW + s) * C + c) = pack8_bf16(o);
  }
}

extern "C" int vs_maxpool_t_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int T, int HW,
                                int C, int kt, void* stream) {
  VS_CHECK_ARG(dy && idx && dx && C % 8 == 0 && kt > 0 && T % kt == 0, "bad args");
  const long long total = (long long)N * T * HW * (C / 8);
  hipLaunchKernelGGL(maxpool_t_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)dy, idx, (uint16_t*)dx, N, T, (long long)HW, C, kt);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ----------------------------------------------------------------------------
// AdaptiveAvgPool3d(1) + concat: out[n][c_off+c] = mean_rows x[n,row,c]  (fp32 out)
// block = 32 chunk-columns x 32 row lanes (1024 threads: the pool sits on the step's critical path between
// the trunk and the heads; 8 row lanes walked 49 rows each) ; grid = (ceil(cpr/32), N)
// ----------------------------------------------------------------------------
#define AVP_RL 32
__global__ __launch_bounds__(1024) void avgpool_fwd_kernel(const uint16_t* x, float* out,
                                                           long long rows, int C, int x_ld,
                                                           int out_ld, int c_off) {
  __shared__ float red[AVP_RL][32][8];
  const int cpr = C >> 3;
  const int col = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int cb = blockIdx.x * 32 + col;
  const long long n = blockIdx.y;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (cb < cpr) {
    const uint16_t* base = x + n * rows * x_ld + cb * 8;
    long long r = rl;
    for (; r + 3 * AVP_RL < rows; r += 4 * AVP_RL) {  // four rows of loads in flight
      uint4 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = *(const uint4*)(base + (r + AVP_RL * u) * x_ld);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[8];
        unpack8_bf16(q[u], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
      }
    }
    for (; r < rows; r += AVP_RL) {
      float v[8];
      unpack8_bf16(*(const uint4*)(base + r * x_ld), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][col][e] = acc[e];
  __syncthreads();
  if (rl < 8 && cb < cpr) {  // thread (rl = e, col): one output channel, fixed order over the row lanes
    const float inv = 1.0f / (float)rows;
    float s = 0.f;
    for (int r = 0; r < AVP_RL; ++r) s += red[r][col][rl];
    out[n * out_ld + c_off + cb * 8 + rl] = s * inv;
  }
}

extern "C" int vs_avgpool_fwd(const void* x, float* out, int N, int64_t rows_per_clip, int C,
                              int x_ld, int out_ld, int c_off, void* stream) {
  VS_CHECK_ARG(x && out && C % 8 == 0 && x_ld % 8 == 0, "bad args");
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3((C / 8 + 31) / 32, N), dim3(1024), 0,
                     (hipStream_t)stream, (const uint16_t*)x, out, (long long)rows_per_clip, C,
                     x_ld, out_ld, c_off);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

__global__ void avgpool_bwd_kernel(const float* dout, uint16_t* dx, long long rows, int C,
                                   int dx_ld, int dout_ld, int c_off, long long total) {
  const int cpr = C >> 3;
  const float inv = 1.0f / (float)rows;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpr) * 8;
    const long long pos = i / cpr;
    const long long n = pos / rows;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = dout[n * dout_ld + c_off + c + e] * inv;
    *(uint4*)(dx + pos * dx_ld + c) = pack8_bf16(v);
  }
}

extern "C" int vs_avgpool_bwd(const float* dout, void* dx, int N, int64_t rows_per_clip, int C,
                              int dx_ld, int dout_ld, int c_off, void* stream) {
  VS_CHECK_ARG(dout && dx && C % 8 == 0 && dx_ld % 8 == 0, "bad args");
  const long long total = (long long)N * rows_per_clip * (C / 8);
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     dout, (uint16_t*)dx, (long long)rows_per_clip, C, dx_ld, dout_ld, c_off, total);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// =============================================================================================
// Non-local block pieces (slowfast/models/nonlocal_helper.py `Nonlocal`, used by the i3d_r50_nl_8x8
// feature model: configs/vsitu_mdl_cfgs/Kinetics_c2_I3D_NLN_8x8_R50.yaml:25-28):
// MaxPool3d([1,2,2], stride [1,2,2]) on channels-last bf16 with a 2-bit argmax, row softmax of the
// theta.phi scores (bf16 in / out, fp32 math), its backward, and a bf16 column sum (conv bias gradients).
// The two batched matrix products run on the implicit-GEMM conv kernel (per clip, phi / g as "weights").
// =============================================================================================
__global__ void maxpool_hw2_fwd_kernel(const uint16_t* x, uint16_t* y, uint8_t* idx, long long NT, int H,
                                       int W, int C, long long total) {
  const int Ho = H >> 1, Wo = W >> 1, cpr = C >> 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cpr);
    long long r = i / cpr;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const long long nt = r / Ho;
    float best[8];
    uint8_t bi[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long long pos = (nt * H + ho * 2 + (q >> 1)) * W + wo * 2 + (q & 1);
      float v[8];
      unpack8_bf16(*(const uint4*)(x + pos * C + c8 * 8), v);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (q == 0 || v[e] > best[e]) {  // first maximum wins (torch's tie rule)
          best[e] = v[e];
          bi[e] = (uint8_t)q;
        }
    }
    const long long o = ((nt * Ho + ho) * Wo + wo) * C + c8 * 8;
    *(uint4*)(y + o) = pack8_bf16(best);
    *(uint2*)(idx + o) = make_uint2(bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24),
                                    bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24));
  }
}

extern "C" int vs_maxpool_hw2_fwd(const void* x, void* y, uint8_t* idx, int64_t NT, int H, int W, int C,
                                  void* stream) {
  VS_CHECK_ARG(x && y && idx && NT > 0 && H >= 2 && W >= 2 && C % 8 == 0, "bad args");
  const long long total = (long long)NT * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(maxpool_hw2_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)x, (uint16_t*)y, idx, (long long)NT, H, W, C, total);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// dx[input position] = dy[its window] where the position was the window's argmax, else 0 (windows do
// not overlap; rows / columns beyond 2*floor(H/2), 2*floor(W/2) get 0)
__global__ void maxpool_hw2_bwd_kernel(const uint16_t* dy, const uint8_t* idx, uint16_t* dx, long long NT,
                                       int H, int W, int C, long long total) {
  const int Ho = H >> 1, Wo = W >> 1, cpr = C >> 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cpr);
    long long r = i / cpr;
    const int w = (int)(r % W);
    r /= W;
    const int h = (int)(r % H);
    const long long nt = r / H;
    float out[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int ho = h >> 1, wo = w >> 1;
    if (ho < Ho && wo < Wo) {
      const long long o = ((nt * Ho + ho) * Wo + wo) * C + c8 * 8;
      float g[8];
      unpack8_bf16(*(const uint4*)(dy + o), g);
      const uint2 b = *(const uint2*)(idx + o);
      const int q = ((h & 1) << 1) | (w & 1);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int bi = ((e < 4 ? b.x : b.y) >> ((e & 3) * 8)) & 0xff;
        out[e] = bi == q ? g[e] : 0.f;
      }
    }
    *(uint4*)(dx + ((nt * H + h) * W + w) * C + c8 * 8) = pack8_bf16(out);
  }
}

extern "C" int vs_maxpool_hw2_bwd(const void* dy, const uint8_t* idx, void* dx, int64_t NT, int H, int W,
                                  int C, void* stream) {
  VS_CHECK_ARG(dy && idx && dx && NT > 0 && H >= 2 && W >= 2 && C % 8 == 0, "bad args");
  const long long total = (long long)NT * H * W * (C / 8);
  hipLaunchKernelGGL(maxpool_hw2_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)dy, idx, (uint16_t*)dx, (long long)NT, H, W, C, total);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// y[r, :] = softmax(x[r, :]) over P <= 4096 bf16 columns (P % 4 == 0), one wave per row, fp32 math;
// in place allowed.
#define SMR_MAX 16  // 8-byte pieces per lane
__global__ __launch_bounds__(256) void softmax_rows_bf16_kernel(const uint16_t* x, uint16_t* y, long long rows,
                                                                int P) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int P4 = P >> 2;
  const uint2* xr = (const uint2*)(x + row * P);
  float v[SMR_MAX][4];
  float mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < SMR_MAX; ++e) {
    const int i = lane + 64 * e;
    const bool ok = i < P4;
    const uint2 q = xr[ok ? i : 0];
    v[e][0] = ok ? bf16_to_f32((uint16_t)(q.x & 0xffff)) : -INFINITY;
    v[e][1] = ok ? bf16_to_f32((uint16_t)(q.x >> 16)) : -INFINITY;
    v[e][2] = ok ? bf16_to_f32((uint16_t)(q.y & 0xffff)) : -INFINITY;
    v[e][3] = ok ? bf16_to_f32((uint16_t)(q.y >> 16)) : -INFINITY;
    mx = fmaxf(mx, fmaxf(fmaxf(v[e][0], v[e][1]), fmaxf(v[e][2], v[e][3])));
  }
  mx = wave_reduce_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < SMR_MAX; ++e)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[e][k] = expf(v[e][k] - mx);  // exp(-inf) = 0 for the padding
      sum += v[e][k];
    }
  const float inv = 1.0f / wave_reduce_sum(sum);
  uint2* yr = (uint2*)(y + row * P);
#pragma unroll
  for (int e = 0; e < SMR_MAX; ++e) {
    const int i = lane + 64 * e;
    if (i < P4) yr[i] = make_uint2(pack2_bf16(v[e][0] * inv, v[e][1] * inv), pack2_bf16(v[e][2] * inv, v[e][3] * inv));
  }
}

extern "C" int vs_softmax_rows_bf16(const void* x, void* y, int64_t rows, int P, void* stream) {
  VS_CHECK_ARG(x && y && rows > 0 && P > 0 && (P & 3) == 0 && P <= 64 * 4 * SMR_MAX, "P % 4 == 0, P <= 4096");
  hipLaunchKernelGGL(softmax_rows_bf16_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, (long long)rows, P);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ds = scale * p * (dp - sum_j dp_j p_j): backward of softmax(scale * s) w.r.t. s; in place on dp allowed
__global__ __launch_bounds__(256) void softmax_rows_bwd_bf16_kernel(const uint16_t* p, const uint16_t* dp,
                                                                    uint16_t* ds, long long rows, int P,
                                                                    float scale) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int P4 = P >> 2;
  const uint2* pr = (const uint2*)(p + row * P);
  const uint2* gr = (const uint2*)(dp + row * P);
  float pv[SMR_MAX][4], gv[SMR_MAX][4];
  float dot = 0.f;
#pragma unroll
  for (int e = 0; e < SMR_MAX; ++e) {
    const int i = lane + 64 * e;
    const bool ok = i < P4;
    const uint2 a = pr[ok ? i : 0], b = gr[ok ? i : 0];
    pv[e][0] = ok ? bf16_to_f32((uint16_t)(a.x & 0xffff)) : 0.f;
    pv[e][1] = ok ? bf16_to_f32((uint16_t)(a.x >> 16)) : 0.f;
    pv[e][2] = ok ? bf16_to_f32((uint16_t)(a.y & 0xffff)) : 0.f;
    pv[e][3] = ok ? bf16_to_f32((uint16_t)(a.y >> 16)) : 0.f;
    gv[e][0] = bf16_to_f32((uint16_t)(b.x & 0xffff));
    gv[e][1] = bf16_to_f32((uint16_t)(b.x >> 16));
    gv[e][2] = bf16_to_f32((uint16_t)(b.y & 0xffff));
    gv[e][3] = bf16_to_f32((uint16_t)(b.y >> 16));
#pragma unroll
    for (int k = 0; k < 4; ++k) dot += pv[e][k] * gv[e][k];  // padding: p = 0
  }
  dot = wave_reduce_sum(dot);
  uint2* dr = (uint2*)(ds + row * P);
#pragma unroll
  for (int e = 0; e < SMR_MAX; ++e) {
    const int i = lane + 64 * e;
    if (i < P4) {
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = scale * pv[e][k] * (gv[e][k] - dot);
      dr[i] = make_uint2(pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3]));
    }
  }
}

extern "C" int vs_softmax_rows_bwd_bf16(const void* p, const void* dp, void* ds, int64_t rows, int P,
                                        float scale, void* stream) {
  VS_CHECK_ARG(p && dp && ds && rows > 0 && P > 0 && (P & 3) == 0 && P <= 64 * 4 * SMR_MAX, "P % 4 == 0, P <= 4096");
  hipLaunchKernelGGL(softmax_rows_bwd_bf16_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, (const uint16_t*)p, (const uint16_t*)dp, (uint16_t*)ds,
                     (long long)rows, P, scale);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// out[c] = sum_rows x[row][c] (bf16 in, fp32 out): a block owns 64 columns, 16 waves take rows
// w, w+16, ..., eight rows of loads in flight, partial sums combined through LDS in wave order.
__global__ __launch_bounds__(1024) void colsum_bf16_kernel(const uint16_t* x, float* out, long long rows, int C,
                                                           int ld) {
  __shared__ float part[16][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.x * 64 + lane;
  const int cc = c < C ? c : 0;
  float acc = 0.f;
  long long r = wave;
  for (; r + 16 * 7 < rows; r += 16 * 8) {
    uint16_t q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = x[(r + 16 * u) * ld + cc];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += bf16_to_f32(q[u]);
  }
  for (; r < rows; r += 16) acc += bf16_to_f32(x[r * ld + cc]);
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < C) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += part[w][lane];
    out[c] = s;
  }
}

extern "C" int vs_colsum_bf16(const void* x, float* out, int64_t rows, int C, int ld, void* stream) {
  VS_CHECK_ARG(x && out && rows > 0 && C > 0 && ld >= C, "bad args");
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3((C + 63) / 64), dim3(1024), 0, (hipStream_t)stream,
                     (const uint16_t*)x, out, (long long)rows, C, ld);
  VS_CHECK_LAUNCH();
  return VS_OK;
}
