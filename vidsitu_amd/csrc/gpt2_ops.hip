// GPT-2 decoder kernels (fp32, gfx950): the language model the reference decodes SRL
// arguments with -- vidsitu_code/hf_gpt2_fseq.py:124-215 (huggingface GPT2LMHeadModel:
// Conv1D projections, causal + key-padding attention, gelu_new MLP, tied lm_head) -- and the
// per-step scoring of its beam search (vidsitu_code/seq_gen.py:310-385 + fairseq BeamSearch.step).
// All arithmetic is fp32 like the reference (no AMP anywhere): dense projections
// run on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, 157 TFLOP/s dense on MI355X) when there
// are more than 64 rows and as weight-streaming dot products (txenc_ops.hip) below that.
#include <math.h>

#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float gelu_new_f(float x) {
  // transformers activations.gelu_new
  return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}

// ----------------------------------------------------------------------------
// y[M,N] = act(x[M,K] . w[N,K]^T + b) + res      ("NT": both operands K-contiguous)
// 128x128x32 (or 64x64x32) tile, 4 waves as 2x2, each wave a block of 32x32 MFMA accumulators.
// LDS rows are K-major with an odd pitch (33 floats): the fragment read of a 32x32x2 MFMA
// (lane l: row l&31, k = l>>5) is then conflict free.  Register-staged double buffer.
// ----------------------------------------------------------------------------
#define GM_BK 32
#define GM_LD 33

// Split-K (gridDim.z = S > 1): slice z covers k-tiles [z * kts, (z + 1) * kts) and stores its raw partial
// tile into slab z of `y` (M * N floats each); gemm_splitk_reduce_kernel adds the slabs in slice order and
// applies the epilogue.  Layers with few output tiles (600 x 1024: 160 tiles of 64 x 64 for 256 CUs) then
// run 3-4 blocks per CU and overlap each other's memory latency.
template <int BM, int BN, bool KVEC>  // 128x128 (2x2 MFMA tiles per wave) or 64x64 (one per wave); K % 4 == 0
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          const float* res, float* y, int M, int N,
                                                          int K, int act, int kts) {
  constexpr int MR = BM / 64, NR = BN / 64;  // 32x32 accumulators per wave
  constexpr int PA = BM / 32, PB = BN / 32;  // loader passes (32 rows each)
  __shared__ float As[2][BM * GM_LD];
  __shared__ float Bs[2][BN * GM_LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  // loader mapping: 8 float4 per 32-float row, 32 rows per pass
  const int lc = tid & 7, lr = tid >> 3;
  constexpr bool kvec = KVEC;  // compile time: a runtime flag put every load behind a branch (and a wait)
  float4 ra[PA], rb[PB];

  // Loads are branch-free (a load inside `if (ok)` sits in its own basic block behind its own wait: the 8
  // loads of a k-tile were 8 serialised round trips) and RAW: out-of-range lanes read element 0 and are
  // zeroed only when the tile is written to LDS, so nothing touches the loaded registers -- and no wait is
  // placed -- before the MFMAs of the current tile have been issued.
  unsigned okmask = 0;
  auto ld4 = [&](const float* base, int row, int rows, int k, int bit) __attribute__((always_inline)) {
    const bool ok = row < rows && k < K;
    okmask = ok ? (okmask | (1u << bit)) : (okmask & ~(1u << bit));
    if constexpr (kvec) return *(const float4*)(base + (ok ? (long long)row * K + k : 0));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
      const float* s = base + (long long)row * K + k;
      v.x = s[0]; if (k + 1 < K) v.y = s[1]; if (k + 2 < K) v.z = s[2]; if (k + 3 < K) v.w = s[3];
    }
    return v;
  };
  auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PA; ++i) ra[i] = ld4(x, m0 + lr + 32 * i, M, k0 + lc * 4, i);
#pragma unroll
    for (int i = 0; i < PB; ++i) rb[i] = ld4(w, n0 + lr + 32 * i, N, k0 + lc * 4, PA + i);
  };
  auto sstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      float* a = &As[buf][(lr + 32 * i) * GM_LD + lc * 4];
      const bool ok = (okmask >> i) & 1;
      a[0] = ok ? ra[i].x : 0.f; a[1] = ok ? ra[i].y : 0.f; a[2] = ok ? ra[i].z : 0.f; a[3] = ok ? ra[i].w : 0.f;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      float* b = &Bs[buf][(lr + 32 * i) * GM_LD + lc * 4];
      const bool ok = (okmask >> (PA + i)) & 1;
      b[0] = ok ? rb[i].x : 0.f; b[1] = ok ? rb[i].y : 0.f; b[2] = ok ? rb[i].z : 0.f; b[3] = ok ? rb[i].w : 0.f;
    }
  };

  f32x16 acc[MR][NR];
#pragma unroll
  for (int a = 0; a < MR; ++a)
#pragma unroll
    for (int b = 0; b < NR; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int fr = lane & 31, fk = lane >> 5;
  const int nk_all = (K + GM_BK - 1) / GM_BK;
  const bool split = gridDim.z > 1;
  const int kt0 = split ? blockIdx.z * kts : 0;
  const int nk = split ? min(nk_all, kt0 + kts) : nk_all;
  if (split) y += (long long)blockIdx.z * M * N;
  gload(kt0 * GM_BK);
  sstore(kt0 & 1);
  __syncthreads();
  for (int kt = kt0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * GM_BK);
    const float* A = &As[cur][(wm * (BM / 2) + fr) * GM_LD + fk];
    const float* B = &Bs[cur][(wn * (BN / 2) + fr) * GM_LD + fk];
#pragma unroll
    for (int ks = 0; ks < GM_BK; ks += 2) {
      float av[MR], bv[NR];
#pragma unroll
      for (int a = 0; a < MR; ++a) av[a] = A[a * 32 * GM_LD + ks];
#pragma unroll
      for (int b = 0; b < NR; ++b) bv[b] = B[b * 32 * GM_LD + ks];
#pragma unroll
      for (int a = 0; a < MR; ++a)
#pragma unroll
        for (int b = 0; b < NR; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(cur ^ 1);
    __syncthreads();
  }
  // D layout of 32x32x2: col = lane & 31, row = 8*(e/4) + 4*(lane>>5) + (e&3)
#pragma unroll
  for (int a = 0; a < MR; ++a)
#pragma unroll
    for (int b = 0; b < NR; ++b) {
      const int n = n0 + wn * (BN / 2) + b * 32 + (lane & 31);
      if (n >= N) continue;
      if (split) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm * (BM / 2) + a * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
          if (m < M) y[(long long)m * N + n] = acc[a][b][e];
        }
        continue;
      }
      const float bvv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * (BM / 2) + a * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
        if (m < M) {
          float v = acc[a][b][e] + bvv;
          if (act == 1) v = fmaxf(v, 0.f);
          else if (act == 2) v = gelu_new_f(v);
          if (res) v += res[(long long)m * N + n];
          y[(long long)m * N + n] = v;
        }
      }
    }
}

__global__ void gemm_splitk_reduce_kernel(const float* slabs, const float* bias, const float* res, float* y,
                                          long long MN, int N, int S, int act) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < MN; i += (long long)gridDim.x * blockDim.x) {
    float v = slabs[i];
    for (int s2 = 1; s2 < S; ++s2) v += slabs[(long long)s2 * MN + i];
    if (bias) v += bias[i % N];
    if (act == 1) v = fmaxf(v, 0.f);
    else if (act == 2) v = gelu_new_f(v);
    if (res) v += res[i];
    y[i] = v;
  }
}

// K slices for a GEMM with few output tiles: about 700 blocks, at least 4 k-tiles (128 floats) per slice
static int gemm_split(int M, int N, int K) {
  const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128);
  if (t128 >= 256) return 1;
  const long long t64 = (long long)((M + 63) / 64) * ((N + 63) / 64);
  const int nk = (K + GM_BK - 1) / GM_BK;
  int S = (int)(704 / t64);
  if (S > nk / 4) S = nk / 4;
  if (S > 16) S = 16;
  return S < 2 ? 1 : S;
}

extern "C" size_t vs_gemm_nt_f32_workspace_bytes(int M, int N, int K) {
  if (M <= 64) return 0;
  const int S = gemm_split(M, N, K);
  return S > 1 ? (size_t)S * M * N * sizeof(float) : 0;
}

int vs_gemm_nt_f32_mfma(const float* x, const float* w, const float* b, const float* res, float* y,
                        int M, int N, int K, int act, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0) {
  // fewer than one 128x128 tile per CU: 64x64 tiles (4x the blocks) keep the chip busy
  const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128);
  const bool kv = (K & 3) == 0 && ((((uintptr_t)x | (uintptr_t)w)) & 15) == 0;
  if (t128 >= 256) {
    if (kv)
      hipLaunchKernelGGL((gemm_nt_f32_kernel<128, 128, true>), dim3((N + 127) / 128, (M + 127) / 128),
                         dim3(256), 0, st, x, w, b, res, y, M, N, K, act, 0);
    else
      hipLaunchKernelGGL((gemm_nt_f32_kernel<128, 128, false>), dim3((N + 127) / 128, (M + 127) / 128),
                         dim3(256), 0, st, x, w, b, res, y, M, N, K, act, 0);
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  const int S = ws ? gemm_split(M, N, K) : 1;
  if (S > 1 && ws_bytes >= (size_t)S * M * N * sizeof(float)) {
    const int nk = (K + GM_BK - 1) / GM_BK;
    const int kts = (nk + S - 1) / S;
    const int Se = (nk + kts - 1) / kts;  // slices that own at least one k-tile
    if (kv)
      hipLaunchKernelGGL((gemm_nt_f32_kernel<64, 64, true>), dim3((N + 63) / 64, (M + 63) / 64, Se), dim3(256),
                         0, st, x, w, (const float*)nullptr, (const float*)nullptr, (float*)ws, M, N, K, 0, kts);
    else
      hipLaunchKernelGGL((gemm_nt_f32_kernel<64, 64, false>), dim3((N + 63) / 64, (M + 63) / 64, Se), dim3(256),
                         0, st, x, w, (const float*)nullptr, (const float*)nullptr, (float*)ws, M, N, K, 0, kts);
    const long long MN = (long long)M * N;
    long long grid = (MN + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)grid), dim3(256), 0, st, (const float*)ws, b,
                       res, y, MN, N, Se, act);
  } else if (kv) {
    hipLaunchKernelGGL((gemm_nt_f32_kernel<64, 64, true>), dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0,
                       st, x, w, b, res, y, M, N, K, act, 0);
  } else {
    hipLaunchKernelGGL((gemm_nt_f32_kernel<64, 64, false>), dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0,
                       st, x, w, b, res, y, M, N, K, act, 0);
  }
  VS_CHECK_LAUNCH();
  return VS_OK;
}

/* vs_gemm_nt_f32 with a split-K workspace (vs_gemm_nt_f32_workspace_bytes(M, N, K); 0 = not needed). */
extern "C" int vs_gemm_nt_f32_ws(const float* x, const float* w, const float* b, const float* res, float* y,
                                 int M, int N, int K, int act, void* workspace, size_t ws_bytes, void* stream) {
  VS_CHECK_ARG(x && w && y && M > 0 && N > 0 && K > 0, "bad args");
  VS_CHECK_ARG(act >= 0 && act <= 2, "act must be 0 (none), 1 (relu) or 2 (gelu_new)");
  if (M <= 64 || !workspace) return vs_gemm_nt_f32(x, w, b, res, y, M, N, K, act, stream);
  return vs_gemm_nt_f32_mfma(x, w, b, res, y, M, N, K, act, (hipStream_t)stream, workspace, ws_bytes);
}

// ----------------------------------------------------------------------------
// h[r, l, :] = wte[tok[r, l]] + wpe[pos0 + l]      (GPT2Model: default position ids)
// ----------------------------------------------------------------------------
__global__ void gpt2_embed_kernel(const int64_t* tok, const float* wte, const float* wpe, float* out,
                                  int rows, int L, int D, int pos0, int V) {
  const int row = blockIdx.x;  // r * L + l
  const int l = row % L;
  long long t = tok[row];
  if (t < 0 || t >= V) t = 0;  // never dereference outside the table
  const float4* a = (const float4*)(wte + t * D);
  const float4* p = (const float4*)(wpe + (long long)(pos0 + l) * D);
  float4* o = (float4*)(out + (long long)row * D);
  for (int i = threadIdx.x; i < D / 4; i += blockDim.x) {
    const float4 u = a[i], v = p[i];
    o[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
  }
}

extern "C" int vs_gpt2_embed(const int64_t* tokens, const float* wte, const float* wpe, float* out,
                             int R, int L, int D, int pos0, int V, void* stream) {
  VS_CHECK_ARG(tokens && wte && wpe && out && R > 0 && L > 0 && D % 4 == 0, "bad args");
  hipLaunchKernelGGL(gpt2_embed_kernel, dim3(R * L), dim3(256), 0, (hipStream_t)stream, tokens, wte,
                     wpe, out, R * L, L, D, pos0, V);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ----------------------------------------------------------------------------
// Causal self-attention over a fused qkv buffer [R, L, 3D] (c_attn output), modeling_gpt2
// Attention._attn:  w = q k^T / sqrt(dh);  w = where(j <= i, w, -1e4);  w += (1-mask_j) * -1e4;
// softmax over ALL j (the -1e4 entries underflow to exactly 0 unless a whole row is masked,
// in which case the reference's behaviour is reproduced too);  out = p v, heads merged.
// One block per (r, head): K and V of the head live in LDS (pitch dh+1), one wave per query.
// ----------------------------------------------------------------------------
// Grid = (r, head) x chunks of AC_QC queries: 160 blocks of 60 queries each were a latency-bound launch
// (91 us per layer); a chunk's block loads the K rows up to its last query and all V rows.
#define AC_QC 16
__global__ __launch_bounds__(256) void attn_causal_kernel(const float* qkv, const uint8_t* kmask,
                                                          float* out, int L, int H, int dh) {
  extern __shared__ float sm[];
  const int D = H * dh, ld = dh + 1;
  const int r = blockIdx.x / H, h = blockIdx.x % H;
  const int i0 = blockIdx.y * AC_QC, i1 = min(L, i0 + AC_QC);
  float* Ks = sm;
  float* Vs = Ks + L * ld;
  float* Ps = Vs + L * ld;   // [4][L]
  float* Qs = Ps + 4 * L;    // [4][dh]
  const float* base = qkv + (long long)r * L * 3 * D;
  for (int i = threadIdx.x; i < L * dh; i += 256) {
    const int j = i / dh, d = i - j * dh;
    if (j < i1) Ks[j * ld + d] = base[(long long)j * 3 * D + D + h * dh + d];  // keys beyond are masked
    Vs[j * ld + d] = base[(long long)j * 3 * D + 2 * D + h * dh + d];
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float scale = 1.0f / sqrtf((float)dh);
  float* P = Ps + wave * L;
  float* Q = Qs + wave * dh;
  for (int i = i0 + wave; i < i1; i += 4) {
    for (int d = lane; d < dh; d += 64) Q[d] = base[(long long)i * 3 * D + h * dh + d];
    __builtin_amdgcn_wave_barrier();
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) {
      float s = -1e4f;
      if (j <= i) {
        s = 0.f;
        for (int d = 0; d < dh; ++d) s += Q[d] * Ks[j * ld + d];
        s *= scale;
      }
      if (kmask && !kmask[(long long)r * L + j]) s += -1e4f;
      P[j] = s;
      mx = fmaxf(mx, s);
    }
    mx = wave_reduce_max(mx);
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) {
      const float e = expf(P[j] - mx);
      P[j] = e;
      sum += e;
    }
    sum = wave_reduce_sum(sum);
    __builtin_amdgcn_wave_barrier();
    const float inv = 1.0f / sum;
    for (int d = lane; d < dh; d += 64) {
      float o = 0.f;
      for (int j = 0; j < L; ++j) o += P[j] * Vs[j * ld + d];
      out[((long long)r * L + i) * D + h * dh + d] = o * inv;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

extern "C" int vs_attn_causal_fwd(const float* qkv, const uint8_t* key_mask, float* out, int R, int L,
                                  int H, int dh, void* stream) {
  VS_CHECK_ARG(qkv && out && R > 0 && L > 0 && H > 0 && dh > 0, "bad args");
  const size_t smem = ((size_t)2 * L * (dh + 1) + 4 * L + 4 * dh) * sizeof(float);
  VS_CHECK_ARG(smem <= 160 * 1024, "sequence too long for the LDS-resident attention (L*dh)");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_causal_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(attn_causal_kernel, dim3(R * H, (L + AC_QC - 1) / AC_QC), dim3(256), smem,
                     (hipStream_t)stream, qkv, key_mask, out, L, H, dh);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ----------------------------------------------------------------------------
// Incremental decoding step with a KV cache [rows][H][Lmax][dh]: append this step's k, v at
// position t, attend the single new query over positions 0..t (every cached key is causal-
// visible; key_mask as above).  One wave per (row, head).
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(64) void attn_decode_kernel(const float* qkv, float* kc, float* vc,
                                                         const uint8_t* kmask, float* out, int H,
                                                         int dh, int Lmax, int t) {
  extern __shared__ float sm[];  // P[t+1] + Q[dh]
  const int r = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
  const int D = H * dh;
  float* P = sm;
  float* Q = sm + (t + 1);
  float* kch = kc + ((long long)r * H + h) * Lmax * dh;
  float* vch = vc + ((long long)r * H + h) * Lmax * dh;
  const float* src = qkv + (long long)r * 3 * D;
  for (int d = lane; d < dh; d += 64) {
    Q[d] = src[h * dh + d];
    kch[(long long)t * dh + d] = src[D + h * dh + d];
    vch[(long long)t * dh + d] = src[2 * D + h * dh + d];
  }
  __syncthreads();  // one wave: orders the global writes above before the reads below
  const float scale = 1.0f / sqrtf((float)dh);
  float mx = -INFINITY;
  for (int j = lane; j <= t; j += 64) {
    float s = 0.f;
    for (int d = 0; d < dh; ++d) s += Q[d] * kch[(long long)j * dh + d];
    s *= scale;
    if (kmask && !kmask[(long long)r * Lmax + j]) s += -1e4f;
    P[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_reduce_max(mx);
  float sum = 0.f;
  for (int j = lane; j <= t; j += 64) {
    const float e = expf(P[j] - mx);
    P[j] = e;
    sum += e;
  }
  sum = wave_reduce_sum(sum);
  __syncthreads();
  const float inv = 1.0f / sum;
  for (int d = lane; d < dh; d += 64) {
    float o = 0.f;
    for (int j = 0; j <= t; ++j) o += P[j] * vch[(long long)j * dh + d];
    out[(long long)r * D + h * dh + d] = o * inv;
  }
}

// dh in {16, 32, 64}: dh/16 lanes share a key (16-byte loads, 64 contiguous bytes per lane), the new
// position's k / v come straight from qkv, and the cached rows of position j are read from cache row
// anc[r][j] when an ancestry table is given: a beam reorder then only permutes that small table
// (vs_beam_step) instead of copying the whole cache every step (vs_kv_gather: 590 MB per step for
// GPT-2 medium at 50 rows x 30 positions).
template <int DH>
__global__ __launch_bounds__(64) void attn_decode_anc_kernel(const float* qkv, float* kc, float* vc,
                                                             const uint8_t* kmask, const int* anc,
                                                             float* out, int H, int Lmax, int t, int opk) {
  extern __shared__ float sm[];  // P[t+1], then row index [t+1]
  constexpr int PARTS = DH / 16, KPP = 64 / PARTS;
  const int r = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
  const int D = H * DH;
  float* P = sm;
  int* rowj = (int*)(sm + (t + 1));
  const float* src = qkv + (long long)r * 3 * D + h * DH;
  for (int dd = lane; dd < DH; dd += 64) {
    const long long own = (((long long)r * H + h) * Lmax + t) * DH + dd;
    kc[own] = src[D + dd];
    vc[own] = src[2 * D + dd];
  }
  const int part = lane % PARTS, jj = lane / PARTS;
  float4 q4[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) q4[u] = *(const float4*)(src + part * 16 + u * 4);
  const float scale = 1.0f / sqrtf((float)DH);
  float mx = -INFINITY;
  for (int j0 = 0; j0 <= t; j0 += KPP) {
    const int j = j0 + jj;
    const bool ok = j <= t;
    const int jc = ok ? j : t;
    const int rj = (anc && jc < t) ? anc[(long long)r * Lmax + jc] : r;
    const float* kp = jc == t ? src + D + part * 16
                              : kc + (((long long)rj * H + h) * Lmax + jc) * DH + part * 16;
    float4 k4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) k4[u] = *(const float4*)(kp + u * 4);
    float sdot = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      sdot += (q4[u].x * k4[u].x + q4[u].y * k4[u].y) + (q4[u].z * k4[u].z + q4[u].w * k4[u].w);
#pragma unroll
    for (int o = 1; o < PARTS; o <<= 1) sdot += __shfl_xor(sdot, o, 64);
    sdot *= scale;
    if (kmask && !kmask[(long long)r * Lmax + jc]) sdot += -1e4f;
    if (ok) {
      if (part == 0) {
        P[j] = sdot;
        rowj[j] = rj;
      }
      mx = fmaxf(mx, sdot);
    }
  }
  mx = wave_reduce_max(mx);
  __syncthreads();
  float sum = 0.f;
  for (int j = lane; j <= t; j += 64) {
    const float e = expf(P[j] - mx);
    P[j] = e;
    sum += e;
  }
  sum = wave_reduce_sum(sum);
  __syncthreads();
  for (int dd = lane; dd < DH; dd += 64) {
    float o = P[t] * src[2 * D + dd];
    const float* vb = vc + (long long)h * Lmax * DH + dd;
    // 16 rows of V in flight per step (4 made the loop a chain of ~t/4 memory latencies); positions past
    // t - 1 are clamped and weighted 0, same summation order as a plain loop
    for (int j = 0; j < t; j += 16) {
      float vv[16], pp[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int jj = j + u < t ? j + u : t - 1;
        vv[u] = vb[((long long)rowj[jj] * H * Lmax + jj) * DH];
        pp[u] = j + u < t ? P[jj] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (j + u < t) o += pp[u] * vv[u];
    }
    const int kk = h * DH + dd;
    // opk: fragment-major output (vs_pack_rows_f32's layout), the x operand of vs_gemm_nt_f32_packed
    if (opk) out[(((long long)(r >> 4) * (D >> 4) + (kk >> 4)) * 64 + ((kk & 15) >> 2) * 16 + (r & 15)) * 4 + (kk & 3)] = o / sum;
    else out[(long long)r * D + kk] = o / sum;
  }
}

extern "C" int vs_attn_decode(const float* qkv, float* kcache, float* vcache, const uint8_t* key_mask,
                              const int32_t* ancestry, float* out, int rows, int H, int dh, int Lmax,
                              int t, int out_packed, void* stream) {
  VS_CHECK_ARG(qkv && kcache && vcache && out && rows > 0 && t >= 0 && t < Lmax, "bad args");
  VS_CHECK_ARG(!out_packed || (dh == 16 || dh == 32 || dh == 64 || dh == 128),
               "a packed output needs dh in {16, 32, 64, 128}");
  const size_t smem = (size_t)(2 * (t + 1) + dh) * sizeof(float);
  VS_CHECK_ARG(smem <= 64 * 1024, "cache too long");
  const bool al = (((uintptr_t)qkv | (uintptr_t)kcache) & 15) == 0;
  if (dh == 64 && al)
    hipLaunchKernelGGL(attn_decode_anc_kernel<64>, dim3(rows * H), dim3(64), smem, (hipStream_t)stream,
                       qkv, kcache, vcache, key_mask, ancestry, out, H, Lmax, t, out_packed);
  else if (dh == 32 && al)
    hipLaunchKernelGGL(attn_decode_anc_kernel<32>, dim3(rows * H), dim3(64), smem, (hipStream_t)stream,
                       qkv, kcache, vcache, key_mask, ancestry, out, H, Lmax, t, out_packed);
  else if (dh == 16 && al)
    hipLaunchKernelGGL(attn_decode_anc_kernel<16>, dim3(rows * H), dim3(64), smem, (hipStream_t)stream,
                       qkv, kcache, vcache, key_mask, ancestry, out, H, Lmax, t, out_packed);
  else if (dh == 128 && al)
    hipLaunchKernelGGL(attn_decode_anc_kernel<128>, dim3(rows * H), dim3(64), smem, (hipStream_t)stream,
                       qkv, kcache, vcache, key_mask, ancestry, out, H, Lmax, t, out_packed);
  else {
    VS_CHECK_ARG(!ancestry && !out_packed, "ancestry tables / packed outputs need dh in {16, 32, 64, 128}");
    hipLaunchKernelGGL(attn_decode_kernel, dim3(rows * H), dim3(64), smem, (hipStream_t)stream, qkv,
                       kcache, vcache, key_mask, out, H, dh, Lmax, t);
  }
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// Beam reorder of a KV cache: dst[r] = src[index[r]] for the first `len` positions of every head
// (fairseq reorder_incremental_state; seq_gen.py:305-313).
__global__ void kv_gather_kernel(const float* src, float* dst, const int64_t* index, int H, int dh,
                                 int Lmax, int len) {
  const int r = blockIdx.x / H, h = blockIdx.x % H;
  const float4* s = (const float4*)(src + (((long long)index[r]) * H + h) * Lmax * dh);
  float4* d = (float4*)(dst + ((long long)r * H + h) * Lmax * dh);
  const int n4 = len * dh / 4;
  for (int i = threadIdx.x; i < n4; i += blockDim.x) d[i] = s[i];
}

extern "C" int vs_kv_gather(const float* src, float* dst, const int64_t* index, int rows_out, int H,
                            int dh, int Lmax, int len, void* stream) {
  VS_CHECK_ARG(src && dst && index && rows_out > 0 && len >= 0 && len <= Lmax && dh % 4 == 0, "bad args");
  if (len == 0) return VS_OK;
  hipLaunchKernelGGL(kv_gather_kernel, dim3(rows_out * H), dim3(128), 0, (hipStream_t)stream, src,
                     dst, index, H, dh, Lmax, len);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ----------------------------------------------------------------------------
// Beam-search scoring of one step (seq_gen.py:318-353 + EnsembleModel.forward_decoder :845-853
// + fairseq BeamSearch.step):  lp = log_softmax(logits / T);  NaN -> -inf;  lp[pad] = -inf;
// lp[unk] -= unk_penalty;  flags & 1: only eos allowed (step >= max_len);  flags & 2: eos banned
// (step < min_len);  forced[r] >= 0: only that token keeps its score (prefix forcing);
// then + cum[r] and the k best (value, token) of the row, descending, ties -> lowest token.
// One block per row; every thread keeps its own top-k over a strided slice, merged through LDS.
// ----------------------------------------------------------------------------
#define BT_MAXK 32

__device__ __forceinline__ bool bt_better(float v, int i, float w, int j) {
  return v > w || (v == w && i < j);
}

__global__ __launch_bounds__(256) void beam_topk_kernel(const float* logits, const float* cum,
                                                        const int64_t* forced, float* out_val,
                                                        int64_t* out_idx, int V, int k, int pad,
                                                        int eos, int unk, float unk_penalty,
                                                        float inv_temp, int flags) {
  __shared__ float red[256];
  __shared__ float cv[256 * 4];
  __shared__ int ci[256 * 4];
  const int r = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (long long)r * V;
  float mx = -INFINITY;
  for (int j = tid; j < V; j += 256) mx = fmaxf(mx, x[j] * inv_temp);
  red[tid] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
    __syncthreads();
  }
  mx = red[0];
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < V; j += 256) sum += expf(x[j] * inv_temp - mx);
  red[tid] = sum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const float lse = mx + logf(red[0]);
  const float add = cum ? cum[r] : 0.f;
  const long long f = forced ? forced[r] : -1;
  const bool force = f >= 0 && f != pad;

  auto score = [&](int j) {
    float lp = x[j] * inv_temp - lse;
    if (lp != lp) lp = -INFINITY;
    if (j == pad) lp = -INFINITY;
    if (j == unk) lp -= unk_penalty;
    if ((flags & 1) && j != eos) lp = -INFINITY;
    if (force) {
      if (j != (int)f) lp = -INFINITY;
    } else if ((flags & 2) && j == eos) {
      lp = -INFINITY;
    }
    return lp + add;
  };
  // k rounds of a block-wide argmax; each round every thread rescans its slice for its best
  // candidate that is worse than the previous winner (value, index order) -- k <= 2*beam is small
  float pv = INFINITY;
  int pi = -1;
  for (int round = 0; round < k; ++round) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = tid; j < V; j += 256) {
      const float v = score(j);
      // strictly after the previous winner in (value desc, index asc) order
      const bool after = (v < pv) || (v == pv && j > pi);
      if (after && bt_better(v, j, bv, bi)) {
        bv = v;
        bi = j;
      }
    }
    cv[tid] = bv;
    ci[tid] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s && bt_better(cv[tid + s], ci[tid + s], cv[tid], ci[tid])) {
        cv[tid] = cv[tid + s];
        ci[tid] = ci[tid + s];
      }
      __syncthreads();
    }
    pv = cv[0];
    pi = ci[0];
    if (tid == 0) {
      out_val[(long long)r * k + round] = pv;
      out_idx[(long long)r * k + round] = pi;
    }
    __syncthreads();
  }
}

// ---- large vocabularies: the row is cut into slices of BT_SLICE tokens (grid = slices x rows
// instead of one block per row: 50 blocks x 12 passes over 50 k tokens took 550 us per step).
//   1. per slice: max and sum exp      2. per slice: the row's lse from the partials (fixed order,
//   the same bits in every block), exact scores of the slice in registers, k rounds of block argmax
//   3. per row: merge slices x k candidates.  Same (value desc, token asc) order as the one-block kernel.
#define BT_SLICE 2048
#define BT_E (BT_SLICE / 256)

__global__ __launch_bounds__(256) void beam_lse_part_kernel(const float* logits, float2* part, int V,
                                                            float inv_temp) {
  __shared__ float red[4];
  const int sl = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, S = gridDim.x;
  const float* x = logits + (long long)r * V;
  float v[BT_E];
  float mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < BT_E; ++e) {
    const int j = sl * BT_SLICE + e * 256 + tid;
    const bool ok = j < V;
    const float t = x[ok ? j : 0] * inv_temp;
    v[e] = ok ? t : -INFINITY;
    mx = fmaxf(mx, v[e]);
  }
  mx = wave_reduce_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < BT_E; ++e) {
    const int j = sl * BT_SLICE + e * 256 + tid;
    if (j < V) sum += expf(v[e] - mx);
  }
  sum = wave_reduce_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  if (tid == 0) part[(long long)r * S + sl] = make_float2(mx, (red[0] + red[1]) + (red[2] + red[3]));
}

__device__ __forceinline__ void bt_block_argbest(float& bv, int& bi, float* cv, int* ci, int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (bt_better(ov, oi, bv, bi)) {
      bv = ov;
      bi = oi;
    }
  }
  __syncthreads();  // previous round's readers are done
  if ((tid & 63) == 0) {
    cv[tid >> 6] = bv;
    ci[tid >> 6] = bi;
  }
  __syncthreads();
  bv = cv[0];
  bi = ci[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (bt_better(cv[w], ci[w], bv, bi)) {
      bv = cv[w];
      bi = ci[w];
    }
}

__global__ __launch_bounds__(256) void beam_topk_part_kernel(
    const float* logits, const float2* part, const float* cum, const int64_t* forced, float* cand_val,
    int* cand_idx, int V, int k, int pad, int eos, int unk, float unk_penalty, float inv_temp, int flags) {
  __shared__ float cv[4];
  __shared__ int ci[4];
  const int sl = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, S = gridDim.x;
  const float* x = logits + (long long)r * V;
  float M = -INFINITY;
  for (int q = 0; q < S; ++q) M = fmaxf(M, part[(long long)r * S + q].x);
  float sum = 0.f;
  for (int q = 0; q < S; ++q) {
    const float2 pq = part[(long long)r * S + q];
    sum += pq.y * expf(pq.x - M);
  }
  const float lse = M + logf(sum);
  const float add = cum ? cum[r] : 0.f;
  const long long f = forced ? forced[r] : -1;
  const bool force = f >= 0 && f != pad;
  float v[BT_E];
  int idx[BT_E];
#pragma unroll
  for (int e = 0; e < BT_E; ++e) {
    const int j = sl * BT_SLICE + e * 256 + tid;
    const bool ok = j < V;
    float lp = x[ok ? j : 0] * inv_temp - lse;
    if (lp != lp) lp = -INFINITY;
    if (j == pad) lp = -INFINITY;
    if (j == unk) lp -= unk_penalty;
    if ((flags & 1) && j != eos) lp = -INFINITY;
    if (force) {
      if (j != (int)f) lp = -INFINITY;
    } else if ((flags & 2) && j == eos) {
      lp = -INFINITY;
    }
    v[e] = ok ? lp + add : -INFINITY;
    idx[e] = ok ? j : 0x7fffffff;
  }
  for (int round = 0; round < k; ++round) {
    float bv = v[0];
    int bi = idx[0];
#pragma unroll
    for (int e = 1; e < BT_E; ++e)
      if (bt_better(v[e], idx[e], bv, bi)) {
        bv = v[e];
        bi = idx[e];
      }
    bt_block_argbest(bv, bi, cv, ci, tid);
#pragma unroll
    for (int e = 0; e < BT_E; ++e)
      if (idx[e] == bi) {  // consumed (a padding sentinel never matches a live entry again)
        v[e] = -INFINITY;
        idx[e] = 0x7fffffff;
      }
    if (tid == 0) {
      cand_val[((long long)r * S + sl) * k + round] = bv;
      cand_idx[((long long)r * S + sl) * k + round] = bi;
    }
  }
}

__global__ __launch_bounds__(256) void beam_topk_merge_kernel(const float* cand_val, const int* cand_idx,
                                                              float* out_val, int64_t* out_idx, int n,
                                                              int k) {
  __shared__ float cv[4];
  __shared__ int ci[4];
  const int r = blockIdx.x, tid = threadIdx.x;
  float v[4];
  int idx[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = e * 256 + tid;
    const bool ok = c < n;
    v[e] = ok ? cand_val[(long long)r * n + (ok ? c : 0)] : -INFINITY;
    idx[e] = ok ? cand_idx[(long long)r * n + (ok ? c : 0)] : 0x7fffffff;
  }
  for (int round = 0; round < k; ++round) {
    float bv = v[0];
    int bi = idx[0];
#pragma unroll
    for (int e = 1; e < 4; ++e)
      if (bt_better(v[e], idx[e], bv, bi)) {
        bv = v[e];
        bi = idx[e];
      }
    bt_block_argbest(bv, bi, cv, ci, tid);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (idx[e] == bi) {
        v[e] = -INFINITY;
        idx[e] = 0x7fffffff;
      }
    if (tid == 0) {
      out_val[(long long)r * k + round] = bv;
      out_idx[(long long)r * k + round] = bi;
    }
  }
}

static inline int bt_slices(int V) { return (V + BT_SLICE - 1) / BT_SLICE; }

extern "C" size_t vs_beam_topk_workspace_bytes(int rows, int V, int k) {
  const size_t S = (size_t)bt_slices(V);
  return (size_t)rows * S * 8 + (size_t)rows * S * (size_t)k * 8 + 256;
}

extern "C" int vs_beam_topk(const float* logits, const float* cum, const int64_t* forced,
                            float* out_val, int64_t* out_idx, int rows, int V, int k, int pad,
                            int eos, int unk, float unk_penalty, float temperature, int flags,
                            void* workspace, size_t ws_bytes, void* stream) {
  VS_CHECK_ARG(logits && out_val && out_idx && rows > 0 && V > 1, "bad args");
  VS_CHECK_ARG(k >= 1 && k <= BT_MAXK && k < V, "k must be in [1, 32] and < V");
  VS_CHECK_ARG(temperature > 0.f, "temperature must be > 0");
  const int S = bt_slices(V);
  if (workspace && S >= 2 && S * k <= 1024) {
    VS_CHECK_ARG(ws_bytes >= vs_beam_topk_workspace_bytes(rows, V, k), "workspace too small");
    float2* part = (float2*)workspace;
    float* cval = (float*)((char*)workspace + (((size_t)rows * S * 8 + 127) & ~(size_t)127));
    int* cidx = (int*)(cval + (size_t)rows * S * k);
    hipLaunchKernelGGL(beam_lse_part_kernel, dim3(S, rows), dim3(256), 0, (hipStream_t)stream, logits,
                       part, V, 1.0f / temperature);
    hipLaunchKernelGGL(beam_topk_part_kernel, dim3(S, rows), dim3(256), 0, (hipStream_t)stream, logits,
                       part, cum, forced, cval, cidx, V, k, pad, eos, unk, unk_penalty,
                       1.0f / temperature, flags);
    hipLaunchKernelGGL(beam_topk_merge_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, cval, cidx,
                       out_val, out_idx, S * k, k);
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  hipLaunchKernelGGL(beam_topk_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, cum,
                     forced, out_val, out_idx, V, k, pad, eos, unk, unk_penalty, 1.0f / temperature,
                     flags);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// Token-level cross entropy with ignore_index (Simple_TxDec.forward, mdl_sf_base.py:660-664):
// per row nll = logsumexp(x) - x[label]; rows with label == ignore contribute nothing.
// out[0] = sum of nll, out[1] = number of counted rows (fixed-order single-block finish).
__global__ __launch_bounds__(256) void xent_ignore_rows_kernel(const float* logits,
                                                               const int64_t* labels, float* nll,
                                                               int V, long long ld, int ignore) {
  __shared__ float red[256];
  const int r = blockIdx.x, tid = threadIdx.x;
  const long long lb = labels[r];
  if (lb == ignore) {
    if (tid == 0) nll[r] = 0.f;
    return;
  }
  const float* x = logits + (long long)r * ld;
  float mx = -INFINITY;
  for (int j = tid; j < V; j += 256) mx = fmaxf(mx, x[j]);
  red[tid] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
    __syncthreads();
  }
  mx = red[0];
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < V; j += 256) sum += expf(x[j] - mx);
  red[tid] = sum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) nll[r] = mx + logf(red[0]) - x[lb];
}

__global__ __launch_bounds__(256) void xent_ignore_finish_kernel(const float* nll,
                                                                 const int64_t* labels, float* out,
                                                                 int rows, int ignore) {
  __shared__ double rs[256];
  __shared__ int rc[256];
  double s = 0.0;
  int c = 0;
  for (int r = threadIdx.x; r < rows; r += 256)
    if (labels[r] != ignore) {
      s += (double)nll[r];
      ++c;
    }
  rs[threadIdx.x] = s;
  rc[threadIdx.x] = c;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      rs[threadIdx.x] += rs[threadIdx.x + st];
      rc[threadIdx.x] += rc[threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = rc[0] > 0 ? (float)(rs[0] / rc[0]) : 0.f;
    out[1] = (float)rc[0];
  }
}

extern "C" int vs_xent_ignore(const float* logits, const int64_t* labels, float* nll_rows,
                              float* loss_out, int rows, int V, int64_t ld, int ignore_index,
                              void* stream) {
  VS_CHECK_ARG(logits && labels && nll_rows && loss_out && rows > 0 && V > 0 && ld >= V, "bad args");
  hipLaunchKernelGGL(xent_ignore_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits,
                     labels, nll_rows, V, (long long)ld, ignore_index);
  hipLaunchKernelGGL(xent_ignore_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, nll_rows,
                     labels, loss_out, rows, ignore_index);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// =============================================================================================
// Backward of the GPT-2 decoder (fine-tuning the language model: Simple_TxDec.forward ->
// loss.backward(), vidsitu_code/mdl_sf_base.py:653-667).  Dense gradients reuse vs_gemm_nt_f32
// (dx = dy . W with the Conv1D parameter itself as the K-contiguous operand; dW = x^T dy on
// transposed activations); the kernels below are the rest of the chain.
// =============================================================================================
__device__ __forceinline__ float gelu_new_grad_f(float x) {
  const float k = 0.7978845608028654f, c = 0.044715f;
  const float u = k * (x + c * x * x * x);
  const float t = tanhf(u);
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * k * (1.0f + 3.0f * c * x * x);
}

__global__ void gelu_new_fwd_kernel(const float* x, float* y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    y[i] = gelu_new_f(x[i]);
}
__global__ void gelu_new_bwd_kernel(const float* dy, const float* x, float* dx, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    dx[i] = dy[i] * gelu_new_grad_f(x[i]);
}
__global__ void add_f32_kernel(const float* a, const float* b, float* out, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    out[i] = a[i] + b[i];
}
static inline unsigned ew_blocks(long long n) {
  long long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  return (unsigned)(g < 1 ? 1 : g);
}
extern "C" int vs_gelu_new_fwd(const float* x, float* y, int64_t n, void* stream) {
  VS_CHECK_ARG(x && y && n > 0, "bad args");
  hipLaunchKernelGGL(gelu_new_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, y,
                     (long long)n);
  VS_CHECK_LAUNCH();
  return VS_OK;
}
extern "C" int vs_gelu_new_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
  VS_CHECK_ARG(dy && x && dx && n > 0, "bad args");
  hipLaunchKernelGGL(gelu_new_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, x,
                     dx, (long long)n);
  VS_CHECK_LAUNCH();
  return VS_OK;
}
extern "C" int vs_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
  VS_CHECK_ARG(a && b && out && n > 0, "bad args");
  hipLaunchKernelGGL(add_f32_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, out,
                     (long long)n);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// out[n] = sum_m x[m][n]  (bias gradients): a block owns 64 columns, its 16 waves take rows w, w+16, ...
// with eight rows of loads in flight, partial sums meet in LDS in wave order (bitwise reproducible).  The
// one-thread-per-column loop was a chain of M / 4 memory latencies: 46 us for 600 x 1024.
__global__ __launch_bounds__(1024) void colsum_f32_kernel(const float* x, float* out, int M, int N) {
  __shared__ float part[16][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.x * 64 + lane;
  const int nc = n < N ? n : 0;
  float s = 0.f;
  int m = wave;
  for (; m + 16 * 7 < M; m += 16 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[(long long)(m + 16 * u) * N + nc];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; m < M; m += 16) s += x[(long long)m * N + nc];
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += part[w][lane];
    out[n] = t;
  }
}
extern "C" int vs_colsum_f32(const float* x, float* out, int M, int N, void* stream) {
  VS_CHECK_ARG(x && out && M > 0 && N > 0, "bad args");
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((N + 63) / 64), dim3(1024), 0, (hipStream_t)stream, x, out, M, N);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ----------------------------------------------------------------------------
// Causal attention backward in two launches over (r, head) x chunks of AC_QC queries / keys (one block
// per (r, head) with every query and key was 160 latency-bound blocks: 177-223 us per layer).
// Launch A (query chunk, one wave per query i): p_i = softmax(s_i) with the forward's masking,
// dP_ij = dO_i . V_j, D_i = sum_j p_ij dP_ij, dS_ij = p_ij (dP_ij - D_i); dQ_i = scale * sum_{j<=i} dS_ij K_j
// (a future key's score is the constant -1e4: no gradient); the P and dS rows go to the scratch [L][L] pair.
// Launch B (key chunk, one wave per key j, lanes over d): dK_j = scale * sum_{i>=j} dS_ij Q_i,
// dV_j = sum_i p_ij dO_i, the chunk's P / dS columns staged in LDS -- every output has one owner, fixed
// summation order, no atomics.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_causal_bwd_q_kernel(const float* qkv, const uint8_t* kmask,
                                                                const float* dout, float* dqkv,
                                                                float* scratch, int L, int H, int dh) {
  extern __shared__ float sm[];
  const int D = H * dh, ld = dh + 1;
  const int r = blockIdx.x / H, h = blockIdx.x % H;
  const int i0 = blockIdx.y * AC_QC, i1 = min(L, i0 + AC_QC);
  float* Ks = sm;
  float* Vs = Ks + L * ld;
  float* Qs = Vs + L * ld;       // [AC_QC][ld] the chunk's queries
  float* Os = Qs + AC_QC * ld;   // [AC_QC][ld] the chunk's dO rows
  float* Pw = Os + AC_QC * ld;   // [4][L] per-wave p row
  float* Sw = Pw + 4 * L;        // [4][L] per-wave dS row
  const float* base = qkv + (long long)r * L * 3 * D;
  const float* dob = dout + (long long)r * L * D;
  float* P = scratch + (long long)blockIdx.x * 2 * L * L;
  float* dS = P + (long long)L * L;
  for (int i = threadIdx.x; i < L * dh; i += 256) {
    const int j = i / dh, d = i - j * dh;
    if (j < i1) Ks[j * ld + d] = base[(long long)j * 3 * D + D + h * dh + d];
    Vs[j * ld + d] = base[(long long)j * 3 * D + 2 * D + h * dh + d];
  }
  for (int i = threadIdx.x; i < (i1 - i0) * dh; i += 256) {
    const int q = i / dh, d = i - q * dh;
    Qs[q * ld + d] = base[(long long)(i0 + q) * 3 * D + h * dh + d];
    Os[q * ld + d] = dob[(long long)(i0 + q) * D + h * dh + d];
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float scale = 1.0f / sqrtf((float)dh);
  float* pw = Pw + wave * L;
  float* sw = Sw + wave * L;
  for (int i = i0 + wave; i < i1; i += 4) {
    const float* qi = Qs + (i - i0) * ld;
    const float* oi = Os + (i - i0) * ld;
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) {
      float sc = -1e4f;
      if (j <= i) {
        sc = 0.f;
        for (int d = 0; d < dh; ++d) sc += qi[d] * Ks[j * ld + d];
        sc *= scale;
      }
      if (kmask && !kmask[(long long)r * L + j]) sc += -1e4f;
      pw[j] = sc;
      mx = fmaxf(mx, sc);
    }
    mx = wave_reduce_max(mx);
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) {
      const float e = expf(pw[j] - mx);
      pw[j] = e;
      sum += e;
    }
    sum = wave_reduce_sum(sum);
    const float inv = 1.0f / sum;
    float dsum = 0.f;
    for (int j = lane; j < L; j += 64) {
      const float pj = pw[j] * inv;
      float dp = 0.f;
      for (int d = 0; d < dh; ++d) dp += oi[d] * Vs[j * ld + d];
      pw[j] = pj;
      sw[j] = dp;
      dsum += pj * dp;
    }
    dsum = wave_reduce_sum(dsum);
    for (int j = lane; j < L; j += 64) {
      const float ds = pw[j] * (sw[j] - dsum);
      sw[j] = ds;
      P[(long long)i * L + j] = pw[j];
      dS[(long long)i * L + j] = ds;
    }
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < dh; d += 64) {
      float acc = 0.f;
      for (int j = 0; j <= i; ++j) acc += sw[j] * Ks[j * ld + d];
      dqkv[((long long)r * L + i) * 3 * D + h * dh + d] = acc * scale;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(256) void attn_causal_bwd_kv_kernel(const float* qkv, const float* dout,
                                                                 float* dqkv, const float* scratch, int L,
                                                                 int H, int dh) {
  extern __shared__ float sm[];
  const int D = H * dh, ld = dh + 1;
  const int r = blockIdx.x / H, h = blockIdx.x % H;
  const int j0 = blockIdx.y * AC_QC, j1 = min(L, j0 + AC_QC);
  float* Qs = sm;                 // rows j0 .. L-1 (dK sums over i >= j)
  float* Os = Qs + L * ld;        // all rows (dV sums over every i)
  float* Pc = Os + L * ld;        // [L][AC_QC] the chunk's columns of P
  float* Sc = Pc + L * AC_QC;     // [L][AC_QC] ... of dS
  const float* base = qkv + (long long)r * L * 3 * D;
  const float* dob = dout + (long long)r * L * D;
  const float* P = scratch + (long long)blockIdx.x * 2 * L * L;
  const float* dS = P + (long long)L * L;
  for (int i = threadIdx.x; i < L * dh; i += 256) {
    const int q = i / dh, d = i - q * dh;
    if (q >= j0) Qs[q * ld + d] = base[(long long)q * 3 * D + h * dh + d];
    Os[q * ld + d] = dob[(long long)q * D + h * dh + d];
  }
  for (int i = threadIdx.x; i < L * AC_QC; i += 256) {
    const int q = i / AC_QC, c = i - q * AC_QC;
    const bool ok = j0 + c < j1;
    Pc[i] = ok ? P[(long long)q * L + j0 + c] : 0.f;
    Sc[i] = ok ? dS[(long long)q * L + j0 + c] : 0.f;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float scale = 1.0f / sqrtf((float)dh);
  for (int j = j0 + wave; j < j1; j += 4) {
    const int c = j - j0;
    for (int d = lane; d < dh; d += 64) {
      float dk = 0.f, dv = 0.f;
      for (int i = 0; i < L; ++i) dv += Pc[i * AC_QC + c] * Os[i * ld + d];
      for (int i = j; i < L; ++i) dk += Sc[i * AC_QC + c] * Qs[i * ld + d];
      dqkv[((long long)r * L + j) * 3 * D + D + h * dh + d] = dk * scale;
      dqkv[((long long)r * L + j) * 3 * D + 2 * D + h * dh + d] = dv;
    }
  }
}

extern "C" size_t vs_attn_causal_bwd_scratch_bytes(int R, int L, int H) {
  return (size_t)R * H * 2 * L * L * sizeof(float);
}

extern "C" int vs_attn_causal_bwd(const float* qkv, const uint8_t* key_mask, const float* dout,
                                  float* dqkv, void* scratch, size_t scratch_bytes, int R, int L, int H,
                                  int dh, void* stream) {
  VS_CHECK_ARG(qkv && dout && dqkv && scratch && R > 0 && L > 0 && H > 0 && dh > 0, "bad args");
  VS_CHECK_ARG(scratch_bytes >= vs_attn_causal_bwd_scratch_bytes(R, L, H), "scratch too small");
  const size_t smem_q = ((size_t)(2 * L + 2 * AC_QC) * (dh + 1) + 8 * L) * sizeof(float);
  const size_t smem_kv = ((size_t)2 * L * (dh + 1) + 2 * L * AC_QC) * sizeof(float);
  VS_CHECK_ARG(smem_q <= 160 * 1024 && smem_kv <= 160 * 1024,
               "sequence too long for the LDS-resident attention backward");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_causal_bwd_q_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attn_causal_bwd_kv_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const dim3 grid(R * H, (L + AC_QC - 1) / AC_QC);
  hipLaunchKernelGGL(attn_causal_bwd_q_kernel, grid, dim3(256), smem_q, (hipStream_t)stream, qkv, key_mask,
                     dout, dqkv, (float*)scratch, L, H, dh);
  hipLaunchKernelGGL(attn_causal_bwd_kv_kernel, grid, dim3(256), smem_kv, (hipStream_t)stream, qkv, dout, dqkv,
                     (const float*)scratch, L, H, dh);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// Embedding backward: dwte[tok] += dh, dwpe[pos0 + l] += dh (fp32 atomics: several rows may hit the
// same token / position; the sum order, hence the last bits, can differ between runs).
__global__ void gpt2_embed_bwd_kernel(const int64_t* tok, const float* dh, float* dwte, float* dwpe,
                                      int L, int D, int pos0, int V) {
  const int row = blockIdx.x;
  const int l = row % L;
  long long t = tok[row];
  if (t < 0 || t >= V) return;
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    const float g = dh[(long long)row * D + i];
    atomicAdd(dwte + t * D + i, g);
    atomicAdd(dwpe + (long long)(pos0 + l) * D + i, g);
  }
}
extern "C" int vs_gpt2_embed_bwd(const int64_t* tokens, const float* dh, float* dwte, float* dwpe, int R,
                                 int L, int D, int pos0, int V, void* stream) {
  VS_CHECK_ARG(tokens && dh && dwte && dwpe && R > 0 && L > 0, "bad args");
  hipLaunchKernelGGL(gpt2_embed_bwd_kernel, dim3(R * L), dim3(256), 0, (hipStream_t)stream, tokens, dh,
                     dwte, dwpe, L, D, pos0, V);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// dlogits of the ignore-index mean cross entropy: (softmax - onehot) * gscale / count for counted
// rows, 0 for ignored ones; count = loss_out[1] of vs_xent_ignore (device memory).
__global__ __launch_bounds__(256) void xent_ignore_grad_kernel(const float* logits,
                                                               const int64_t* labels,
                                                               const float* loss_out, float* dlogits,
                                                               int V, long long ld, int ignore,
                                                               float gscale_host, const float* gscale_dev) {
  __shared__ float red[256];
  const int r = blockIdx.x, tid = threadIdx.x;
  // the upstream gradient of the scalar loss: a host number, or (graph-safe, no host sync) read from the device
  const float gscale = gscale_dev ? gscale_host * gscale_dev[0] : gscale_host;
  const long long lb = labels[r];
  float* dst = dlogits + (long long)r * ld;
  if (lb == ignore) {
    for (int j = tid; j < V; j += 256) dst[j] = 0.f;
    return;
  }
  const float* x = logits + (long long)r * ld;
  float mx = -INFINITY;
  for (int j = tid; j < V; j += 256) mx = fmaxf(mx, x[j]);
  red[tid] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
    __syncthreads();
  }
  mx = red[0];
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < V; j += 256) sum += expf(x[j] - mx);
  red[tid] = sum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const float inv = gscale / (red[0] * fmaxf(loss_out[1], 1.0f));
  const float onehot = gscale / fmaxf(loss_out[1], 1.0f);
  for (int j = tid; j < V; j += 256) dst[j] = expf(x[j] - mx) * inv - (j == lb ? onehot : 0.f);
}
extern "C" int vs_xent_ignore_grad(const float* logits, const int64_t* labels, const float* loss_out,
                                   float* dlogits, int rows, int V, int64_t ld, int ignore_index,
                                   float grad_scale, void* stream) {
  VS_CHECK_ARG(logits && labels && loss_out && dlogits && rows > 0 && V > 0 && ld >= V, "bad args");
  hipLaunchKernelGGL(xent_ignore_grad_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits,
                     labels, loss_out, dlogits, V, (long long)ld, ignore_index, grad_scale, (const float*)nullptr);
  VS_CHECK_LAUNCH();
  return VS_OK;
}
extern "C" int vs_xent_ignore_grad_dev(const float* logits, const int64_t* labels, const float* loss_out,
                                       float* dlogits, int rows, int V, int64_t ld, int ignore_index,
                                       const float* grad_scale_dev, void* stream) {
  VS_CHECK_ARG(logits && labels && loss_out && dlogits && grad_scale_dev && rows > 0 && V > 0 && ld >= V, "bad args");
  hipLaunchKernelGGL(xent_ignore_grad_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits,
                     labels, loss_out, dlogits, V, (long long)ld, ignore_index, 1.0f, grad_scale_dev);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// =============================================================================================
// Device-side beam-search bookkeeping: everything SeqGenCustom._generate does
// on the host between two decoder calls (seq_gen.py:368-520 + finalize_hypos :579-697) for ONE
// step, one block per sentence, no host synchronisation.  Finished sentences are not removed from
// the batch (the reference shrinks it); their rows idle, results are unchanged.
// =============================================================================================
struct BeamStepP {
  const float* row_val;     // [bsz*beam][k]   per-row top-k of vs_beam_topk (cumulative scores added)
  const int64_t* row_idx;   // [bsz*beam][k]
  const int64_t* tok_in;    // [bsz*beam][Lt]  Lt = max_len + 2
  int64_t* tok_out;
  const float* sc_in;       // [bsz*beam][Ls]  Ls = max_len + 1
  float* sc_out;
  uint8_t* ignore;          // [bsz][beam]     cands_to_ignore
  uint8_t* finished;        // [bsz]
  int* nfin;                // [bsz]           hypotheses finalized so far
  int* remaining;           // [1]             sentences not finished yet
  int64_t* fin_tok;         // [bsz][beam][Ls] finalized token sequences (eos included)
  float* fin_score;         // [bsz][beam]
  float* fin_pos;           // [bsz][beam][Ls] positional scores
  int* fin_len;             // [bsz][beam]
  int64_t* reorder;         // [bsz*beam]      parent row of every new row (KV-cache gather index)
  const int* anc_in;        // [bsz*beam][anc_ld] or NULL: cache row holding position j of the hypothesis
  int* anc_out;             //                  (vs_attn_decode's ancestry table; replaces the gather)
  int anc_ld;
  int beam, k, V, step, max_len, Lt, Ls, eos, normalize;
  float len_penalty;
};

#define BS_MAXC 64  // 2 * beam candidates kept per sentence (beam <= 32)

__device__ __forceinline__ void bs_anc_row(const BeamStepP& p, long long nrow, long long prow, int lane) {
  if (!p.anc_out) return;
  for (int j = lane; j <= p.step + 1 && j < p.anc_ld; j += 64)
    p.anc_out[nrow * p.anc_ld + j] = j <= p.step ? p.anc_in[prow * p.anc_ld + j] : (int)nrow;
}

__global__ __launch_bounds__(64) void beam_step_kernel(BeamStepP p) {
  __shared__ float cv[BS_MAXC];      // merged candidates: value
  __shared__ int ct[BS_MAXC];        //                    token
  __shared__ int cb[BS_MAXC];        //                    beam (parent, relative)
  __shared__ int act[BS_MAXC];       // active hypotheses (candidate index) of the next step
  __shared__ float bestv[64];
  __shared__ long long bestf[64];
  __shared__ int besti[64];
  __shared__ int sh_new_fin;
  const int s = blockIdx.x, lane = threadIdx.x;
  const int beam = p.beam, k = p.k, csz = 2 * beam;
  const long long row0 = (long long)s * beam;
  if (p.finished[s]) {  // idle rows: carry the state over unchanged
    for (int b = 0; b < beam; ++b) {
      for (int j = lane; j < p.Lt; j += 64) p.tok_out[(row0 + b) * p.Lt + j] = p.tok_in[(row0 + b) * p.Lt + j];
      for (int j = lane; j < p.Ls; j += 64) p.sc_out[(row0 + b) * p.Ls + j] = p.sc_in[(row0 + b) * p.Ls + j];
      if (lane == 0) p.reorder[row0 + b] = row0 + b;
      bs_anc_row(p, row0 + b, row0 + b, lane);
    }
    return;
  }
  // ---- fairseq BeamSearch.step on the per-row lists: the csz best of beam*k entries, by value
  //      descending, ties towards the lowest flattened index beam*V + token (step 0: first beam only)
  const int nb = p.step == 0 ? 1 : beam;
  const int total = nb * k;
  float pv = INFINITY;
  long long pf = -1;
  for (int c = 0; c < csz; ++c) {
    float bv = -INFINITY;
    long long bf = 0x7fffffffffffffffll;
    int bi = -1;
    if (c < k) {
      for (int e = lane; e < total; e += 64) {
        const float v = p.row_val[row0 * k + e];
        const long long f = (long long)(e / k) * p.V + p.row_idx[row0 * k + e];
        const bool after = (v < pv) || (v == pv && f > pf);
        if (after && (v > bv || (v == bv && f < bf))) { bv = v; bf = f; bi = e; }
      }
    }
    bestv[lane] = bv; bestf[lane] = bf; besti[lane] = bi;
    __syncthreads();
    for (int st = 32; st > 0; st >>= 1) {
      if (lane < st) {
        const float v2 = bestv[lane + st]; const long long f2 = bestf[lane + st];
        if (besti[lane + st] >= 0 && (besti[lane] < 0 || v2 > bestv[lane] || (v2 == bestv[lane] && f2 < bestf[lane]))) {
          bestv[lane] = v2; bestf[lane] = f2; besti[lane] = besti[lane + st];
        }
      }
      __syncthreads();
    }
    if (lane == 0) {
      if (besti[0] >= 0) {
        cv[c] = bestv[0]; ct[c] = (int)p.row_idx[row0 * k + besti[0]]; cb[c] = besti[0] / k;
      } else {  // fewer than 2*beam candidates exist (tiny vocabularies): -inf padding
        cv[c] = -INFINITY; ct[c] = 0; cb[c] = 0;
      }
    }
    pv = bestv[0]; pf = bestf[0];
    __syncthreads();
  }
  // ---- finalize hypotheses that end in eos among the first `beam` candidates (lane 0: in order)
  if (lane == 0) {
    int seen = 0;
    for (int c = 0; c < beam; ++c) {
      const bool is_eos = ct[c] == p.eos && cv[c] != -INFINITY && !p.ignore[s * beam + c];
      if (!is_eos) continue;
      seen = 1;
      if (p.nfin[s] < beam) {
        const int h = p.nfin[s]++;
        const long long prow = row0 + cb[c];
        int64_t* ft = p.fin_tok + ((long long)s * beam + h) * p.Ls;
        float* fp = p.fin_pos + ((long long)s * beam + h) * p.Ls;
        for (int j = 0; j < p.step; ++j) ft[j] = p.tok_in[prow * p.Lt + 1 + j];
        ft[p.step] = p.eos;
        float prev = 0.f;
        for (int j = 0; j < p.step; ++j) {
          const float cur = p.sc_in[prow * p.Ls + j];
          fp[j] = j == 0 ? cur : cur - prev;
          prev = cur;
        }
        fp[p.step] = p.step == 0 ? cv[c] : cv[c] - prev;
        float sc = cv[c];
        if (p.normalize) sc = sc / powf((float)(p.step + 1), p.len_penalty);
        p.fin_score[s * beam + h] = sc;
        p.fin_len[s * beam + h] = p.step + 1;
      }
    }
    int fin_now = 0;
    if (seen && (p.nfin[s] == beam || p.step == p.max_len)) {
      p.finished[s] = 1;
      atomicSub(p.remaining, 1);
      fin_now = 1;
    }
    sh_new_fin = fin_now;
  }
  __syncthreads();
  if (sh_new_fin) {
    for (int b = 0; b < beam; ++b) {
      for (int j = lane; j < p.Lt; j += 64) p.tok_out[(row0 + b) * p.Lt + j] = p.tok_in[(row0 + b) * p.Lt + j];
      for (int j = lane; j < p.Ls; j += 64) p.sc_out[(row0 + b) * p.Ls + j] = p.sc_in[(row0 + b) * p.Ls + j];
      if (lane == 0) p.reorder[row0 + b] = row0 + b;
      bs_anc_row(p, row0 + b, row0 + b, lane);
    }
    return;
  }
  // ---- the `beam` live candidates with the smallest rank (eos / ignored ones sort behind)
  if (lane == 0) {
    int n = 0;
    for (int pass = 0; pass < 2 && n < beam; ++pass)
      for (int c = 0; c < csz && n < beam; ++c) {
        bool dead = ct[c] == p.eos && cv[c] != -INFINITY;
        if (c < beam) dead = dead || p.ignore[s * beam + c];
        if ((pass == 0) == !dead) act[n++] = c | (dead ? 0x10000 : 0);
      }
    for (int b = 0; b < beam; ++b) p.ignore[s * beam + b] = (act[b] & 0x10000) ? 1 : 0;
  }
  __syncthreads();
  for (int b = 0; b < beam; ++b) {
    const int c = act[b] & 0xffff;
    const long long prow = row0 + cb[c], nrow = row0 + b;
    for (int j = lane; j < p.Lt; j += 64) {
      int64_t t = p.tok_in[prow * p.Lt + j];
      if (j == p.step + 1) t = ct[c];
      p.tok_out[nrow * p.Lt + j] = t;
    }
    for (int j = lane; j < p.Ls; j += 64) {
      float v = p.sc_in[prow * p.Ls + j];
      if (j == p.step) v = cv[c];
      p.sc_out[nrow * p.Ls + j] = v;
    }
    if (lane == 0) p.reorder[nrow] = prow;
    bs_anc_row(p, nrow, prow, lane);
  }
}

extern "C" int vs_beam_step(const float* row_val, const int64_t* row_idx, const int64_t* tok_in,
                            int64_t* tok_out, const float* sc_in, float* sc_out, uint8_t* ignore,
                            uint8_t* finished, int* nfin, int* remaining, int64_t* fin_tok,
                            float* fin_score, float* fin_pos, int* fin_len, int64_t* reorder,
                            const int32_t* anc_in, int32_t* anc_out, int anc_ld, int bsz, int beam, int k,
                            int V, int step, int max_len, int eos, int normalize, float len_penalty,
                            void* stream) {
  VS_CHECK_ARG(row_val && row_idx && tok_in && tok_out && sc_in && sc_out && ignore && finished && nfin &&
                   remaining && fin_tok && fin_score && fin_pos && fin_len && reorder, "null argument");
  VS_CHECK_ARG(bsz > 0 && beam >= 1 && 2 * beam <= BS_MAXC && k >= 1 && k <= 2 * beam, "beam <= 32, k <= 2*beam");
  VS_CHECK_ARG((anc_in == nullptr) == (anc_out == nullptr) && (!anc_out || anc_ld > step), "ancestry tables");
  BeamStepP p;
  p.anc_in = anc_in; p.anc_out = anc_out; p.anc_ld = anc_ld;
  p.row_val = row_val; p.row_idx = row_idx; p.tok_in = tok_in; p.tok_out = tok_out;
  p.sc_in = sc_in; p.sc_out = sc_out; p.ignore = ignore; p.finished = finished; p.nfin = nfin;
  p.remaining = remaining; p.fin_tok = fin_tok; p.fin_score = fin_score; p.fin_pos = fin_pos;
  p.fin_len = fin_len; p.reorder = reorder;
  p.beam = beam; p.k = k; p.V = V; p.step = step; p.max_len = max_len;
  p.Lt = max_len + 2; p.Ls = max_len + 1; p.eos = eos; p.normalize = normalize;
  p.len_penalty = len_penalty;
  hipLaunchKernelGGL(beam_step_kernel, dim3(bsz), dim3(64), 0, (hipStream_t)stream, p);
  VS_CHECK_LAUNCH();
  return VS_OK;
}


// =============================================================================================
// fairseq TransformerDecoder pieces (TxDecoderReal, vidsitu_code/mdl_sf_base.py:435-446):
// x = embed_scale * embed_tokens(tok) + positions (fairseq/models/transformer.py extract_features)
// with the sinusoidal rows gathered from a small table through a per-token index (row 0 = zeros for
// padding), the scatter of its gradient (padding row left at zero, nn.Embedding(padding_idx)), and
// the relu gradient of the FFN.
// =============================================================================================
__global__ void embed_pos_fwd_kernel(const int64_t* tokens, const float* emb, const float* pos_table,
                                     const int64_t* pos_idx, float* out, long long n_tok, int D, float scale) {
  const long long t = blockIdx.x;
  const float4* e = (const float4*)(emb + tokens[t] * D);
  const float4* p = (const float4*)(pos_table + pos_idx[t] * D);
  float4* o = (float4*)(out + t * D);
  for (int i = threadIdx.x; i < D / 4; i += blockDim.x) {
    const float4 a = e[i], b = p[i];
    o[i] = make_float4(scale * a.x + b.x, scale * a.y + b.y, scale * a.z + b.z, scale * a.w + b.w);
  }
}

extern "C" int vs_embed_pos_fwd(const int64_t* tokens, const float* emb, const float* pos_table,
                                const int64_t* pos_idx, float* out, int64_t n_tok, int D, float scale,
                                void* stream) {
  VS_CHECK_ARG(tokens && emb && pos_table && pos_idx && out && n_tok > 0 && D > 0 && (D & 3) == 0, "bad args");
  hipLaunchKernelGGL(embed_pos_fwd_kernel, dim3((unsigned)n_tok), dim3(256), 0, (hipStream_t)stream, tokens, emb,
                     pos_table, pos_idx, out, (long long)n_tok, D, scale);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

__global__ void embed_scatter_bwd_kernel(const int64_t* tokens, const float* dx, float* demb, int D, float scale,
                                         long long pad) {
  const long long t = blockIdx.x;
  const long long tok = tokens[t];
  if (tok == pad) return;
  for (int i = threadIdx.x; i < D; i += blockDim.x) atomicAdd(demb + tok * D + i, scale * dx[t * D + i]);
}

/* demb (zero-filled by the caller) += scale * dx for every non-padding token. */
extern "C" int vs_embed_scatter_bwd(const int64_t* tokens, const float* dx, float* demb, int64_t n_tok, int D,
                                    float scale, int64_t pad, void* stream) {
  VS_CHECK_ARG(tokens && dx && demb && n_tok > 0 && D > 0, "bad args");
  hipLaunchKernelGGL(embed_scatter_bwd_kernel, dim3((unsigned)n_tok), dim3(256), 0, (hipStream_t)stream, tokens,
                     dx, demb, D, scale, (long long)pad);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

__global__ void relu_bwd_kernel(const float* dy, const float* y, float* dx, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

extern "C" int vs_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
  VS_CHECK_ARG(dy && y && dx && n > 0, "bad args");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx,
                     (long long)n);
  VS_CHECK_LAUNCH();
  return VS_OK;
}
