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

template <int BM, int BN>  // 128x128 (2x2 MFMA tiles per wave) or 64x64 (one per wave)
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          const float* res, float* y, int M, int N,
                                                          int K, int act) {
  constexpr int MR = BM / 64, NR = BN / 64;  // 32x32 accumulators per wave
  constexpr int PA = BM / 32, PB = BN / 32;  // loader passes (32 rows each)
  __shared__ float As[2][BM * GM_LD];
  __shared__ float Bs[2][BN * GM_LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  // loader mapping: 8 float4 per 32-float row, 32 rows per pass
  const int lc = tid & 7, lr = tid >> 3;
  const bool kvec = (K & 3) == 0;
  float4 ra[PA], rb[PB];

  auto ld4 = [&](const float* base, int row, int rows, int k) __attribute__((always_inline)) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows && k < K) {
      const float* s = base + (long long)row * K + k;
      if (kvec) v = *(const float4*)s;
      else { v.x = s[0]; if (k + 1 < K) v.y = s[1]; if (k + 2 < K) v.z = s[2]; if (k + 3 < K) v.w = s[3]; }
    }
    return v;
  };
  auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PA; ++i) ra[i] = ld4(x, m0 + lr + 32 * i, M, k0 + lc * 4);
#pragma unroll
    for (int i = 0; i < PB; ++i) rb[i] = ld4(w, n0 + lr + 32 * i, N, k0 + lc * 4);
  };
  auto sstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      float* a = &As[buf][(lr + 32 * i) * GM_LD + lc * 4];
      a[0] = ra[i].x; a[1] = ra[i].y; a[2] = ra[i].z; a[3] = ra[i].w;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      float* b = &Bs[buf][(lr + 32 * i) * GM_LD + lc * 4];
      b[0] = rb[i].x; b[1] = rb[i].y; b[2] = rb[i].z; b[3] = rb[i].w;
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
  const int nk = (K + GM_BK - 1) / GM_BK;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
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

int vs_gemm_nt_f32_mfma(const float* x, const float* w, const float* b, const float* res, float* y,
                        int M, int N, int K, int act, hipStream_t st) {
  // fewer than one 128x128 tile per CU: 64x64 tiles (4x the blocks) keep the chip busy
  const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128);
  if (t128 >= 256)
    hipLaunchKernelGGL((gemm_nt_f32_kernel<128, 128>), dim3((N + 127) / 128, (M + 127) / 128), dim3(256),
                       0, st, x, w, b, res, y, M, N, K, act);
  else
    hipLaunchKernelGGL((gemm_nt_f32_kernel<64, 64>), dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0,
                       st, x, w, b, res, y, M, N, K, act);
  VS_CHECK_LAUNCH();
  return VS_OK;
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
__global__ __launch_bounds__(256) void attn_causal_kernel(const float* qkv, const uint8_t* kmask,
                                                          float* out, int L, int H, int dh) {
  extern __shared__ float sm[];
  const int D = H * dh, ld = dh + 1;
  const int r = blockIdx.x / H, h = blockIdx.x % H;
  float* Ks = sm;
  float* Vs = Ks + L * ld;
  float* Ps = Vs + L * ld;   // [4][L]
  float* Qs = Ps + 4 * L;    // [4][dh]
  const float* base = qkv + (long long)r * L * 3 * D;
  for (int i = threadIdx.x; i < L * dh; i += 256) {
    const int j = i / dh, d = i - j * dh;
    Ks[j * ld + d] = base[(long long)j * 3 * D + D + h * dh + d];
    Vs[j * ld + d] = base[(long long)j * 3 * D + 2 * D + h * dh + d];
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float scale = 1.0f / sqrtf((float)dh);
  float* P = Ps + wave * L;
  float* Q = Qs + wave * dh;
  for (int i = wave; i < L; i += 4) {
    for (int d = lane; d < dh; d += 64) Q[d] = base[(long long)i * 3 * D + h * dh + d];
    __builtin_amdgcn_wave_barrier();
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) {
      float s = 0.f;
      for (int d = 0; d < dh; ++d) s += Q[d] * Ks[j * ld + d];
      s *= scale;
      if (j > i) s = -1e4f;
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
  hipLaunchKernelGGL(attn_causal_kernel, dim3(R * H), dim3(256), smem, (hipStream_t)stream, qkv,
                     key_mask, out, L, H, dh);
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

extern "C" int vs_attn_decode(const float* qkv, float* kcache, float* vcache, const uint8_t* key_mask,
                              float* out, int rows, int H, int dh, int Lmax, int t, void* stream) {
  VS_CHECK_ARG(qkv && kcache && vcache && out && rows > 0 && t >= 0 && t < Lmax, "bad args");
  const size_t smem = (size_t)(t + 1 + dh) * sizeof(float);
  VS_CHECK_ARG(smem <= 64 * 1024, "cache too long");
  hipLaunchKernelGGL(attn_decode_kernel, dim3(rows * H), dim3(64), smem, (hipStream_t)stream, qkv,
                     kcache, vcache, key_mask, out, H, dh, Lmax, t);
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

extern "C" int vs_beam_topk(const float* logits, const float* cum, const int64_t* forced,
                            float* out_val, int64_t* out_idx, int rows, int V, int k, int pad,
                            int eos, int unk, float unk_penalty, float temperature, int flags,
                            void* stream) {
  VS_CHECK_ARG(logits && out_val && out_idx && rows > 0 && V > 1, "bad args");
  VS_CHECK_ARG(k >= 1 && k <= BT_MAXK && k < V, "k must be in [1, 32] and < V");
  VS_CHECK_ARG(temperature > 0.f, "temperature must be > 0");
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

// out[n] = sum_m x[m][n]  (bias gradients): one thread per column, fixed row order.
__global__ void colsum_f32_kernel(const float* x, float* out, int M, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  int m = 0;
  for (; m + 4 <= M; m += 4) {
    const float a = x[(long long)m * N + n], b = x[(long long)(m + 1) * N + n],
                c = x[(long long)(m + 2) * N + n], d = x[(long long)(m + 3) * N + n];
    s += a; s += b; s += c; s += d;
  }
  for (; m < M; ++m) s += x[(long long)m * N + n];
  out[n] = s;
}
extern "C" int vs_colsum_f32(const float* x, float* out, int M, int N, void* stream) {
  VS_CHECK_ARG(x && out && M > 0 && N > 0, "bad args");
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, x,
                     out, M, N);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ----------------------------------------------------------------------------
// Causal attention backward.  One block per (r, head): Q, K, V, dO of the head in LDS
// (pitch dh+1).  Pass A (one wave per query i): p_i = softmax(s_i) with the forward's masking,
// dP_ij = dO_i . V_j, D_i = sum_j p_ij dP_ij, dS_ij = p_ij (dP_ij - D_i); dQ_i = scale * dS_i K;
// P and dS rows go to a scratch [L][L] pair.  Pass B (one wave per key j, lanes over d):
// dK_j = scale * sum_i dS_ij Q_i, dV_j = sum_i p_ij dO_i -- every output has one owner, fixed
// summation order, no atomics.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_causal_bwd_kernel(const float* qkv, const uint8_t* kmask,
                                                              const float* dout, float* dqkv,
                                                              float* scratch, int L, int H, int dh) {
  extern __shared__ float sm[];
  const int D = H * dh, ld = dh + 1;
  const int r = blockIdx.x / H, h = blockIdx.x % H;
  float* Qs = sm;
  float* Ks = Qs + L * ld;
  float* Vs = Ks + L * ld;
  float* Os = Vs + L * ld;   // dO
  float* Pw = Os + L * ld;   // [4][L] per-wave p row
  float* Sw = Pw + 4 * L;    // [4][L] per-wave dS row
  const float* base = qkv + (long long)r * L * 3 * D;
  const float* dob = dout + (long long)r * L * D;
  float* P = scratch + (long long)blockIdx.x * 2 * L * L;
  float* dS = P + (long long)L * L;
  for (int i = threadIdx.x; i < L * dh; i += 256) {
    const int j = i / dh, d = i - j * dh;
    Qs[j * ld + d] = base[(long long)j * 3 * D + h * dh + d];
    Ks[j * ld + d] = base[(long long)j * 3 * D + D + h * dh + d];
    Vs[j * ld + d] = base[(long long)j * 3 * D + 2 * D + h * dh + d];
    Os[j * ld + d] = dob[(long long)j * D + h * dh + d];
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float scale = 1.0f / sqrtf((float)dh);
  float* pw = Pw + wave * L;
  float* sw = Sw + wave * L;
  for (int i = wave; i < L; i += 4) {
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) {
      float s = 0.f;
      for (int d = 0; d < dh; ++d) s += Qs[i * ld + d] * Ks[j * ld + d];
      s *= scale;
      if (j > i) s = -1e4f;
      if (kmask && !kmask[(long long)r * L + j]) s += -1e4f;
      pw[j] = s;
      mx = fmaxf(mx, s);
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
      const float p = pw[j] * inv;
      float dp = 0.f;
      for (int d = 0; d < dh; ++d) dp += Os[i * ld + d] * Vs[j * ld + d];
      pw[j] = p;
      sw[j] = dp;
      dsum += p * dp;
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
      float a = 0.f;
      for (int j = 0; j < L; ++j) a += sw[j] * Ks[j * ld + d];
      dqkv[((long long)r * L + i) * 3 * D + h * dh + d] = a * scale;
    }
    __builtin_amdgcn_wave_barrier();
  }
  __threadfence_block();
  __syncthreads();
  for (int j = wave; j < L; j += 4) {
    for (int d = lane; d < dh; d += 64) {
      float dk = 0.f, dv = 0.f;
      for (int i = 0; i < L; ++i) {
        dk += dS[(long long)i * L + j] * Qs[i * ld + d];
        dv += P[(long long)i * L + j] * Os[i * ld + d];
      }
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
  const size_t smem = ((size_t)4 * L * (dh + 1) + 8 * L) * sizeof(float);
  VS_CHECK_ARG(smem <= 160 * 1024, "sequence too long for the LDS-resident attention backward");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_causal_bwd_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(attn_causal_bwd_kernel, dim3(R * H), dim3(256), smem, (hipStream_t)stream, qkv,
                     key_mask, dout, dqkv, (float*)scratch, L, H, dh);
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
                                                               float gscale) {
  __shared__ float red[256];
  const int r = blockIdx.x, tid = threadIdx.x;
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
                     labels, loss_out, dlogits, V, (long long)ld, ignore_index, grad_scale);
  VS_CHECK_LAUNCH();
  return VS_OK;
}
