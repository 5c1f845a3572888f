// fp32 kernels for the 5-token transformer encoder and the small heads.
//
// Replaces (reference call sites): nn.Linear of utils/transformer_code.py:51-79
// (wq/wk/wv/wo, linear1/linear2), the per-head softmax attention :33-48, the
// post-LN residual block :21-30, proj_head / vid_feat_encoder of
// vidsitu_code/mdl_sf_base.py:161-167,767-769, F.cross_entropy :226-231 and the
// softmax->sort->top-5 of vidsitu_code/evl_vsitu.py:39-42.
//
// Everything here is weight-bandwidth / latency bound (M = 5*B <= 64 rows), so
// it stays in exact fp32 on the VALU: each weight row is streamed once per
// 64-row slab, activations are staged in LDS, reductions are wave shuffles.
#include "common.h"

#include <mutex>

// ----------------------------------------------------------------------------
// y[M,N] = act(x[M,K] @ W[N,K]^T + b).   One wave per output column; lanes split
// K in float4 units; x chunk [MT][256] staged in LDS and shared by the 4 waves.
// grid = (ceil(N/4), ceil(M/MT))
// ----------------------------------------------------------------------------
// One output column per wave (N/4 blocks keep every CU busy); the wave issues the weight loads
// of KU = 4 consecutive 256-wide K chunks up front, so 4 KiB per wave / 16 KiB per CU are in
// flight while x chunks are staged through LDS (one chunk in flight left it latency bound).
template <int MT>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ b,
                                                         const float* res, float* y, int M, int N,
                                                         int K, int act) {
  constexpr int KU = 4;
  __shared__ float4 xs[MT][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + wave;
  const int mbase = blockIdx.y * MT;
  float acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = 0.f;
  const bool vec_ok = (K & 3) == 0;
  for (int kg = 0; kg < K; kg += 256 * KU) {
    float4 wv[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int k = kg + u * 256 + lane * 4;
      wv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N && k < K) {
        const float* src = w + (long long)n * K + k;
        if (vec_ok && k + 3 < K) {
          wv[u] = *(const float4*)src;
        } else {
          wv[u].x = src[0];
          if (k + 1 < K) wv[u].y = src[1];
          if (k + 2 < K) wv[u].z = src[2];
          if (k + 3 < K) wv[u].w = src[3];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int k0 = kg + u * 256;
      if (k0 >= K) break;
      __syncthreads();
      for (int i = threadIdx.x; i < MT * 64; i += 256) {
        const int m = i >> 6, l = i & 63;
        const int kk = k0 + l * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mbase + m < M && kk < K) {
          const float* src = x + (long long)(mbase + m) * K + kk;
          if (vec_ok && kk + 3 < K) {
            v = *(const float4*)src;
          } else {
            v.x = src[0];
            if (kk + 1 < K) v.y = src[1];
            if (kk + 2 < K) v.z = src[2];
            if (kk + 3 < K) v.w = src[3];
          }
        }
        xs[m][l] = v;
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float4 xv = xs[m][lane];
        acc[m] += xv.x * wv[u].x + xv.y * wv[u].y + xv.z * wv[u].z + xv.w * wv[u].w;
      }
    }
  }
  if (n < N) {
    const float bias = b ? b[n] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] += __shfl_xor(acc[m], o, 64);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float s = acc[m];
      if (lane == 0 && mbase + m < M) {
        s += bias;
        if (act == 1) s = fmaxf(s, 0.f);
        else if (act == 2)  // transformers gelu_new
          s = 0.5f * s * (1.0f + tanhf(0.7978845608028654f * (s + 0.044715f * s * s * s)));
        if (res) s += res[(long long)(mbase + m) * N + n];
        y[(long long)(mbase + m) * N + n] = s;
      }
    }
  }
}

// Few rows (M <= 16) and K <= 4096 / MT floats: the whole x lives in LDS (one staging pass, one
// barrier) and every weight chunk of the wave's row is requested before anything is waited for.
// The chunked kernel above pays a global-load latency plus two barriers per 256-wide K chunk,
// which at M = 8 (TxEncoder tokens of one rank) left each launch at ~10 us for 4 MB of weights.
// (xmask: the rows of x are multiplied by (xmask > 0) while they are staged -- the ReLU backward of the layer whose
//  output gradient x is, fused into the data-gradient product; bid: the block's index within this product)
// PRE (K <= 1024 only): the rows of x are already in LDS, row-major with pitch K at the start of the dynamic area -- put
// there by the prologue of a fused kernel (LayerNorm in front of its consumer, below); the staging loads are skipped,
// everything else -- chunk order, FMA chains, the cross-lane sums -- is the same code, hence the same bits.
template <int MT, int KC, bool PRE = false>  // KC = number of 256-wide K chunks held in registers (K <= 256 * KC)
__device__ __forceinline__ void linear_fullx_body(const float* __restrict__ x, const float* __restrict__ xmask,
                                                  const float* __restrict__ w, const float* __restrict__ b,
                                                  const float* res, float* y, int M, int N, int K, int act,
                                                  int bid) {
  extern __shared__ float4 xs4[];  // [MT][min(K, 1024) / 4]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = bid * 4 + wave;
  const int K4 = K >> 2;
  float4 wv[KC];
#pragma unroll
  for (int u = 0; u < KC; ++u) {
    const int k4 = u * 64 + lane;
    const bool ok = n < N && k4 < K4;  // branch-free (see linear_rows16_kernel)
    // (a weight row is read once per launch and the matrix once per pass: a non-temporal load -- the encoder's 600 MB of
    //  weight traffic per step otherwise flushes the trunk's last activations out of the Infinity Cache right before
    //  the backward pass wants them)
    typedef __attribute__((ext_vector_type(4))) float nt_f4;
    const nt_f4 vt = __builtin_nontemporal_load((const nt_f4*)(w + (ok ? (long long)n * K + k4 * 4 : 0)));
    const float4 v = make_float4(vt[0], vt[1], vt[2], vt[3]);
    wv[u] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // x goes through LDS 1024 columns (one float4 column per thread) at a time: thread t owns column c * 256 + t of
  // chunk c, all MT loads of it in flight at once.  (Staging all of a 3072-wide x -- the q|k|v projection's data
  // gradient -- took 96 KB: one block per CU for the whole launch, its horizontally fused weight-gradient blocks
  // included: 22.6 us against 9.)  The weights of every chunk are requested up front; the sums run over the
  // chunks in order, i.e. in the same order as with x staged whole.
  constexpr int NCH = (KC + 3) / 4;
  static_assert(!PRE || NCH == 1, "pre-staged x: one chunk");
  const int CW4 = NCH == 1 ? K4 : 256;  // row pitch of the staged chunk, in float4s
  float acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = 0.f;
#pragma unroll
  for (int h = 0; h < NCH; ++h) {
    const int k4 = h * 256 + threadIdx.x;
    if constexpr (!PRE) {
      float4 t[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const bool ok = m < M && k4 < K4;
        float4 v = *(const float4*)(x + (ok ? (long long)m * K + k4 * 4 : 0));
        if (xmask) {
          const float4 mk = *(const float4*)(xmask + (ok ? (long long)m * K + k4 * 4 : 0));
          v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
          v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
        }
        t[m] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (h) __syncthreads();  // the previous chunk has been consumed
      if (k4 < K4) {
#pragma unroll
        for (int m = 0; m < MT; ++m) xs4[m * CW4 + threadIdx.x] = t[m];
      }
    }
    __syncthreads();
#pragma unroll
    for (int uu = 0; uu < 4; ++uu) {
      const int u = h * 4 + uu;
      if (u < KC) {
        const int k4u = u * 64 + lane;
        if (k4u < K4) {
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float4 xv = xs4[m * CW4 + uu * 64 + lane];
            // explicit fused multiply-adds: left to the compiler's contraction the four products were fused
            // differently in different kernels that inline this body (stand-alone launch vs the encoder stack)
            float a = acc[m];
            a = __fmaf_rn(xv.x, wv[u].x, a);
            a = __fmaf_rn(xv.y, wv[u].y, a);
            a = __fmaf_rn(xv.z, wv[u].z, a);
            acc[m] = __fmaf_rn(xv.w, wv[u].w, a);
          }
        }
      }
    }
  }
  // cross-lane sums through LDS (see linear_rows16_kernel): lane m < MT finishes output (m, n)
  __syncthreads();
  float* red = (float*)xs4;
#pragma unroll
  for (int m = 0; m < MT; ++m) red[(wave * MT + m) * 65 + lane] = acc[m];
  __syncthreads();
  if (lane < MT && lane < M && n < N) {
    const float* row = red + (wave * MT + lane) * 65;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 64; ++j) s += row[j];
    s += b ? b[n] : 0.f;
    if (act == 1) s = fmaxf(s, 0.f);
    else if (act == 2) s = 0.5f * s * (1.0f + tanhf(0.7978845608028654f * (s + 0.044715f * s * s * s)));
    if (res) s += res[(long long)lane * N + n];
    y[(long long)lane * N + n] = s;
  }
}

template <int MT, int KC>
__global__ __launch_bounds__(256) void linear_fullx_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ b,
                                                           const float* res, float* y, int M, int N,
                                                           int K, int act) {
  linear_fullx_body<MT, KC>(x, nullptr, w, b, res, y, M, N, K, act, blockIdx.x);
}

template <int MT, int KC>
static void launch_fullx(const float* x, const float* w, const float* b, const float* res, float* y,
                         int M, int N, int K, int act, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)linear_fullx_kernel<MT, KC>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  size_t smem = (size_t)MT * (K < 1024 ? K : 1024) * 4;  // one 1024-column chunk of x
  if (smem < (size_t)4 * MT * 65 * 4) smem = (size_t)4 * MT * 65 * 4;
  hipLaunchKernelGGL((linear_fullx_kernel<MT, KC>), dim3((N + 3) / 4), dim3(256), smem, st, x, w, b,
                     res, y, M, N, K, act);
}

// 17..64 rows (GPT-2 decode steps: batch x beam rows): 16-row x 16-column blocks, four columns per
// wave so that one LDS read of x feeds 16 FMAs (the one-column kernel is LDS-read bound above
// ~16 rows: 123 GB/s of weights at M = 50), x staged 1024 floats of K at a time (64 KB, two blocks per CU).
#define LR_ROWS 16
#define LR_CPW 4
#define LR_KCH 1024

__global__ __launch_bounds__(256) void linear_rows16_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ b,
                                                            const float* res, float* y, int M, int N,
                                                            int K, int act) {
  extern __shared__ float4 xs4[];  // [LR_ROWS][LR_KCH / 4]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = (blockIdx.x * 4 + wave) * LR_CPW;
  const int m0 = blockIdx.y * LR_ROWS;
  float acc[LR_ROWS][LR_CPW];
#pragma unroll
  for (int m = 0; m < LR_ROWS; ++m)
#pragma unroll
    for (int c = 0; c < LR_CPW; ++c) acc[m][c] = 0.f;
  for (int kc = 0; kc < K; kc += LR_KCH) {
    float4 wv[LR_CPW][4];
#pragma unroll
    for (int c = 0; c < LR_CPW; ++c)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        // branch-free: a predicated load would sit in its own basic block behind its own wait
        // (16 serialised L2 round trips per chunk); load a safe address and select instead
        const int k = kc + (u * 64 + lane) * 4;
        const bool ok = n0 + c < N && k < K;
        const float4 v = *(const float4*)(w + (ok ? (long long)(n0 + c) * K + k : 0));
        wv[c][u] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    __syncthreads();  // previous chunk's readers are done
    {  // all LR_ROWS loads of this thread in flight before the first LDS store
      float4 t[LR_ROWS];
      const int k = kc + threadIdx.x * 4;
#pragma unroll
      for (int m = 0; m < LR_ROWS; ++m) {
        const bool ok = m0 + m < M && k < K;
        const float4 v = *(const float4*)(x + (ok ? (long long)(m0 + m) * K + k : 0));
        t[m] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int m = 0; m < LR_ROWS; ++m) xs4[m * (LR_KCH / 4) + threadIdx.x] = t[m];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int m = 0; m < LR_ROWS; ++m) {
        const float4 xv = xs4[m * (LR_KCH / 4) + u * 64 + lane];
#pragma unroll
        for (int c = 0; c < LR_CPW; ++c)
          acc[m][c] += xv.x * wv[c][u].x + xv.y * wv[c][u].y + xv.z * wv[c][u].z + xv.w * wv[c][u].w;
      }
    }
  }
  // cross-lane sums through LDS (the x buffer is free now): wave w stores accumulator a of lane l
  // at red[(w*64 + a) * 65 + l]; thread (w, a) then adds its row of 64 -- every thread finishes ONE
  // output.  (64 butterfly reductions on ds_bpermute were 384 dependent LDS-crossbar round trips
  // per wave: ~16 us per block.)
  __syncthreads();
  float* red = (float*)xs4;
#pragma unroll
  for (int m = 0; m < LR_ROWS; ++m)
#pragma unroll
    for (int c = 0; c < LR_CPW; ++c) red[(wave * 64 + m * LR_CPW + c) * 65 + lane] = acc[m][c];
  __syncthreads();
  {
    const float* row = red + (wave * 64 + lane) * 65;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 64; ++j) s += row[j];
    const int m = lane / LR_CPW, c = lane % LR_CPW;
    const int n = n0 + c;
    if (n < N && m0 + m < M) {
      s += b ? b[n] : 0.f;
      if (act == 1) s = fmaxf(s, 0.f);
      else if (act == 2) s = 0.5f * s * (1.0f + tanhf(0.7978845608028654f * (s + 0.044715f * s * s * s)));
      if (res) s += res[(long long)(m0 + m) * N + n];
      y[(long long)(m0 + m) * N + n] = s;
    }
  }
}

// 17..64 rows on the fp32 matrix cores (GPT-2 decode steps: sentences x beam rows).
// One block = NW waves that split K (wave w takes the 16-float chunk w of every 16*NW) and each
// accumulate the block's whole (RB*16 rows) x (CB*16 columns) tile with v_mfma_f32_16x16x4_f32;
// fragments come straight from global memory in the MFMA operand layout (lane = (row or column,
// k-quad): one float4 feeds four MFMAs), two register stages of CH chunks, loads unconditional and
// fenced from the MFMAs by scheduling barriers (hipcc otherwise sinks a stage's loads next to their
// first use, or waits vmcnt(0) at the join of a conditional load: no overlap); the NW partial tiles
// are added through LDS in wave order (bitwise reproducible).  These layers are a few MB: a launch is
// one memory latency + 2-4 us of MFMA work, so what counts is (a) enough blocks for 256 CUs without a
// K split across blocks -- measured, a slab + ticket combine costs three memory round trips (~6 us) --
// and (b) every load of a wave in flight from the start: wide K splits inside the block (8-16 waves,
// both register stages cover all of K = 1024) rather than a deep loop in 4 waves.  Tile shapes
// (launch_skinny):
//   <4,4,2, 4>  64 rows x 64 columns, 8 loads per 64 MFMAs   (lm_head: N / 64 = 786 blocks)
//   <4,1,4, 8>  64 rows x 16 columns, 5 loads per 16 MFMAs   (c_attn, mlp.c_fc: N / 16 = 192 / 256 blocks)
//   <1,1,4,16>  16 rows x 16 columns, 2 loads per  4 MFMAs   (attn.c_proj, mlp.c_proj with N = 1024:
//               256 blocks; the row groups of a column tile are neighbours on one XCD, so the weight
//               tile leaves HBM once)
typedef float lsk_v4f __attribute__((ext_vector_type(4)));

template <int RB, int CB, int CH>
struct LskStage {
  float4 a[RB][CH];
  float4 bw[CB][CH];
};

__device__ __forceinline__ float lsk_f4e(const float4& v, int e) {
  return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w;
}

template <int RB, int CB, int CH, int NW, bool PK>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void linear_skinny_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
    const float* res, float* y, int M, int N, int K, int act, int tiles, int rgroups, int ypk) {
  extern __shared__ lsk_v4f lsk_red[];  // [NW waves][RB*CB tiles][64 lanes]
  // block id -> (column tile, row group): ids are dealt round-robin to the 8 XCDs, so the row groups
  // of one column tile get consecutive slots of the same XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile = (slot / rgroups) * 8 + xcd, rg = slot % rgroups;
  if (tile >= tiles) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n0 = tile * (16 * CB), m0 = rg * (16 * RB), col = lane & 15, kq = lane >> 4;
  const int kbase = wave * 16 + kq * 4;
  constexpr int GK = 16 * NW * CH;  // floats of K per group (all waves, one stage)
  // PK: both operands in the fragment-major layout of vs_pack_rows_f32 (a 16-row x 16-float block =
  // 1 KB in lane order): every fragment load is 64 lanes x 16 contiguous bytes instead of 16 rows x 64 B
  // (which keeps the texture-address path busy 4x longer: 6.6 TB/s chip-wide was all it delivered)
  constexpr int CSTR = PK ? 256 : 16;  // floats between consecutive 16-float chunks of one row block
  const float* wp[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    if (PK) {
      const int nt16 = (N + 15) >> 4;
      const int t16 = n0 / 16 + cb < nt16 ? n0 / 16 + cb : nt16 - 1;
      wp[cb] = w + ((long long)t16 * (K >> 4) + wave) * 256 + lane * 4;
    } else {
      const int n = n0 + cb * 16 + col < N ? n0 + cb * 16 + col : N - 1;
      wp[cb] = w + (long long)n * K + kbase;
    }
  }
  const float* xp[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    if (PK) {
      xp[rb] = x + ((long long)(m0 / 16 + rb) * (K >> 4) + wave) * 256 + lane * 4;
    } else {
      const int m = m0 + rb * 16 + col < M ? m0 + rb * 16 + col : M - 1;
      xp[rb] = x + (long long)m * K + kbase;
    }
  }
  lsk_v4f acc[RB][CB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = lsk_v4f{0.f, 0.f, 0.f, 0.f};

  const int groups = K / GK;
  auto load = [&](LskStage<RB, CB, CH>& st, int g) {
    const int off = g * (NW * CH * CSTR);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) st.bw[cb][c] = *(const float4*)(wp[cb] + off + c * (NW * CSTR));
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) st.a[rb][c] = *(const float4*)(xp[rb] + off + c * (NW * CSTR));
    }
  };
  auto compute = [&](const LskStage<RB, CB, CH>& st) {
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int cb = 0; cb < CB; ++cb)
            acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(lsk_f4e(st.a[rb][c], e), lsk_f4e(st.bw[cb][c], e),
                                                               acc[rb][cb], 0, 0, 0);
  };
  LskStage<RB, CB, CH> s0, s1;
  load(s0, 0);
  const int glast = groups - 1;
  int g = 0;
  for (; g + 2 <= groups; g += 2) {
    load(s1, g + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(s0);
    __builtin_amdgcn_sched_barrier(0);
    if (g + 2 < groups) load(s0, g + 2 < glast ? g + 2 : glast);  // (uniform; skipped on the last trip)
    __builtin_amdgcn_sched_barrier(0);
    compute(s1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (g < groups) compute(s0);  // odd tail: the loop's last load (or the prologue's) fetched it
  constexpr int NT = RB * CB;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) lsk_red[(wave * NT + rb * CB + cb) * 64 + lane] = acc[rb][cb];
  __syncthreads();
  // wave w' finishes tiles w', w'+NW, ...: rows m0 + rb*16 + 4*(lane/16) + 0..3, column cb*16 + lane%16
  for (int t = wave; t < NT; t += NW) {
    const int rb = t / CB, cb = t % CB;
    lsk_v4f v = lsk_red[t * 64 + lane];
#pragma unroll
    for (int q = 1; q < NW; ++q) v += lsk_red[(q * NT + t) * 64 + lane];
    const int n = n0 + cb * 16 + col;
    if (n >= N) continue;
    const float bias = b ? b[n] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = m0 + rb * 16 + 4 * kq + e;
      if (m < M) {
        float sv = v[e] + bias;
        if (act == 1) sv = fmaxf(sv, 0.f);
        else if (act == 2) sv = 0.5f * sv * (1.0f + tanhf(0.7978845608028654f * (sv + 0.044715f * sv * sv * sv)));
        if (res) sv += res[(long long)m * N + n];
        if (ypk) y[(((long long)(m >> 4) * (N >> 4) + (n >> 4)) * 64 + ((n & 15) >> 2) * 16 + (m & 15)) * 4 + (n & 3)] = sv;
        else y[(long long)m * N + n] = sv;
      }
    }
  }
}

template <int RB, int CB, int CH, int NW, bool PK>
static void launch_skinny_cfg(const float* x, const float* w, const float* b, const float* res, float* y,
                              int M, int N, int K, int act, int ypk, hipStream_t st) {
  constexpr int smem = NW * RB * CB * 64 * 16;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)linear_skinny_kernel<RB, CB, CH, NW, PK>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr = true;
  }
  const int tiles = (N + 16 * CB - 1) / (16 * CB), tiles8 = (tiles + 7) & ~7;
  const int rgroups = (M + 16 * RB - 1) / (16 * RB);
  hipLaunchKernelGGL((linear_skinny_kernel<RB, CB, CH, NW, PK>), dim3(tiles8 * rgroups), dim3(64 * NW), smem, st,
                     x, w, b, res, y, M, N, K, act, tiles, rgroups, ypk);
}

template <bool PK>
static bool linear_skinny(const float* x, const float* w, const float* b, const float* res, float* y,
                          int M, int N, int K, int act, int ypk, hipStream_t st) {
  if ((!PK && M <= 16) || M > 64 || (K & 127) != 0) return false;
  if (((uintptr_t)x | (uintptr_t)w) & 15) return false;
  const int rbn = (M + 15) / 16;
  // the largest tile that still yields ~190 blocks; its K group (16 * NW * CH floats) must divide K
  if (N >= 190 * 64 || ((K & 511) != 0)) {
    if (rbn == 1) launch_skinny_cfg<1, 4, 2, 4, PK>(x, w, b, res, y, M, N, K, act, ypk, st);
    else if (rbn == 2) launch_skinny_cfg<2, 4, 2, 4, PK>(x, w, b, res, y, M, N, K, act, ypk, st);
    else if (rbn == 3) launch_skinny_cfg<3, 4, 2, 4, PK>(x, w, b, res, y, M, N, K, act, ypk, st);
    else launch_skinny_cfg<4, 4, 2, 4, PK>(x, w, b, res, y, M, N, K, act, ypk, st);
  } else if (N >= 190 * 16 || ((K & 1023) != 0)) {
    if (rbn == 1) launch_skinny_cfg<1, 1, 4, 8, PK>(x, w, b, res, y, M, N, K, act, ypk, st);
    else if (rbn == 2) launch_skinny_cfg<2, 1, 4, 8, PK>(x, w, b, res, y, M, N, K, act, ypk, st);
    else if (rbn == 3) launch_skinny_cfg<3, 1, 4, 8, PK>(x, w, b, res, y, M, N, K, act, ypk, st);
    else launch_skinny_cfg<4, 1, 4, 8, PK>(x, w, b, res, y, M, N, K, act, ypk, st);
  } else {
    launch_skinny_cfg<1, 1, 4, 16, PK>(x, w, b, res, y, M, N, K, act, ypk, st);
  }
  return true;
}

// Fragment-major copy of a row-major matrix src[R][K] (K % 16 == 0): block (r / 16, k / 16) of 16 x 16
// floats is 1 KB at index (r / 16) * (K / 16) + k / 16; inside, lane (k % 16 / 4) * 16 + r % 16 holds
// the four floats k % 4 = 0..3 -- the v_mfma_f32_16x16x4_f32 operand order, so a fragment load is one
// contiguous KB.  Rows up to the next multiple of 16 are zero-filled.
__global__ void pack_rows_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int K) {
  const long long blk = blockIdx.x;  // one 1-KB block per wave; 4 per workgroup
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int kc16 = K >> 4;
  const long long id = blk * 4 + wave;
  const long long nblk = (long long)((R + 15) >> 4) * kc16;
  if (id >= nblk) return;
  const int rblk = (int)(id / kc16), kc = (int)(id % kc16);
  const int r = rblk * 16 + (lane & 15), k = kc * 16 + (lane >> 4) * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r < R) v = *(const float4*)(src + (long long)r * K + k);
  *(float4*)(dst + id * 256 + lane * 4) = v;
}

extern "C" int vs_pack_rows_f32(const float* src, float* dst, int R, int K, void* stream) {
  VS_CHECK_ARG(src && dst && R > 0 && K > 0 && (K & 15) == 0, "K must be a multiple of 16");
  VS_CHECK_ARG((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "16-byte aligned pointers");
  const long long nblk = (long long)((R + 15) >> 4) * (K >> 4);
  hipLaunchKernelGGL(pack_rows_f32_kernel, dim3((unsigned)((nblk + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     src, dst, R, K);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

extern "C" int vs_gemm_nt_f32_packed(const float* x_packed, const float* w_packed, const float* b,
                                     const float* res, float* y, int M, int N, int K, int act,
                                     int y_packed, void* stream) {
  VS_CHECK_ARG(x_packed && w_packed && y && M > 0 && M <= 64 && N > 0 && K > 0, "1..64 rows");
  VS_CHECK_ARG(act >= 0 && act <= 2, "act must be 0 (none), 1 (relu) or 2 (gelu_new)");
  VS_CHECK_ARG((K & 127) == 0, "K must be a multiple of 128");
  VS_CHECK_ARG(!y_packed || (N & 15) == 0, "a packed output needs N % 16 == 0");
  VS_CHECK_ARG(linear_skinny<true>(x_packed, w_packed, b, res, y, M, N, K, act, y_packed, (hipStream_t)stream),
               "unaligned operands");
  VS_CHECK_LAUNCH();
  return VS_OK;
}

static int linear_small_m(const float* x, const float* w, const float* b, const float* res, float* y,
                          int M, int N, int K, int act, hipStream_t st) {
  const int gx = (N + 3) / 4;
  if (M <= 8 && (K & 3) == 0 && K > 2048 && K <= 4096) {
    launch_fullx<8, 16>(x, w, b, res, y, M, N, K, act, st);
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  if (M <= 16 && (K & 3) == 0 && K <= 2048) {
    const bool k1 = K <= 1024;
    if (M <= 8) {
      if (k1) launch_fullx<8, 4>(x, w, b, res, y, M, N, K, act, st);
      else launch_fullx<8, 8>(x, w, b, res, y, M, N, K, act, st);
    } else {
      if (k1) launch_fullx<16, 4>(x, w, b, res, y, M, N, K, act, st);
      else launch_fullx<16, 8>(x, w, b, res, y, M, N, K, act, st);
    }
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  if (M > 16 && (K & 3) == 0) {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)linear_rows16_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr = true;
    }
    hipLaunchKernelGGL(linear_rows16_kernel, dim3((N + 15) / 16, (M + LR_ROWS - 1) / LR_ROWS), dim3(256),
                       (size_t)4 * 64 * 65 * 4, st, x, w, b, res, y, M, N, K, act);  // >= LR_ROWS * LR_KCH * 4
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  if (M <= 8)
    hipLaunchKernelGGL(linear_fwd_kernel<8>, dim3(gx, 1), dim3(256), 0, st, x, w, b, res, y, M, N, K, act);
  else if (M <= 16)
    hipLaunchKernelGGL(linear_fwd_kernel<16>, dim3(gx, 1), dim3(256), 0, st, x, w, b, res, y, M, N, K, act);
  else if (M <= 32)
    hipLaunchKernelGGL(linear_fwd_kernel<32>, dim3(gx, 1), dim3(256), 0, st, x, w, b, res, y, M, N, K, act);
  else if (M <= 48)
    hipLaunchKernelGGL(linear_fwd_kernel<48>, dim3(gx, 1), dim3(256), 0, st, x, w, b, res, y, M, N, K, act);
  else
    hipLaunchKernelGGL(linear_fwd_kernel<64>, dim3(gx, (M + 63) / 64), dim3(256), 0, st, x, w, b, res,
                       y, M, N, K, act);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

extern "C" int vs_linear_fwd(const float* x, const float* w, const float* b, float* y, int M,
                             int N, int K, int relu, void* stream) {
  VS_CHECK_ARG(x && w && y && M > 0 && N > 0 && K > 0, "bad args");
  return linear_small_m(x, w, b, nullptr, y, M, N, K, relu ? 1 : 0, (hipStream_t)stream);
}

// y = act(x . w^T + b) + res, act 0 none / 1 relu / 2 gelu_new.  Up to 64 rows every weight row is
// streamed exactly once by one wave (HBM-bound decode steps); above that the fp32 matrix cores.
int vs_gemm_nt_f32_mfma(const float* x, const float* w, const float* b, const float* res, float* y,
                        int M, int N, int K, int act, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0);

extern "C" int vs_gemm_nt_f32(const float* x, const float* w, const float* b, const float* res,
                              float* y, int M, int N, int K, int act, void* stream) {
  VS_CHECK_ARG(x && w && y && M > 0 && N > 0 && K > 0, "bad args");
  VS_CHECK_ARG(act >= 0 && act <= 2, "act must be 0 (none), 1 (relu) or 2 (gelu_new)");
  if (linear_skinny<false>(x, w, b, res, y, M, N, K, act, 0, (hipStream_t)stream)) {
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  if (M <= 64) return linear_small_m(x, w, b, res, y, M, N, K, act, (hipStream_t)stream);
  return vs_gemm_nt_f32_mfma(x, w, b, res, y, M, N, K, act, (hipStream_t)stream);
}

// dx[M,K] = dy[M,N] @ W[N,K] is the forward kernel on the transposed weight
// wT[K,N] (made once per step by vs_transpose_f32): every weight row is again
// streamed exactly once and N/4.. blocks keep the chip busy.
__global__ void transpose_f32_kernel(const float* __restrict__ w, float* __restrict__ wt, int R,
                                     int C) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < R && c < C) ? w[(long long)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < R && c < C) wt[(long long)c * R + r] = tile[threadIdx.x][i];
  }
}

extern "C" int vs_transpose_f32(const float* w, float* wt, int R, int C, void* stream) {
  VS_CHECK_ARG(w && wt && R > 0 && C > 0, "bad args");
  hipLaunchKernelGGL(transpose_f32_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(32, 8), 0,
                     (hipStream_t)stream, w, wt, R, C);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

extern "C" int vs_linear_bwd_data(const float* dy, const float* wt, float* dx, int M, int N, int K,
                                  void* stream) {
  return vs_linear_fwd(dy, wt, nullptr, dx, M, K, N, 0, stream);
}

// dw[N,K] = dy^T x ; db[N] = sum_m dy.  thread per (n, 4 k's); the M rows in batches of 8 with every load
// of a batch in flight before the first use (a row-at-a-time loop with the vector / scalar choice made at
// run time was a chain of M memory latencies: 7.5 us at M = 8, on the critical path of the train step).
template <bool VEC>
__device__ __forceinline__ void linear_bwd_weight_body(const float* __restrict__ dy, const float* __restrict__ dymask,
                                                       const float* __restrict__ x, float* dw, float* db, int M,
                                                       int N, int K, int bid, int nblocks) {
  const int K4 = (K + 3) / 4;
  const long long total = (long long)N * K4;
  for (long long i = (long long)bid * blockDim.x + threadIdx.x; i < total;
       i += (long long)nblocks * blockDim.x) {
    const int n = (int)(i / K4);
    const int k = (int)(i - (long long)n * K4) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float bsum = 0.f;
    for (int m0 = 0; m0 < M; m0 += 8) {
      float dd[8];
      float4 xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int m = m0 + u < M ? m0 + u : 0;  // clamped: branch-free loads
        dd[u] = dy[(long long)m * N + n];
        if (dymask) dd[u] = dymask[(long long)m * N + n] > 0.f ? dd[u] : 0.f;
        const float* src = x + (long long)m * K + k;
        if (VEC) {
          xv[u] = *(const float4*)src;
        } else {
          xv[u].x = src[0];
          xv[u].y = k + 1 < K ? src[1] : 0.f;
          xv[u].z = k + 2 < K ? src[2] : 0.f;
          xv[u].w = k + 3 < K ? src[3] : 0.f;
        }
      }
      // Rows past M contribute through a ZEROED dy (acc + 0 * x == acc bit for bit), not through a predicated update:
      // with `if (m0 + u < M) acc += ...` the fused launch's copy of this body was compiled to v_pk_fma_f32 followed
      // two or three instructions later by v_cndmask_b32 reads of the packed result, and with an MFMA convolution
      // kernel running beside it on another stream (same process or another one) single 16-lane passes of those reads
      // saw the register BEFORE the packed FMA's write -- one term of one component of dW missing, in every launch of
      // 900 (tools/probes/linear_vs_mfma_neighbor.py, profiles/r03_linear_fused_neighbor.txt; the stand-alone kernel,
      // compiled to an unbroken v_pk_fma_f32 chain, never showed it).  An unconditional chain in both copies.
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float d = (m0 + u < M) ? dd[u] : 0.f;  // same order as a plain loop over m
        acc.x = __fmaf_rn(d, xv[u].x, acc.x);
        acc.y = __fmaf_rn(d, xv[u].y, acc.y);
        acc.z = __fmaf_rn(d, xv[u].z, acc.z);
        acc.w = __fmaf_rn(d, xv[u].w, acc.w);
        bsum += d;
      }
    }
    float* dst = dw + (long long)n * K + k;
    if (VEC) {  // (read next by the optimizer at the end of the step: a non-temporal store)
      typedef __attribute__((ext_vector_type(4))) float nt_f4;
      const nt_f4 t = {acc.x, acc.y, acc.z, acc.w};
      __builtin_nontemporal_store(t, (nt_f4*)dst);
    } else {
      dst[0] = acc.x;
      if (k + 1 < K) dst[1] = acc.y;
      if (k + 2 < K) dst[2] = acc.z;
      if (k + 3 < K) dst[3] = acc.w;
    }
    if (db && k == 0) db[n] = bsum;
  }
}

template <bool VEC>
__global__ void linear_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                         float* dw, float* db, int M, int N, int K) {
  linear_bwd_weight_body<VEC>(dy, nullptr, x, dw, db, M, N, K, blockIdx.x, gridDim.x);
}

// Both gradients of y = x W^T (+ b) behind ONE launch for the few-row case (the 8 tokens of the encoder / heads):
// blocks [0, g1) are the data-gradient product dx = dy_eff . W on the transposed image wt (linear_fullx_body),
// blocks [g1, g1 + g2) the weight gradient dW = dy_eff^T x, db = colsum dy_eff (linear_bwd_weight_body), with
// dy_eff = dy * (relu_y > 0) when the layer had a ReLU -- bitwise the three launches (relu_bwd, bwd_data,
// bwd_weight) this replaces; every launch of this section is ~6 us of pure latency on the step's critical path.
template <int MT, int KC, bool VEC>
__global__ __launch_bounds__(256) void linear_bwd_fused_kernel(const float* __restrict__ dy,
                                                               const float* __restrict__ relu_y,
                                                               const float* __restrict__ x,
                                                               const float* __restrict__ wt, float* dx, float* dw,
                                                               float* db, int M, int N, int K, int g1, int g2,
                                                               const float* dx_res) {
  if ((int)blockIdx.x < g1)
    linear_fullx_body<MT, KC>(dy, relu_y, wt, nullptr, dx_res, dx, M, K, N, 0, blockIdx.x);
  else
    linear_bwd_weight_body<VEC>(dy, relu_y, x, dw, db, M, N, K, blockIdx.x - g1, g2);
}

template <int KC, bool VEC>
static void launch_bwd_fused(const float* dy, const float* relu_y, const float* x, const float* wt, float* dx,
                             float* dw, float* db, int M, int N, int K, hipStream_t st, const float* dx_res) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)linear_bwd_fused_kernel<8, KC, VEC>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  size_t smem = (size_t)8 * (N < 1024 ? N : 1024) * 4;  // one staged chunk of the dy rows (inner dimension N)
  if (smem < (size_t)4 * 8 * 65 * 4) smem = (size_t)4 * 8 * 65 * 4;
  const int g1 = (K + 3) / 4;  // four output columns (of dx) per block
  long long g2 = ((long long)N * ((K + 3) / 4) + 255) / 256;
  if (g2 > 4096) g2 = 4096;
  {
    // tools only (timing experiment, GARBAGE parameter gradients): what the weight-gradient half of these launches costs
    // the step -- the upper bound of moving it to a side stream.  Needs VS_WHATIF_OK=1 beside it.
    static const int skip_dw = [] {
      const char* e = getenv("VS_WHATIF_LINEAR_DW"); const char* ok = getenv("VS_WHATIF_OK");
      return (e && atoi(e) && ok && atoi(ok)) ? 1 : 0; }();
    if (skip_dw) g2 = 0;
  }
  hipLaunchKernelGGL((linear_bwd_fused_kernel<8, KC, VEC>), dim3((unsigned)(g1 + g2)), dim3(256), smem, st, dy,
                     relu_y, x, wt, dx, dw, db, M, N, K, g1, (int)g2, dx_res);
}

extern "C" int vs_linear_bwd_fused_res(const float* dy, const float* relu_y, const float* x, const float* wt,
                                       const float* dx_res, float* dx, float* dw, float* db, int M, int N, int K,
                                       void* stream) {
  VS_CHECK_ARG(dy && x && wt && dx && dw && M > 0 && N > 0 && K > 0, "bad args");
  const uintptr_t al = (uintptr_t)dy | (uintptr_t)relu_y | (uintptr_t)wt;
  if (M > 8 || (N & 3) || N > 4096 || (al & 15)) {
    vs_set_error("vs_linear_bwd_fused: M <= 8, N %% 4 == 0, N <= 4096 and 16-byte aligned dy / relu_y / wt "
                 "(use vs_linear_bwd_data + vs_linear_bwd_weight)");
    return VS_ERR_UNSUPPORTED;
  }
  const bool vec = (K & 3) == 0 && ((((uintptr_t)x | (uintptr_t)dw)) & 15) == 0;
  hipStream_t st = (hipStream_t)stream;
#define VS_LBF(KC_) (vec ? launch_bwd_fused<KC_, true>(dy, relu_y, x, wt, dx, dw, db, M, N, K, st, dx_res) \
                         : launch_bwd_fused<KC_, false>(dy, relu_y, x, wt, dx, dw, db, M, N, K, st, dx_res))
  if (N <= 1024) VS_LBF(4);
  else if (N <= 2048) VS_LBF(8);
  else VS_LBF(16);
#undef VS_LBF
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    vs_set_error("vs_linear_bwd_fused: launch failed: %s", hipGetErrorString(e));
    return VS_ERR_LAUNCH;
  }
  return VS_OK;
}

extern "C" int vs_linear_bwd_fused(const float* dy, const float* relu_y, const float* x, const float* wt, float* dx,
                                   float* dw, float* db, int M, int N, int K, void* stream) {
  return vs_linear_bwd_fused_res(dy, relu_y, x, wt, nullptr, dx, dw, db, M, N, K, stream);
}

extern "C" int vs_linear_bwd_weight(const float* dy, const float* x, float* dw, float* db, int M,
                                    int N, int K, void* stream) {
  VS_CHECK_ARG(dy && x && dw && M > 0 && N > 0 && K > 0, "bad args");
  const long long total = (long long)N * ((K + 3) / 4);
  long long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  const bool vec = (K & 3) == 0 && ((((uintptr_t)x | (uintptr_t)dw)) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(linear_bwd_weight_kernel<true>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dy,
                       x, dw, db, M, N, K);
  else
    hipLaunchKernelGGL(linear_bwd_weight_kernel<false>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dy,
                       x, dw, db, M, N, K);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ----------------------------------------------------------------------------
// Attention for L <= 16: one 256-thread block per (batch, head); q/k/v of the head
// live in LDS (L*dh*3 floats), scores and probabilities in LDS [L][L].
// probs saved as [B][H][L][L] for the backward.
// ----------------------------------------------------------------------------
#define ATT_MAXL 16

// (bid: the block's (batch, head) index -- blockIdx.x of the stand-alone launch, a virtual block of the
//  one-launch encoder stack below)
__device__ __forceinline__ void attn_small_fwd_body(const float* q, const float* k, const float* v, float* o,
                                                    float* probs, const float* drop_mask, int L, int H, int dh,
                                                    float inv_scale, int ld, int bid, char* smem_att) {
  float* qs = (float*)smem_att;  // [L][dh]
  float* ks = qs + L * dh;
  float* vs = ks + L * dh;
  float* ps = vs + L * dh;  // [L][L]
  const int b = bid / H, h = bid % H;
  const int D = H * dh;
  for (int i = threadIdx.x; i < L * dh; i += 256) {
    const int r = i / dh, d = i - r * dh;
    const long long off = ((long long)b * L + r) * ld + h * dh + d;  // ld: row pitch of q / k / v
    qs[i] = q[off];
    ks[i] = k[off];
    vs[i] = v[off];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < L * L; e += 256) {
    const int i = e / L, j = e - i * L;
    float s = 0.f;
    for (int d = 0; d < dh; ++d) s += qs[i * dh + d] * ks[j * dh + d];
    ps[e] = s * inv_scale;
  }
  __syncthreads();
  if (threadIdx.x < L) {
    const int i = threadIdx.x;
    float mx = -INFINITY;
    for (int j = 0; j < L; ++j) mx = fmaxf(mx, ps[i * L + j]);
    float den = 0.f;
    for (int j = 0; j < L; ++j) {
      const float ev = expf(ps[i * L + j] - mx);
      ps[i * L + j] = ev;
      den += ev;
    }
    const float inv = 1.0f / den;
    for (int j = 0; j < L; ++j) {
      const float pj = ps[i * L + j] * inv;
      const long long pidx = (((long long)b * H + h) * L + i) * L + j;
      if (probs) probs[pidx] = pj;  // pre-dropout probabilities (softmax backward needs them)
      ps[i * L + j] = drop_mask ? pj * drop_mask[pidx] : pj;
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < L * dh; e += 256) {
    const int i = e / dh, d = e - i * dh;
    float acc = 0.f;
    for (int j = 0; j < L; ++j) acc += ps[i * L + j] * vs[j * dh + d];
    o[((long long)b * L + i) * D + h * dh + d] = acc;
  }
}

__global__ __launch_bounds__(256) void attn_small_fwd_kernel(const float* q, const float* k,
                                                             const float* v, float* o, float* probs,
                                                             const float* drop_mask, int L, int H,
                                                             int dh, float inv_scale, int ld) {
  extern __shared__ __attribute__((aligned(16))) char smem_att[];
  attn_small_fwd_body(q, k, v, o, probs, drop_mask, L, H, dh, inv_scale, ld, blockIdx.x, smem_att);
}

extern "C" int vs_attn_small_fwd(const float* q, const float* k, const float* v, float* o,
                                 float* probs, const float* drop_mask, int B, int L, int H, int dh,
                                 int ld_qkv, float scale, void* stream) {
  VS_CHECK_ARG(q && k && v && o, "null tensor");
  VS_CHECK_ARG(L >= 1 && L <= ATT_MAXL && dh >= 1 && dh <= 512, "L <= 16, dh <= 512");
  VS_CHECK_ARG(ld_qkv == 0 || ld_qkv >= H * dh, "row pitch of q / k / v");
  const size_t smem = (size_t)(3 * L * dh + L * L) * sizeof(float);
  hipLaunchKernelGGL(attn_small_fwd_kernel, dim3(B * H), dim3(256), smem, (hipStream_t)stream, q, k,
                     v, o, probs, drop_mask, L, H, dh, 1.0f / scale, ld_qkv ? ld_qkv : H * dh);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

__device__ __forceinline__ void attn_small_bwd_body(const float* q, const float* k, const float* v,
                                                    const float* probs, const float* dout, float* dq, float* dk,
                                                    float* dv, const float* drop_mask, int L, int H, int dh,
                                                    float inv_scale, int ld, int bid, char* smem_att) {
  float* qs = (float*)smem_att;  // [L][dh]
  float* ks = qs + L * dh;
  float* vs = ks + L * dh;
  float* dos = vs + L * dh;
  float* ps = dos + L * dh;  // [L][L]  pre-dropout probabilities
  float* dss = ps + L * L;   // [L][L]
  float* ms = dss + L * L;   // [L][L]  dropout mask (1 when absent)
  const int b = bid / H, h = bid % H;
  const int D = H * dh;
  for (int i = threadIdx.x; i < L * dh; i += 256) {
    const int r = i / dh, d = i - r * dh;
    const long long off = ((long long)b * L + r) * ld + h * dh + d;  // ld: pitch of q/k/v and dq/dk/dv
    qs[i] = q[off];
    ks[i] = k[off];
    vs[i] = v[off];
    dos[i] = dout[((long long)b * L + r) * D + h * dh + d];
  }
  for (int e = threadIdx.x; e < L * L; e += 256) {
    ps[e] = probs[((long long)b * H + h) * L * L + e];
    ms[e] = drop_mask ? drop_mask[((long long)b * H + h) * L * L + e] : 1.f;
  }
  __syncthreads();
  // dP[i][j] = mask * (dO[i] . V[j])
  for (int e = threadIdx.x; e < L * L; e += 256) {
    const int i = e / L, j = e - i * L;
    float s = 0.f;
    for (int d = 0; d < dh; ++d) s += dos[i * dh + d] * vs[j * dh + d];
    dss[e] = s * ms[e];
  }
  __syncthreads();
  // dS = P * (dP - sum_j dP*P) / scale
  if (threadIdx.x < L) {
    const int i = threadIdx.x;
    float dot = 0.f;
    for (int j = 0; j < L; ++j) dot += dss[i * L + j] * ps[i * L + j];
    for (int j = 0; j < L; ++j) dss[i * L + j] = ps[i * L + j] * (dss[i * L + j] - dot) * inv_scale;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < L * dh; e += 256) {
    const int r = e / dh, d = e - r * dh;
    float aq = 0.f, ak = 0.f, av = 0.f;
    for (int j = 0; j < L; ++j) {
      aq += dss[r * L + j] * ks[j * dh + d];   // dQ[r] = sum_j dS[r][j] K[j]
      ak += dss[j * L + r] * qs[j * dh + d];   // dK[r] = sum_i dS[i][r] Q[i]
      av += ps[j * L + r] * ms[j * L + r] * dos[j * dh + d];  // dV[r] = sum_i P'[i][r] dO[i]
    }
    const long long off = ((long long)b * L + r) * ld + h * dh + d;
    dq[off] = aq;
    dk[off] = ak;
    dv[off] = av;
  }
}

__global__ __launch_bounds__(256) void attn_small_bwd_kernel(const float* q, const float* k,
                                                             const float* v, const float* probs,
                                                             const float* dout, float* dq, float* dk,
                                                             float* dv, const float* drop_mask,
                                                             int L, int H, int dh, float inv_scale, int ld) {
  extern __shared__ __attribute__((aligned(16))) char smem_att[];
  attn_small_bwd_body(q, k, v, probs, dout, dq, dk, dv, drop_mask, L, H, dh, inv_scale, ld, blockIdx.x, smem_att);
}

extern "C" int vs_attn_small_bwd(const float* q, const float* k, const float* v,
                                 const float* probs, const float* dout, float* dq, float* dk,
                                 float* dv, const float* drop_mask, int B, int L, int H, int dh,
                                 int ld_qkv, float scale, void* stream) {
  VS_CHECK_ARG(q && k && v && probs && dout && dq && dk && dv, "null tensor");
  VS_CHECK_ARG(L >= 1 && L <= ATT_MAXL && dh >= 1 && dh <= 512, "L <= 16, dh <= 512");
  VS_CHECK_ARG(ld_qkv == 0 || ld_qkv >= H * dh, "row pitch of q / k / v");
  const size_t smem = (size_t)(4 * L * dh + 3 * L * L) * sizeof(float);
  hipLaunchKernelGGL(attn_small_bwd_kernel, dim3(B * H), dim3(256), smem, (hipStream_t)stream, q, k,
                     v, probs, dout, dq, dk, dv, drop_mask, L, H, dh, 1.0f / scale,
                     ld_qkv ? ld_qkv : H * dh);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ----------------------------------------------------------------------------
// y = LayerNorm(x + r): one wave per row, two-pass mean / centred variance.
// ----------------------------------------------------------------------------
#define LN_MAXE 32  // elements per lane (D <= 2048)

__device__ __forceinline__ float ln_input(const float* x, const float* r, const float* rmask,
                                          long long i) {
  float v = x[i];
  if (r) v += rmask ? r[i] * rmask[i] : r[i];
  return v;
}

__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const float* x, const float* r,
                                                                const float* rmask,
                                                                const float* gamma,
                                                                const float* beta, float* y,
                                                                float* mean, float* rstd, int rows,
                                                                int D, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float v[LN_MAXE];
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < LN_MAXE; ++e) {
    const int d = lane + 64 * e;
    v[e] = 0.f;
    if (d < D) {
      v[e] = ln_input(x, r, rmask, (long long)row * D + d);
      s += v[e];
    }
  }
  const float mu = wave_reduce_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < LN_MAXE; ++e) {
    const int d = lane + 64 * e;
    if (d < D) {
      const float c = v[e] - mu;
      q += c * c;
    }
  }
  const float rs = rsqrtf(wave_reduce_sum(q) / (float)D + eps);
#pragma unroll
  for (int e = 0; e < LN_MAXE; ++e) {
    const int d = lane + 64 * e;
    if (d < D) y[(long long)row * D + d] = (v[e] - mu) * rs * gamma[d] + beta[d];
  }
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
}

// D % 4 == 0: 16-byte accesses, every load of the row issued before the first use and no load inside
// a divergent branch (the scalar kernel's `if (d < D)` loads were 16 serialised round trips: 12.5 us
// for 50 x 1024)
__device__ __forceinline__ void add_layernorm_fwd_vec_body(
    const float* x, const float* r, const float* rmask, const float* gamma, const float* beta, float* y,
    float* mean, float* rstd, int rows, int D, float eps, int ypk, int row) {
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  constexpr int NE = LN_MAXE / 4;
  const int D4 = D >> 2;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* x4 = (const float4*)(x + (long long)row * D);
  float4 v[NE], g4[NE], b4[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int d = lane + 64 * e;
    const bool ok = d < D4;
    const float4 t = x4[ok ? d : 0];
    v[e] = ok ? t : z4;
    g4[e] = ((const float4*)gamma)[ok ? d : 0];
    b4[e] = ((const float4*)beta)[ok ? d : 0];
  }
  if (r) {
    const float4* r4 = (const float4*)(r + (long long)row * D);
    const float4* m4 = rmask ? (const float4*)(rmask + (long long)row * D) : nullptr;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int d = lane + 64 * e;
      const bool ok = d < D4;
      float4 t = r4[ok ? d : 0];
      if (m4) {
        const float4 m = m4[ok ? d : 0];
        t.x *= m.x; t.y *= m.y; t.z *= m.z; t.w *= m.w;
      }
      if (ok) { v[e].x += t.x; v[e].y += t.y; v[e].z += t.z; v[e].w += t.w; }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < NE; ++e) s += (v[e].x + v[e].y) + (v[e].z + v[e].w);
  const float mu = wave_reduce_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    if (lane + 64 * e < D4) {
      const float a = v[e].x - mu, b = v[e].y - mu, c = v[e].z - mu, d2 = v[e].w - mu;
      q += (a * a + b * b) + (c * c + d2 * d2);
    }
  }
  const float rs = rsqrtf(wave_reduce_sum(q) / (float)D + eps);
  // ypk: fragment-major output (vs_pack_rows_f32's layout) for vs_gemm_nt_f32_packed; D % 16 == 0
  float4* y4 = ypk ? (float4*)y + ((long long)(row >> 4) * (D >> 4)) * 64 + (row & 15)
                   : (float4*)(y + (long long)row * D);
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int d = lane + 64 * e;
    if (d < D4) {
      float4 o;
      o.x = (v[e].x - mu) * rs * g4[e].x + b4[e].x;
      o.y = (v[e].y - mu) * rs * g4[e].y + b4[e].y;
      o.z = (v[e].z - mu) * rs * g4[e].z + b4[e].z;
      o.w = (v[e].w - mu) * rs * g4[e].w + b4[e].w;
      y4[ypk ? (d >> 2) * 64 + (d & 3) * 16 : d] = o;
    }
  }
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
}

__global__ __launch_bounds__(256) void add_layernorm_fwd_vec_kernel(
    const float* x, const float* r, const float* rmask, const float* gamma, const float* beta, float* y,
    float* mean, float* rstd, int rows, int D, float eps, int ypk) {
  add_layernorm_fwd_vec_body(x, r, rmask, gamma, beta, y, mean, rstd, rows, D, eps, ypk,
                             blockIdx.x * 4 + (threadIdx.x >> 6));
}

extern "C" int vs_add_layernorm_fwd(const float* x, const float* r, const float* rmask,
                                    const float* gamma, const float* beta, float* y, float* mean,
                                    float* rstd, int rows, int D, float eps, void* stream) {
  VS_CHECK_ARG(x && gamma && beta && y && rows > 0, "bad args");
  VS_CHECK_ARG(D >= 1 && D <= 64 * LN_MAXE, "D <= 2048");
  const uintptr_t al = (uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)r |
                       (uintptr_t)rmask;
  if ((D & 3) == 0 && (al & 15) == 0)
    hipLaunchKernelGGL(add_layernorm_fwd_vec_kernel, dim3((rows + 3) / 4), dim3(256), 0,
                       (hipStream_t)stream, x, r, rmask, gamma, beta, y, mean, rstd, rows, D, eps, 0);
  else
    hipLaunchKernelGGL(add_layernorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0,
                       (hipStream_t)stream, x, r, rmask, gamma, beta, y, mean, rstd, rows, D, eps);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

/* LayerNorm forward whose output is written fragment-major (vs_pack_rows_f32's layout, rows padded
 * to 16 by the caller's allocation) -- the x operand of vs_gemm_nt_f32_packed. */
extern "C" int vs_layernorm_fwd_packed(const float* x, const float* gamma, const float* beta,
                                       float* y_packed, int rows, int D, float eps, void* stream) {
  VS_CHECK_ARG(x && gamma && beta && y_packed && rows > 0, "bad args");
  VS_CHECK_ARG(D >= 16 && D <= 64 * LN_MAXE && (D & 15) == 0, "D % 16 == 0, D <= 2048");
  VS_CHECK_ARG((((uintptr_t)x | (uintptr_t)y_packed | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0,
               "16-byte aligned pointers");
  hipLaunchKernelGGL(add_layernorm_fwd_vec_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x,
                     (const float*)nullptr, (const float*)nullptr, gamma, beta, y_packed, (float*)nullptr,
                     (float*)nullptr, rows, D, eps, 1);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// dx = rstd * (g - mean(g) - xhat*mean(g*xhat)), g = gamma*dy ; per-row wave
__global__ __launch_bounds__(256) void add_layernorm_bwd_dx_kernel(
    const float* dy, const float* x, const float* r, const float* rmask, const float* gamma,
    const float* mean, const float* rstd, float* dx, float* dr, int rows, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float mu = mean[row], rs = rstd[row];
  float g[LN_MAXE], xh[LN_MAXE];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int e = 0; e < LN_MAXE; ++e) {
    const int d = lane + 64 * e;
    g[e] = 0.f;
    xh[e] = 0.f;
    if (d < D) {
      const float v = ln_input(x, r, rmask, (long long)row * D + d);
      xh[e] = (v - mu) * rs;
      g[e] = gamma[d] * dy[(long long)row * D + d];
      s1 += g[e];
      s2 += g[e] * xh[e];
    }
  }
  const float m1 = wave_reduce_sum(s1) / (float)D, m2 = wave_reduce_sum(s2) / (float)D;
#pragma unroll
  for (int e = 0; e < LN_MAXE; ++e) {
    const int d = lane + 64 * e;
    if (d < D) {
      const float dv = rs * (g[e] - m1 - xh[e] * m2);
      dx[(long long)row * D + d] = dv;
      if (dr) dr[(long long)row * D + d] = rmask ? dv * rmask[(long long)row * D + d] : dv;
    }
  }
}

// dgamma[d] = sum_rows dy*xhat ; dbeta[d] = sum_rows dy   (thread per column)
__global__ void add_layernorm_bwd_param_kernel(const float* dy, const float* x, const float* r,
                                               const float* rmask, const float* mean,
                                               const float* rstd, float* dgamma, float* dbeta,
                                               int rows, int D) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  float sg = 0.f, sb = 0.f;
  for (int row = 0; row < rows; ++row) {
    const float v = ln_input(x, r, rmask, (long long)row * D + d);
    const float g = dy[(long long)row * D + d];
    sg += g * (v - mean[row]) * rstd[row];
    sb += g;
  }
  dgamma[d] = sg;
  dbeta[d] = sb;
}

// D % 4 == 0: 16-byte branch-free loads, everything in flight before the first use (the scalar kernel's
// `if (d < D)` loads were serialised round trips: 18 us for 8 x 1024)
template <int NE>  // float4 columns per lane: D <= 256 * NE
__device__ __forceinline__ void add_layernorm_bwd_dx_vec_body(
    const float* dy, const float* x, const float* r, const float* rmask, const float* gamma,
    const float* mean, const float* rstd, float* dx, float* dr, int rows, int D, int row,
    const float* dy2 = nullptr) {  // dy2: a second addend of the output gradient (the encoder stack's fan-ins)
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int D4 = D >> 2;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long base = (long long)row * D4;
  const float mu = mean[row], rs = rstd[row];
  float4 v[NE], g[NE], mk[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int d = lane + 64 * e;
    const bool ok = d < D4;
    const long long i = base + (ok ? d : 0);
    const float4 xv = ((const float4*)x)[ok ? i : 0];
    float4 dv = ((const float4*)dy)[ok ? i : 0];
    if (dy2) {
      const float4 d2 = ((const float4*)dy2)[ok ? i : 0];
      dv.x += d2.x; dv.y += d2.y; dv.z += d2.z; dv.w += d2.w;
    }
    const float4 gv = ((const float4*)gamma)[ok ? d : 0];
    v[e] = ok ? xv : z4;
    g[e] = ok ? make_float4(gv.x * dv.x, gv.y * dv.y, gv.z * dv.z, gv.w * dv.w) : z4;
    mk[e] = make_float4(1.f, 1.f, 1.f, 1.f);
  }
  if (r) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int d = lane + 64 * e;
      const bool ok = d < D4;
      const long long i = base + (ok ? d : 0);
      float4 t = ((const float4*)r)[ok ? i : 0];
      if (rmask) {
        const float4 m = ((const float4*)rmask)[ok ? i : 0];
        mk[e] = m;
        t.x *= m.x; t.y *= m.y; t.z *= m.z; t.w *= m.w;
      }
      if (ok) { v[e].x += t.x; v[e].y += t.y; v[e].z += t.z; v[e].w += t.w; }
    }
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const bool ok = lane + 64 * e < D4;
    v[e].x = (v[e].x - mu) * rs; v[e].y = (v[e].y - mu) * rs;
    v[e].z = (v[e].z - mu) * rs; v[e].w = (v[e].w - mu) * rs;  // xhat
    if (ok) {
      s1 += (g[e].x + g[e].y) + (g[e].z + g[e].w);
      s2 += (g[e].x * v[e].x + g[e].y * v[e].y) + (g[e].z * v[e].z + g[e].w * v[e].w);
    }
  }
  const float m1 = wave_reduce_sum(s1) / (float)D, m2 = wave_reduce_sum(s2) / (float)D;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int d = lane + 64 * e;
    if (d < D4) {
      float4 o;
      o.x = rs * (g[e].x - m1 - v[e].x * m2);
      o.y = rs * (g[e].y - m1 - v[e].y * m2);
      o.z = rs * (g[e].z - m1 - v[e].z * m2);
      o.w = rs * (g[e].w - m1 - v[e].w * m2);
      ((float4*)dx)[base + d] = o;
      if (dr) ((float4*)dr)[base + d] = make_float4(o.x * mk[e].x, o.y * mk[e].y, o.z * mk[e].z, o.w * mk[e].w);
    }
  }
}

__global__ __launch_bounds__(256) void add_layernorm_bwd_dx_vec_kernel(
    const float* dy, const float* x, const float* r, const float* rmask, const float* gamma,
    const float* mean, const float* rstd, float* dx, float* dr, int rows, int D) {
  add_layernorm_bwd_dx_vec_body<LN_MAXE / 4>(dy, x, r, rmask, gamma, mean, rstd, dx, dr, rows, D,
                                             blockIdx.x * 4 + (threadIdx.x >> 6));
}

// dgamma / dbeta: a block owns 64 columns, its 16 waves take rows w, w+16, ... (256-byte row pieces,
// four rows of loads in flight), partial sums meet in LDS in wave order (bitwise reproducible).  The
// one-thread-per-column loop over all rows was a chain of rows x latency (600-row decoder batches).
__device__ __forceinline__ void add_layernorm_bwd_param_rows_body(
    const float* dy, const float* x, const float* r, const float* rmask, const float* mean,
    const float* rstd, float* dgamma, float* dbeta, int rows, int D, int bid) {
  __shared__ float sg[16][64], sb[16][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int d = bid * 64 + lane;
  const bool okd = d < D;
  const int dc = okd ? d : 0;
  float ag = 0.f, ab = 0.f;
  for (int row0 = wave; row0 < rows; row0 += 64) {
    float vv[4], gg[4], mu[4], rs[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = row0 + 16 * u;
      const bool ok = row < rows;
      const long long i = (long long)(ok ? row : 0) * D + dc;
      float v = x[i];
      if (r) v += rmask ? r[i] * rmask[i] : r[i];
      vv[u] = v;
      gg[u] = ok ? dy[i] : 0.f;
      mu[u] = mean[ok ? row : 0];
      rs[u] = rstd[ok ? row : 0];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // spelled out: the one-launch stack's copy (tx_ln_bwd_param_body) must round alike
      ag = __fadd_rn(ag, __fmul_rn(__fmul_rn(gg[u], vv[u] - mu[u]), rs[u]));
      ab = __fadd_rn(ab, gg[u]);
    }
  }
  sg[wave][lane] = ag;
  sb[wave][lane] = ab;
  __syncthreads();
  if (wave == 0 && okd) {
    float a = 0.f, b2 = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      a += sg[w][lane];
      b2 += sb[w][lane];
    }
    dgamma[d] = a;
    dbeta[d] = b2;
  }
}

__global__ __launch_bounds__(1024) void add_layernorm_bwd_param_rows_kernel(
    const float* dy, const float* x, const float* r, const float* rmask, const float* mean,
    const float* rstd, float* dgamma, float* dbeta, int rows, int D) {
  add_layernorm_bwd_param_rows_body(dy, x, r, rmask, mean, rstd, dgamma, dbeta, rows, D, blockIdx.x);
}

// Both passes behind one launch (few rows: the encoder's 8 tokens): blocks [0, g1) are the input-gradient pass with
// 16 rows per 1024-thread block, blocks [g1, ..) the parameter-gradient pass -- the same bodies, bitwise the two
// launches; one ~6 us launch less per LayerNorm on the step's critical path.
__global__ __launch_bounds__(1024) void add_layernorm_bwd_fused_kernel(
    const float* dy, const float* x, const float* r, const float* rmask, const float* gamma, const float* mean,
    const float* rstd, float* dx, float* dr, float* dgamma, float* dbeta, int rows, int D, int g1) {
  if ((int)blockIdx.x < g1)
    add_layernorm_bwd_dx_vec_body<4>(dy, x, r, rmask, gamma, mean, rstd, dx, dr, rows, D,  // D <= 1024: 128 VGPRs
                                     blockIdx.x * 16 + (threadIdx.x >> 6));
  else
    add_layernorm_bwd_param_rows_body(dy, x, r, rmask, mean, rstd, dgamma, dbeta, rows, D, blockIdx.x - g1);
}

extern "C" int vs_add_layernorm_bwd(const float* dy, const float* x, const float* r,
                                    const float* rmask, const float* gamma, const float* mean,
                                    const float* rstd, float* dx, float* dr, float* dgamma,
                                    float* dbeta, int rows, int D, void* stream) {
  VS_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dgamma && dbeta, "null tensor");
  VS_CHECK_ARG(D >= 1 && D <= 64 * LN_MAXE, "D <= 2048");
  const uintptr_t al = (uintptr_t)dy | (uintptr_t)x | (uintptr_t)r | (uintptr_t)rmask | (uintptr_t)gamma |
                       (uintptr_t)dx | (uintptr_t)dr;
  static const int fuse = [] { const char* e = getenv("VS_LN_BWD_FUSED"); return e ? atoi(e) : 1; }();
  if (fuse && (D & 3) == 0 && D <= 1024 && (al & 15) == 0 && rows <= 64) {
    const int g1 = (rows + 15) / 16;
    hipLaunchKernelGGL(add_layernorm_bwd_fused_kernel, dim3(g1 + (D + 63) / 64), dim3(1024), 0, (hipStream_t)stream,
                       dy, x, r, rmask, gamma, mean, rstd, dx, dr, dgamma, dbeta, rows, D, g1);
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  if ((D & 3) == 0 && (al & 15) == 0)
    hipLaunchKernelGGL(add_layernorm_bwd_dx_vec_kernel, dim3((rows + 3) / 4), dim3(256), 0,
                       (hipStream_t)stream, dy, x, r, rmask, gamma, mean, rstd, dx, dr, rows, D);
  else
    hipLaunchKernelGGL(add_layernorm_bwd_dx_kernel, dim3((rows + 3) / 4), dim3(256), 0,
                       (hipStream_t)stream, dy, x, r, rmask, gamma, mean, rstd, dx, dr, rows, D);
  hipLaunchKernelGGL(add_layernorm_bwd_param_rows_kernel, dim3((D + 63) / 64), dim3(1024), 0,
                     (hipStream_t)stream, dy, x, r, rmask, mean, rstd, dgamma, dbeta, rows, D);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ----------------------------------------------------------------------------
// mean cross-entropy + dlogits.  One block; waves take rows round-robin and the
// per-wave loss sums are combined in a fixed order (bitwise reproducible).
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_xent_kernel(const float* logits,
                                                           const int64_t* labels, float* loss,
                                                           float* dlogits, int rows, int V) {
  __shared__ float wsum[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float lsum = 0.f;
  const float invR = 1.0f / (float)rows;
  if (V <= 64 * 32) {
    // the whole row in registers (32 per lane), every load in flight before the first use: the three
    // passes below re-read memory in dependent little loops (28 us for 8 x 1564 on the step's critical path)
    for (int row = wave; row < rows; row += 4) {
      const float* lg = logits + (long long)row * V;
      float v[32];
      float mx = -INFINITY;
#pragma unroll
      for (int e = 0; e < 32; ++e) {
        const int j = lane + 64 * e;
        const float t = lg[j < V ? j : 0];
        v[e] = j < V ? t : -INFINITY;
        mx = fmaxf(mx, v[e]);
      }
      mx = wave_reduce_max(mx);
      float den = 0.f;
#pragma unroll
      for (int e = 0; e < 32; ++e) {
        v[e] = expf(v[e] - mx);  // exp(-inf) = 0 for the padding
        den += v[e];
      }
      den = wave_reduce_sum(den);
      const int lab = (int)labels[row];
      lsum += mx + logf(den) - lg[lab];
      if (dlogits) {
        const float inv = 1.0f / den;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
          const int j = lane + 64 * e;
          if (j < V) dlogits[(long long)row * V + j] = (v[e] * inv - (j == lab ? 1.f : 0.f)) * invR;
        }
      }
    }
  } else
  for (int row = wave; row < rows; row += 4) {
    const float* lg = logits + (long long)row * V;
    float mx = -INFINITY;
    for (int j = lane; j < V; j += 64) mx = fmaxf(mx, lg[j]);
    mx = wave_reduce_max(mx);
    float den = 0.f;
    for (int j = lane; j < V; j += 64) den += expf(lg[j] - mx);
    den = wave_reduce_sum(den);
    const int lab = (int)labels[row];
    const float lse = mx + logf(den);
    lsum += lse - lg[lab];
    if (dlogits) {
      const float inv = 1.0f / den;
      for (int j = lane; j < V; j += 64) {
        const float pj = expf(lg[j] - mx) * inv;
        dlogits[(long long)row * V + j] = (pj - (j == lab ? 1.f : 0.f)) * invR;
      }
    }
  }
  if (lane == 0) wsum[wave] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) *loss = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * invR;
}

extern "C" int vs_softmax_xent(const float* logits, const int64_t* labels, float* loss,
                               float* dlogits, int rows, int V, void* stream) {
  VS_CHECK_ARG(logits && labels && loss && rows > 0 && V > 0, "bad args");
  hipLaunchKernelGGL(softmax_xent_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels,
                     loss, dlogits, rows, V);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// softmax -> k largest (descending; ties -> lowest index).  One wave per row.
__global__ __launch_bounds__(64) void softmax_topk_kernel(const float* logits, float* probs_out,
                                                          int64_t* idx_out, int V, int k) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const float* lg = logits + (long long)row * V;
  float mx = -INFINITY;
  for (int j = lane; j < V; j += 64) mx = fmaxf(mx, lg[j]);
  mx = wave_reduce_max(mx);
  float den = 0.f;
  for (int j = lane; j < V; j += 64) den += expf(lg[j] - mx);
  den = wave_reduce_sum(den);
  float prev_v = INFINITY;
  int prev_i = -1;
  for (int t = 0; t < k; ++t) {
    // best (value, index) strictly after (prev_v, prev_i) in (desc value, asc index) order
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = lane; j < V; j += 64) {
      const float v = lg[j];
      const bool after = (v < prev_v) || (v == prev_v && j > prev_i);
      if (after && (v > bv || (v == bv && j < bi))) {
        bv = v;
        bi = j;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      probs_out[(long long)row * k + t] = expf(bv - mx) / den;
      idx_out[(long long)row * k + t] = bi;
    }
    prev_v = bv;
    prev_i = bi;
  }
}

extern "C" int vs_softmax_topk(const float* logits, float* probs_out, int64_t* idx_out, int rows,
                               int V, int k, void* stream) {
  VS_CHECK_ARG(logits && probs_out && idx_out && rows > 0 && V > 0 && k > 0 && k <= V, "bad args");
  hipLaunchKernelGGL(softmax_topk_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, logits,
                     probs_out, idx_out, V, k);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ----------------------------------------------------------------------------
// Adam on a flat fp32 arena (torch.optim.Adam semantics, main_dist.py:50) and the
// fp32 -> bf16 weight cast.
// ----------------------------------------------------------------------------
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long long n, float lr,
                            float b1, float b2, float eps, float bc1, float bc2_sqrt,
                            float grad_scale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * grad_scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}

extern "C" int vs_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                            float beta1, float beta2, float eps, int step, float grad_scale,
                            void* stream) {
  VS_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "bad args");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  long long grid = (n + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, g, m,
                     v, (long long)n, lr, beta1, beta2, eps, bc1, sqrtf(bc2), grad_scale);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// hipGraph-safe variant: the step count lives in device memory (a captured launch would
// otherwise replay the bias correction of the step it was captured at).
__global__ void adam_tick_kernel(int* step) { step[0] += 1; }

// float4-wide; optionally also emits the bf16 kernel copy of the updated parameters (saves the
// separate cast pass over the arena).  G16: the gradients are bf16 (the summed payload of a bf16
// all-reduce); parameters and both moments stay fp32.
template <bool G16>
__global__ void adam_dev_kernel(float* p, const void* gv, float* m, float* v, uint16_t* pb,
                                long long n4, long long n, float lr, float b1, float b2, float eps,
                                const int* step, float grad_scale) {
  const float t = (float)step[0];
  const float bc1 = 1.f - powf(b1, t);
  const float bc2_sqrt = sqrtf(1.f - powf(b2, t));
  const float a = lr / bc1;
  const float* g = (const float*)gv;
  const uint16_t* gh = (const uint16_t*)gv;
  // Two float4 groups per thread and iteration, all eight loads issued before the first use: with one group (four
  // loads, then ~40 dependent VALU instructions incl. a divide and a square root per element, then the stores) the
  // kernel moved its 1.1 GB at 2.4 TB/s.  Same arithmetic per element, same results bit for bit.
  // The gradient is read for the last time and the moments are touched by nobody until the next step's update: non-
  // temporal loads / stores keep this 2.3 GB stream from flushing the parameters (and their bf16 image) it writes --
  // which the next step's first kernels read -- out of the Infinity Cache (profiles/r04_nontemporal.txt).
  typedef __attribute__((ext_vector_type(4))) float nt_f4;
  typedef __attribute__((ext_vector_type(2))) unsigned nt_u2;
  auto ldnt = [](const float* q, long long i) __attribute__((always_inline)) {
    const nt_f4 t = __builtin_nontemporal_load((const nt_f4*)q + i);
    return make_float4(t[0], t[1], t[2], t[3]);
  };
  auto stnt = [](float* q, long long i, float a0, float a1, float a2, float a3) __attribute__((always_inline)) {
    const nt_f4 t = {a0, a1, a2, a3};
    __builtin_nontemporal_store(t, (nt_f4*)q + i);
  };
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += 2 * stride) {
    const long long i1 = i0 + stride;
    const bool two = i1 < n4;
    const long long j1 = two ? i1 : i0;
    float gg[2][4];
    if (G16) {
      const nt_u2 ga_ = __builtin_nontemporal_load((const nt_u2*)gv + i0), gb_ = __builtin_nontemporal_load((const nt_u2*)gv + j1);
      const uint2 ga = make_uint2(ga_[0], ga_[1]), gb = make_uint2(gb_[0], gb_[1]);
      gg[0][0] = __uint_as_float(ga.x << 16) * grad_scale;
      gg[0][1] = __uint_as_float(ga.x & 0xffff0000u) * grad_scale;
      gg[0][2] = __uint_as_float(ga.y << 16) * grad_scale;
      gg[0][3] = __uint_as_float(ga.y & 0xffff0000u) * grad_scale;
      gg[1][0] = __uint_as_float(gb.x << 16) * grad_scale;
      gg[1][1] = __uint_as_float(gb.x & 0xffff0000u) * grad_scale;
      gg[1][2] = __uint_as_float(gb.y << 16) * grad_scale;
      gg[1][3] = __uint_as_float(gb.y & 0xffff0000u) * grad_scale;
    } else {
      const float4 ga = ldnt((const float*)gv, i0), gb = ldnt((const float*)gv, j1);
      gg[0][0] = ga.x * grad_scale; gg[0][1] = ga.y * grad_scale; gg[0][2] = ga.z * grad_scale; gg[0][3] = ga.w * grad_scale;
      gg[1][0] = gb.x * grad_scale; gg[1][1] = gb.y * grad_scale; gg[1][2] = gb.z * grad_scale; gg[1][3] = gb.w * grad_scale;
    }
    const float4 ma = ldnt(m, i0), va = ldnt(v, i0), pa = ((float4*)p)[i0];
    const float4 mb = ldnt(m, j1), vb = ldnt(v, j1), pq = ((float4*)p)[j1];
    float mm[2][4] = {{ma.x, ma.y, ma.z, ma.w}, {mb.x, mb.y, mb.z, mb.w}};
    float vv[2][4] = {{va.x, va.y, va.z, va.w}, {vb.x, vb.y, vb.z, vb.w}};
    float pp[2][4] = {{pa.x, pa.y, pa.z, pa.w}, {pq.x, pq.y, pq.z, pq.w}};
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mm[u][e] = b1 * mm[u][e] + (1.f - b1) * gg[u][e];
        vv[u][e] = b2 * vv[u][e] + (1.f - b2) * gg[u][e] * gg[u][e];
        const float denom = sqrtf(vv[u][e]) / bc2_sqrt + eps;
        pp[u][e] -= a * (mm[u][e] / denom);
      }
    stnt(m, i0, mm[0][0], mm[0][1], mm[0][2], mm[0][3]);
    stnt(v, i0, vv[0][0], vv[0][1], vv[0][2], vv[0][3]);
    ((float4*)p)[i0] = make_float4(pp[0][0], pp[0][1], pp[0][2], pp[0][3]);
    if (pb) ((uint2*)pb)[i0] = make_uint2(pack2_bf16(pp[0][0], pp[0][1]), pack2_bf16(pp[0][2], pp[0][3]));
    if (two) {
      stnt(m, i1, mm[1][0], mm[1][1], mm[1][2], mm[1][3]);
      stnt(v, i1, vv[1][0], vv[1][1], vv[1][2], vv[1][3]);
      ((float4*)p)[i1] = make_float4(pp[1][0], pp[1][1], pp[1][2], pp[1][3]);
      if (pb) ((uint2*)pb)[i1] = make_uint2(pack2_bf16(pp[1][0], pp[1][1]), pack2_bf16(pp[1][2], pp[1][3]));
    }
  }
  // scalar tail (n not a multiple of 4, or unaligned buffers: then n4 == 0 and this is everything)
  for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float gi = (G16 ? bf16_to_f32(gh[i]) : g[i]) * grad_scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float pi = p[i] - a * (mi / denom);
    p[i] = pi;
    if (pb) pb[i] = f32_to_bf16(pi);
  }
}

static int adam_dev_launch(float* p, const void* g, bool g16, float* m, float* v, void* p_bf16, int64_t n,
                           float lr, float beta1, float beta2, float eps, int* step_counter,
                           float grad_scale, void* stream, bool tick = true) {
  const bool aligned = (((uintptr_t)p | (uintptr_t)m | (uintptr_t)v) & 15) == 0 &&
                       (((uintptr_t)g) & (g16 ? 7 : 15)) == 0 && (((uintptr_t)p_bf16) & 7) == 0;
  const long long n4 = aligned ? n / 4 : 0;
  long long grid = ((aligned ? n4 : (long long)n) + 255) / 256;
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  if (tick) hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_counter);
  if (g16)
    hipLaunchKernelGGL(adam_dev_kernel<true>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, g, m,
                       v, (uint16_t*)p_bf16, n4, (long long)n, lr, beta1, beta2, eps,
                       (const int*)step_counter, grad_scale);
  else
    hipLaunchKernelGGL(adam_dev_kernel<false>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, g, m,
                       v, (uint16_t*)p_bf16, n4, (long long)n, lr, beta1, beta2, eps,
                       (const int*)step_counter, grad_scale);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

extern "C" int vs_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                                float beta1, float beta2, float eps, int* step_counter,
                                float grad_scale, void* stream) {
  VS_CHECK_ARG(p && g && m && v && n > 0 && step_counter, "bad args");
  return adam_dev_launch(p, g, false, m, v, nullptr, n, lr, beta1, beta2, eps, step_counter, grad_scale, stream);
}

extern "C" int vs_adam_step_dev_cast(float* p, const float* g, float* m, float* v, void* p_bf16,
                                     int64_t n, float lr, float beta1, float beta2, float eps,
                                     int* step_counter, float grad_scale, void* stream) {
  VS_CHECK_ARG(p && g && m && v && p_bf16 && n > 0 && step_counter, "bad args");
  return adam_dev_launch(p, g, false, m, v, p_bf16, n, lr, beta1, beta2, eps, step_counter, grad_scale, stream);
}

extern "C" int vs_adam_step_dev_cast_g16(float* p, const void* g_bf16, float* m, float* v, void* p_bf16,
                                         int64_t n, float lr, float beta1, float beta2, float eps,
                                         int* step_counter, float grad_scale, void* stream) {
  VS_CHECK_ARG(p && g_bf16 && m && v && p_bf16 && n > 0 && step_counter, "bad args");
  return adam_dev_launch(p, g_bf16, true, m, v, p_bf16, n, lr, beta1, beta2, eps, step_counter, grad_scale,
                         stream);
}

extern "C" int vs_adam_tick(int* step_counter, void* stream) {
  VS_CHECK_ARG(step_counter, "null counter");
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_counter);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

extern "C" int vs_adam_step_dev_range(float* p, const void* g, int g_is_bf16, float* m, float* v, void* p_bf16,
                                      int64_t n, float lr, float beta1, float beta2, float eps,
                                      const int* step_counter, float grad_scale, void* stream) {
  VS_CHECK_ARG(p && g && m && v && n > 0 && step_counter, "bad args");
  return adam_dev_launch(p, g, g_is_bf16 != 0, m, v, p_bf16, n, lr, beta1, beta2, eps, (int*)step_counter,
                         grad_scale, stream, false);
}

__global__ void cast_f32_bf16_kernel(const float* x, uint16_t* y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    y[i] = f32_to_bf16(x[i]);
}

extern "C" int vs_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
  VS_CHECK_ARG(x && y && n > 0, "bad args");
  long long grid = (n + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                     x, (uint16_t*)y, (long long)n);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ------------------------------ error plumbing --------------------------------
#include <stdarg.h>
static thread_local char g_err[512] = "";

void vs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* vs_last_error_string(void) { return g_err; }
extern "C" int vs_version(void) { return 1; }

#include <atomic>
static std::atomic<long long> g_launches{0};
void vs_count_launch(void) { g_launches.fetch_add(1, std::memory_order_relaxed); }
extern "C" int64_t vs_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }
