// Stem convolutions of the SlowFast trunk: Conv3d(3 -> Cout, [kT,7,7], stride [1,2,2],
// pad [kT/2,3,3]) -- slowfast stem_helper via vidsitu_code/mdl_sf_base.py:22 (s1).
//
// Cin = 3 does not fit the 8-channel units of the generic implicit GEMM (padding to 8
// made the gather move 12.6 GB through L2 per step).  Here the input is NDHWC with C padded
// to 4 (8 bytes / pixel) and one kernel ROW (dt, dh) is one k = 32 MFMA step: kw padded 7 -> 8,
// so k = (dw, c) = 8 x 4 and a lane's 8 consecutive k are 2 adjacent pixels = 16 contiguous
// bytes.  Per 8 x 16 output tile the block stages the input patch (kT x 21 x 38 pixels) in
// LDS once; A fragments are ds_read_b128 straight from the patch (stride-2 windows are 16-byte
// aligned because 2*wo + dw0 is even), B fragments come from the whole packed weight
// [Cout][kT][7][8][4], resident in LDS across a persistent loop over tiles.
#include "common.h"

#define ST_TH 8
#define ST_TW 16
#define ST_PH 21                // 2*8 + 5 input rows
#define ST_PW 40                // 2*16 + 5 = 37 input pixels, padded to 40
#define ST_ROWB (ST_PW * 8)     // 320 bytes per patch row
#define ST_FRAMEB (ST_PH * ST_ROWB)
#define ST_TC 4                 // consecutive output frames per work item (frame ring reuse)

struct StemP {
  const uint16_t* x;  // [N][T][H][W][4] bf16
  const uint16_t* w;  // [CP][kT][7][8][4] bf16
  uint16_t* y;        // [N][T][Ho][Wo] rows of y_ld
  const float* scale;
  const float* shift;
  float* stats;       // [tiles][2][Cout]
  int N, T, H, W, Ho, Wo, Cout, kT, y_ld, flags;
  int tilesH, tilesW, tchunks, ntiles;  // ntiles = number of work items
};

template <int NT>  // NT = padded Cout / 16
__global__ __launch_bounds__(256) void stem_conv_kernel(StemP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CP = NT * 16;
  const int K = p.kT * 7 * 32;
  const int wpitch = K * 2 + 16;
  char* wlds = smem;                                   // [CP][wpitch]
  char* patch = smem + ((CP * wpitch + 15) & ~15);     // [kT][21][40] x 8 B
  const int patch_bytes = p.kT * ST_FRAMEB;
  float* E = (float*)(patch + ((patch_bytes + 15) & ~15));  // [128][CP] fp32
  float* statbuf = E + 128 * CP;                        // [2][4][CP]

  const int tid = threadIdx.x;
  // weights: resident for the whole persistent loop
  {
    const int chunks_per_row = K / 8;  // 16-byte chunks
    for (int i = tid; i < CP * chunks_per_row; i += 256) {
      const int r = i / chunks_per_row, c = i - r * chunks_per_row;
      *(u32x4*)(wlds + r * wpitch + c * 16) = *(const u32x4*)(p.w + (long long)r * K + c * 8);
    }
  }
  const int wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lq = lane >> 4;
  const int pT = p.kT >> 1;

  // Work item = (clip, spatial tile, chunk of ST_TC consecutive output frames).  Inside an item
  // the kT input frames live in a ring (slot = frame mod kT): only ONE new frame is staged per
  // output frame after the first, instead of kT.
  auto stage_frame = [&](int n, int ti, int hi0, int wi0) __attribute__((always_inline)) {
    const int slot = ((ti % p.kT) + p.kT) % p.kT;
    char* dst = patch + slot * ST_FRAMEB;
    const bool tin = (unsigned)ti < (unsigned)p.T;
    const uint16_t* src = p.x + (((long long)n * p.T + (tin ? ti : 0)) * p.H) * (long long)p.W * 4;
    uint2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // 21*40 = 840 pixels: 4 per thread, loads issued together
      const int i = tid + u * 256;
      const int r = i / ST_PW, c = i - r * ST_PW;
      const int hi = hi0 + r, wi = wi0 + c;
      v[u] = make_uint2(0u, 0u);
      if (i < ST_PH * ST_PW && tin && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
        v[u] = *(const uint2*)(src + ((long long)hi * p.W + wi) * 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * 256;
      if (i < ST_PH * ST_PW) *(uint2*)(dst + i * 8) = v[u];
    }
  };

  for (int item = blockIdx.x; item < p.ntiles; item += gridDim.x) {
    int t = item;
    const int tc = t % p.tchunks;
    t /= p.tchunks;
    const int tw = t % p.tilesW;
    t /= p.tilesW;
    const int th = t % p.tilesH;
    const int n = t / p.tilesH;
    const int ho0 = th * ST_TH, wo0 = tw * ST_TW;
    const int hi0 = 2 * ho0 - 3, wi0 = 2 * wo0 - 3;
    const int to_beg = tc * ST_TC, to_end = min(p.T, to_beg + ST_TC);
   for (int to = to_beg; to < to_end; ++to) {
    const int tile = ((n * p.T + to) * p.tilesH + th) * p.tilesW + tw;  // stats row
    __syncthreads();  // previous epilogue / weight fill done; ring slot about to be reused
    if (to == to_beg) {
      for (int dt = 0; dt < p.kT; ++dt) stage_frame(n, to - pT + dt, hi0, wi0);
    } else {
      stage_frame(n, to + pT, hi0, wi0);
    }
    __syncthreads();

    f32x4 acc[2][NT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ring slot of frame (to - pT + dt): incremental, no division in the loop
    int slot = (to - pT) % p.kT;
    if (slot < 0) slot += p.kT;
    int ks = 0;
    for (int dt = 0; dt < p.kT; ++dt) {
      const char* fbase = patch + slot * ST_FRAMEB + (2 * lr + 2 * lq) * 8;
      for (int dh = 0; dh < 7; ++dh, ++ks) {
        bf16x8 af[2], bfr[NT];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int hol = wave * 2 + a;  // one output row per 16-position MFMA row tile
          af[a] = *(const bf16x8*)(fbase + (2 * hol + dh) * ST_ROWB);
        }
#pragma unroll
        for (int b = 0; b < NT; ++b)
          bfr[b] = *(const bf16x8*)(wlds + (b * 16 + lr) * wpitch + (ks * 32 + lq * 8) * 2);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < NT; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
      }
      if (++slot == p.kT) slot = 0;
    }

    // ---- epilogue: D[row = lq*4 + r -> wo_l][col = lr -> cout] ----
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int ho = ho0 + wave * 2 + a;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int wo = wo0 + lq * 4 + r;
        if (ho >= p.Ho || wo >= p.Wo) {
#pragma unroll
          for (int b = 0; b < NT; ++b) acc[a][b][r] = 0.f;  // tile tail: keep out of the stats
        }
      }
    }
    if (p.flags & VS_CONV_STATS) {
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[a][b][r];
            s += v;
            q += v * v;
          }
        s += __shfl_xor(s, 16, 64);
        q += __shfl_xor(q, 16, 64);
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        if (lq == 0) {
          statbuf[wave * CP + b * 16 + lr] = s;
          statbuf[4 * CP + wave * CP + b * 16 + lr] = q;
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int col = b * 16 + lr;
      float sc = 1.f, sh = 0.f;
      if ((p.flags & VS_CONV_AFFINE) && col < p.Cout) {
        sc = p.scale[col];
        sh = p.shift[col];
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (wave * 2 + a) * 16 + lq * 4 + r;  // = hol*16 + wo_l
          E[row * CP + col] = acc[a][b][r] * sc + sh;
        }
    }
    __syncthreads();
    if ((p.flags & VS_CONV_STATS) && tid < p.Cout) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s += statbuf[w * CP + tid];
        q += statbuf[4 * CP + w * CP + tid];
      }
      float* dst = p.stats + (long long)tile * 2 * p.Cout;
      dst[tid] = s;
      dst[p.Cout + tid] = q;
    }
    const int cpr = p.Cout >> 3;  // real 16-byte chunks per output row
    for (int idx = tid; idx < 128 * cpr; idx += 256) {
      const int row = idx / cpr, c8 = idx - row * cpr;
      const int ho = ho0 + (row >> 4), wo = wo0 + (row & 15);
      if (ho < p.Ho && wo < p.Wo) {
        float v[8];
        const float4 v0 = *(const float4*)(E + row * CP + c8 * 8);
        const float4 v1 = *(const float4*)(E + row * CP + c8 * 8 + 4);
        v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w;
        v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
        if (p.flags & VS_CONV_RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        const long long pos = (((long long)n * p.T + to) * p.Ho + ho) * p.Wo + wo;
        *(uint4*)(p.y + pos * p.y_ld + c8 * 8) = pack8_bf16(v);
      }
    }
   }  // to
  }    // item
}

// =============================================================================
// Stem with <= 8 output channels (the fast pathway: 3 -> 8, [5,7,7]).  16 MFMA columns for 8 channels wasted
// half of every MFMA and every A-fragment read, and the kernel above is LDS-read bound on this shape (3 x 1 KB
// fragment reads per 2 MFMAs: 134 us of LDS time in a 196 us launch).  Here
//   * the 16 columns are (frame offset j in {0,1}) x (8 channels): one pass over kT + 1 input frames yields TWO
//     consecutive output frames (the weight image in LDS holds W[dt] in columns j = 0 and W[dt - 1] in j = 1):
//     (kT + 1) / (2 kT) of the k-steps;
//   * a wave owns two output rows whose 7-row windows overlap: 9 A fragments feed 14 MFMAs per input frame;
//   * tile = 16 x 16 positions, 8 waves; the two input frames the next pair needs are fetched into registers
//     before this pair's MFMAs and written into the ring slots the pair no longer reads.
// Batch-statistic partial rows keep the 8 x 16 tile numbering of vs_stem_stats_rows (waves 0-3 / 4-7).
// =============================================================================
#define SP_TH 16
#define SP_PH 37                 // 2*16 + 5 input rows
#define SP_FRAMEB (SP_PH * ST_ROWB)
#define SP_TC 8                  // consecutive output frames per work item (4 pairs)

__global__ __launch_bounds__(512) void stem_pair_kernel(StemP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int kT = p.kT, KT1 = kT + 1, pT = kT >> 1;
  const int K = kT * 7 * 32, K2 = KT1 * 7 * 32;
  const int wpitch = K2 * 2 + 16;
  char* wlds = smem;                                                   // [16][wpitch]
  char* patch = smem + ((16 * wpitch + 15) & ~15);                     // [KT1][37][40] x 8 B
  float* E = (float*)(patch + ((KT1 * SP_FRAMEB + 15) & ~15));         // [256][16] fp32
  float* statbuf = E + 256 * 16;                                       // [2][8][16]
  const int tid = threadIdx.x;
  {  // pair-packed weight image: column n = j * 8 + co takes W[co][dt' - j]
    const int cpr = K2 / 8;  // 16-byte chunks per row
    for (int i = tid; i < 16 * cpr; i += 512) {
      const int r = i / cpr, c = i - r * cpr;
      const int ks = c >> 2, sub = c & 3;
      const int dtp = ks / 7, dh = ks - dtp * 7;
      const int j = r >> 3, co = r & 7, dt = dtp - j;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (dt >= 0 && dt < kT && co < p.Cout)
        v = *(const u32x4*)(p.w + (long long)co * K + ((dt * 7 + dh) * 32 + sub * 8));
      *(u32x4*)(wlds + r * wpitch + c * 16) = v;
    }
  }
  const int wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lq = lane >> 4;
  const int tilesH16 = (p.Ho + SP_TH - 1) / SP_TH;

  auto slot_of = [&](int ti) { return ((ti % KT1) + KT1) % KT1; };
  auto frame_load = [&](int n, int ti, int hi0, int wi0, uint2* v) __attribute__((always_inline)) {
    const bool tin = (unsigned)ti < (unsigned)p.T;
    const uint16_t* src = p.x + (((long long)n * p.T + (tin ? ti : 0)) * p.H) * (long long)p.W * 4;
#pragma unroll
    for (int u = 0; u < 3; ++u) {  // 37 * 40 = 1480 pixels: 3 per thread
      const int i = tid + u * 512;
      const int r = i / ST_PW, c = i - r * ST_PW;
      const int hi = hi0 + r, wi = wi0 + c;
      v[u] = make_uint2(0u, 0u);
      if (i < SP_PH * ST_PW && tin && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
        v[u] = *(const uint2*)(src + ((long long)hi * p.W + wi) * 4);
    }
  };
  auto frame_store = [&](int ti, const uint2* v) __attribute__((always_inline)) {
    char* dst = patch + slot_of(ti) * SP_FRAMEB;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = tid + u * 512;
      if (i < SP_PH * ST_PW) *(uint2*)(dst + i * 8) = v[u];
    }
  };

  for (int item = blockIdx.x; item < p.ntiles; item += gridDim.x) {
    int t = item;
    const int tc = t % p.tchunks;
    t /= p.tchunks;
    const int tw = t % p.tilesW;
    t /= p.tilesW;
    const int th = t % tilesH16;
    const int n = t / tilesH16;
    const int ho0 = th * SP_TH, wo0 = tw * ST_TW;
    const int hi0 = 2 * ho0 - 3, wi0 = 2 * wo0 - 3;
    const int to_beg = tc * SP_TC, to_end = min(p.T, to_beg + SP_TC);
    __syncthreads();  // previous item's ring / E are free; the weight image is in place
    for (int f = 0; f < KT1; ++f) {
      uint2 v[3];
      frame_load(n, to_beg - pT + f, hi0, wi0, v);
      frame_store(to_beg - pT + f, v);
    }
    __syncthreads();
    for (int to0 = to_beg; to0 < to_end; to0 += 2) {
      const bool more = to0 + 2 < to_end;
      uint2 nf0[3], nf1[3];
      if (more) {  // the two frames the next pair adds: in flight during this pair's MFMAs
        frame_load(n, to0 - pT + KT1, hi0, wi0, nf0);
        frame_load(n, to0 - pT + KT1 + 1, hi0, wi0, nf1);
      }
      f32x4 acc[2];
      acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      int slot = slot_of(to0 - pT);
      const char* wrow = wlds + lr * wpitch + lq * 16;
      for (int dtp = 0; dtp < KT1; ++dtp) {
        const char* fbase = patch + slot * SP_FRAMEB + (2 * lr + 2 * lq) * 8 + (4 * wave) * ST_ROWB;
        bf16x8 af[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) af[r] = *(const bf16x8*)(fbase + r * ST_ROWB);
#pragma unroll
        for (int dh = 0; dh < 7; ++dh) {
          const bf16x8 bfr = *(const bf16x8*)(wrow + (dtp * 7 + dh) * 64);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dh], bfr, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dh + 2], bfr, acc[1], 0, 0, 0);
        }
        if (++slot == KT1) slot = 0;
      }
      // ---- epilogue: D[row = lq*4 + r -> wo_l][col = lr -> (j, co)] ----
      const int j = lr >> 3, co = lr & 7;
      const bool fok = to0 + j < p.T && co < p.Cout;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int ho = ho0 + wave * 2 + a;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int wo = wo0 + lq * 4 + r;
          if (ho >= p.Ho || wo >= p.Wo || !fok) acc[a][r] = 0.f;  // tile tail: keep out of the stats
        }
      }
      if (p.flags & VS_CONV_STATS) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[a][r];
            s += v;
            q += v * v;
          }
        s += __shfl_xor(s, 16, 64);
        q += __shfl_xor(q, 16, 64);
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        if (lq == 0) {
          statbuf[wave * 16 + lr] = s;
          statbuf[8 * 16 + wave * 16 + lr] = q;
        }
      }
      {
        float sc = 1.f, sh = 0.f;
        if ((p.flags & VS_CONV_AFFINE) && co < p.Cout) {
          sc = p.scale[co];
          sh = p.shift[co];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) E[((wave * 2 + a) * 16 + lq * 4 + r) * 16 + lr] = acc[a][r] * sc + sh;
      }
      __syncthreads();
      if ((p.flags & VS_CONV_STATS) && tid < 32) {
        const int h = tid >> 4, col = tid & 15, jj = col >> 3, cc = col & 7;
        const int th8 = 2 * th + h;
        if (cc < p.Cout && to0 + jj < p.T && th8 < p.tilesH) {
          float s = 0.f, q = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            s += statbuf[(4 * h + w) * 16 + col];
            q += statbuf[8 * 16 + (4 * h + w) * 16 + col];
          }
          const int tile = ((n * p.T + to0 + jj) * p.tilesH + th8) * p.tilesW + tw;  // stats row
          float* dst = p.stats + (long long)tile * 2 * p.Cout;
          dst[cc] = s;
          dst[p.Cout + cc] = q;
        }
      }
      {  // one 16-byte channel vector per (position, frame)
        const int row = tid >> 1, jj = tid & 1;
        const int ho = ho0 + (row >> 4), wo = wo0 + (row & 15);
        if (ho < p.Ho && wo < p.Wo && to0 + jj < p.T) {
          float v[8];
          const float4 v0 = *(const float4*)(E + row * 16 + jj * 8);
          const float4 v1 = *(const float4*)(E + row * 16 + jj * 8 + 4);
          v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w;
          v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
          if (p.flags & VS_CONV_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          const long long pos = (((long long)n * p.T + to0 + jj) * p.Ho + ho) * p.Wo + wo;
          *(uint4*)(p.y + pos * p.y_ld) = pack8_bf16(v);
        }
      }
      __syncthreads();  // every wave is done with the ring slots of the pair's two oldest frames, and with E
      if (more) {
        frame_store(to0 - pT + KT1, nf0);
        frame_store(to0 - pT + KT1 + 1, nf1);
      }
      __syncthreads();
    }
  }
}

// The same pair packing with the WEIGHTS IN REGISTERS: a lane's B fragments of all (kT + 1) * 7 k-steps (16 bytes
// each, 168 VGPRs for kT = 5) are loaded once per persistent block, so the only LDS reads of the main loop are the
// 9 A rows per input frame (54 per pair instead of 96) and the LDS holds nothing but the frame ring (40 KB): two
// 4-wave blocks per CU, tile 8 x 16.  VS_STEM_PAIR=2 selects stem_pair_kernel (weights in LDS) for A/B runs.
template <int KT>
__global__ __launch_bounds__(256) void stem_pair_reg_kernel(StemP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KT1 = KT + 1, pT = KT >> 1, NKS = KT1 * 7;
  constexpr int K = KT * 7 * 32;
  char* patch = smem;                                                  // [KT1][21][40] x 8 B
  float* E = (float*)(patch + ((KT1 * ST_FRAMEB + 15) & ~15));         // [128][16] fp32
  float* statbuf = E + 128 * 16;                                       // [2][4][16]
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lq = lane >> 4;
  const int j = lr >> 3, co = lr & 7;

  bf16x8 bfr[NKS];  // column lr = (j, co): W[co][dt' - j][dh][lq*8 .. +8]
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int dtp = ks / 7, dh = ks - dtp * 7;
    const int dt = dtp - j;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (dt >= 0 && dt < KT && co < p.Cout)
      v = *(const u32x4*)(p.w + (long long)co * K + ((dt * 7 + dh) * 32 + lq * 8));
    bfr[ks] = __builtin_bit_cast(bf16x8, v);
  }

  auto slot_of = [](int ti) { return ((ti % KT1) + KT1) % KT1; };
  auto frame_load = [&](int n, int ti, int hi0, int wi0, uint2* v) __attribute__((always_inline)) {
    const bool tin = (unsigned)ti < (unsigned)p.T;
    const uint16_t* src = p.x + (((long long)n * p.T + (tin ? ti : 0)) * p.H) * (long long)p.W * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // 21 * 40 = 840 pixels: 4 per thread
      const int i = tid + u * 256;
      const int r = i / ST_PW, c = i - r * ST_PW;
      const int hi = hi0 + r, wi = wi0 + c;
      v[u] = make_uint2(0u, 0u);
      if (i < ST_PH * ST_PW && tin && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
        v[u] = *(const uint2*)(src + ((long long)hi * p.W + wi) * 4);
    }
  };
  auto frame_store = [&](int ti, const uint2* v) __attribute__((always_inline)) {
    char* dst = patch + slot_of(ti) * ST_FRAMEB;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * 256;
      if (i < ST_PH * ST_PW) *(uint2*)(dst + i * 8) = v[u];
    }
  };

  for (int item = blockIdx.x; item < p.ntiles; item += gridDim.x) {
    int t = item;
    const int tc = t % p.tchunks;
    t /= p.tchunks;
    const int tw = t % p.tilesW;
    t /= p.tilesW;
    const int th = t % p.tilesH;
    const int n = t / p.tilesH;
    const int ho0 = th * ST_TH, wo0 = tw * ST_TW;
    const int hi0 = 2 * ho0 - 3, wi0 = 2 * wo0 - 3;
    const int to_beg = tc * SP_TC, to_end = min(p.T, to_beg + SP_TC);
    __syncthreads();  // previous item's ring / E are free
    for (int f = 0; f < KT1; ++f) {
      uint2 v[4];
      frame_load(n, to_beg - pT + f, hi0, wi0, v);
      frame_store(to_beg - pT + f, v);
    }
    __syncthreads();
    for (int to0 = to_beg; to0 < to_end; to0 += 2) {
      const bool more = to0 + 2 < to_end;
      uint2 nf0[4], nf1[4];
      if (more) {  // the two frames the next pair adds: in flight during this pair's MFMAs
        frame_load(n, to0 - pT + KT1, hi0, wi0, nf0);
        frame_load(n, to0 - pT + KT1 + 1, hi0, wi0, nf1);
      }
      f32x4 acc[2];
      acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      int slot = slot_of(to0 - pT);
#pragma unroll
      for (int dtp = 0; dtp < KT1; ++dtp) {
        const char* fbase = patch + slot * ST_FRAMEB + (2 * lr + 2 * lq) * 8 + (4 * wave) * ST_ROWB;
        bf16x8 af[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) af[r] = *(const bf16x8*)(fbase + r * ST_ROWB);
#pragma unroll
        for (int dh = 0; dh < 7; ++dh) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dh], bfr[dtp * 7 + dh], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dh + 2], bfr[dtp * 7 + dh], acc[1], 0, 0, 0);
        }
        if (++slot == KT1) slot = 0;
      }
      // ---- epilogue: D[row = lq*4 + r -> wo_l][col = lr -> (j, co)] ----
      const bool fok = to0 + j < p.T && co < p.Cout;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int ho = ho0 + wave * 2 + a;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int wo = wo0 + lq * 4 + r;
          if (ho >= p.Ho || wo >= p.Wo || !fok) acc[a][r] = 0.f;  // tile tail: keep out of the stats
        }
      }
      if (p.flags & VS_CONV_STATS) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[a][r];
            s += v;
            q += v * v;
          }
        s += __shfl_xor(s, 16, 64);
        q += __shfl_xor(q, 16, 64);
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        if (lq == 0) {
          statbuf[wave * 16 + lr] = s;
          statbuf[4 * 16 + wave * 16 + lr] = q;
        }
      }
      {
        float sc = 1.f, sh = 0.f;
        if ((p.flags & VS_CONV_AFFINE) && co < p.Cout) {
          sc = p.scale[co];
          sh = p.shift[co];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) E[((wave * 2 + a) * 16 + lq * 4 + r) * 16 + lr] = acc[a][r] * sc + sh;
      }
      __syncthreads();
      if ((p.flags & VS_CONV_STATS) && tid < 16) {
        const int jj = tid >> 3, cc = tid & 7;
        if (cc < p.Cout && to0 + jj < p.T) {
          float s = 0.f, q = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            s += statbuf[w * 16 + tid];
            q += statbuf[4 * 16 + w * 16 + tid];
          }
          const int tile = ((n * p.T + to0 + jj) * p.tilesH + th) * p.tilesW + tw;  // stats row
          float* dst = p.stats + (long long)tile * 2 * p.Cout;
          dst[cc] = s;
          dst[p.Cout + cc] = q;
        }
      }
      {  // one 16-byte channel vector per (position, frame)
        const int row = tid >> 1, jj = tid & 1;
        const int ho = ho0 + (row >> 4), wo = wo0 + (row & 15);
        if (ho < p.Ho && wo < p.Wo && to0 + jj < p.T) {
          float v[8];
          const float4 v0 = *(const float4*)(E + row * 16 + jj * 8);
          const float4 v1 = *(const float4*)(E + row * 16 + jj * 8 + 4);
          v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w;
          v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
          if (p.flags & VS_CONV_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          const long long pos = (((long long)n * p.T + to0 + jj) * p.Ho + ho) * p.Wo + wo;
          *(uint4*)(p.y + pos * p.y_ld) = pack8_bf16(v);
        }
      }
      __syncthreads();  // every wave is done with the ring slots of the pair's two oldest frames, and with E
      if (more) {
        frame_store(to0 - pT + KT1, nf0);
        frame_store(to0 - pT + KT1 + 1, nf1);
      }
      __syncthreads();
    }
  }
}

static size_t stem_pair_smem(int kT) {
  const int K2 = (kT + 1) * 7 * 32;
  const size_t w = ((size_t)16 * (K2 * 2 + 16) + 15) & ~(size_t)15;
  const size_t patch = ((size_t)(kT + 1) * SP_FRAMEB + 15) & ~(size_t)15;
  return w + patch + (size_t)256 * 16 * 4 + (size_t)2 * 8 * 16 * 4;
}

static size_t stem_smem(int CP, int kT) {
  const int K = kT * 7 * 32;
  const size_t w = ((size_t)CP * (K * 2 + 16) + 15) & ~(size_t)15;
  const size_t patch = ((size_t)kT * ST_FRAMEB + 15) & ~(size_t)15;
  return w + patch + (size_t)128 * CP * 4 + (size_t)2 * 4 * CP * 4;
}

extern "C" int vs_stem_stats_rows(int N, int T, int H, int W) {
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  return N * T * ((Ho + ST_TH - 1) / ST_TH) * ((Wo + ST_TW - 1) / ST_TW);
}

extern "C" int vs_stem_conv_fwd(const void* x4, const void* wp, void* y, int N, int T, int H, int W,
                                int Cout, int kT, int y_ld, int flags, const float* scale,
                                const float* shift, float* stats_partial, void* stream) {
  VS_CHECK_ARG(x4 && wp && y, "null tensor");
  VS_CHECK_ARG(Cout % 8 == 0 && Cout <= 64 && y_ld % 8 == 0 && y_ld >= Cout, "Cout in {8..64}, pitch % 8");
  VS_CHECK_ARG(kT == 1 || kT == 3 || kT == 5, "kT in {1,3,5}");
  VS_CHECK_ARG(!(flags & VS_CONV_AFFINE) || (scale && shift), "AFFINE needs scale/shift");
  VS_CHECK_ARG(!(flags & VS_CONV_STATS) || stats_partial, "STATS needs stats_partial");
  VS_CHECK_ARG(!(flags & VS_CONV_RESIDUAL), "no residual on a stem");
  StemP p;
  p.x = (const uint16_t*)x4;
  p.w = (const uint16_t*)wp;
  p.y = (uint16_t*)y;
  p.scale = scale;
  p.shift = shift;
  p.stats = stats_partial;
  p.N = N; p.T = T; p.H = H; p.W = W;
  p.Ho = (H + 6 - 7) / 2 + 1;
  p.Wo = (W + 6 - 7) / 2 + 1;
  p.Cout = Cout; p.kT = kT; p.y_ld = y_ld; p.flags = flags;
  p.tilesH = (p.Ho + ST_TH - 1) / ST_TH;
  p.tilesW = (p.Wo + ST_TW - 1) / ST_TW;
  p.tchunks = (T + ST_TC - 1) / ST_TC;
  p.ntiles = N * p.tilesH * p.tilesW * p.tchunks;
  const int CP = (Cout + 15) / 16 * 16;
  hipStream_t st = (hipStream_t)stream;
  static const int pair_on = [] { const char* e = getenv("VS_STEM_PAIR"); return e ? atoi(e) : 1; }();
  if (pair_on == 1 && Cout == 8 && kT >= 3 && T >= 2) {  // two output frames per pass, weights in registers
    p.tchunks = (T + SP_TC - 1) / SP_TC;
    p.ntiles = N * p.tilesH * p.tilesW * p.tchunks;
    const int grid2 = p.ntiles < 512 ? p.ntiles : 512;  // two persistent 4-wave blocks per CU
    const size_t smem2 = (((size_t)(kT + 1) * ST_FRAMEB + 15) & ~(size_t)15) + (size_t)128 * 16 * 4 + 2 * 4 * 16 * 4;
    if (kT == 5) hipLaunchKernelGGL(stem_pair_reg_kernel<5>, dim3(grid2), dim3(256), smem2, st, p);
    else hipLaunchKernelGGL(stem_pair_reg_kernel<3>, dim3(grid2), dim3(256), smem2, st, p);
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  if (pair_on && Cout == 8 && kT >= 3 && T >= 2) {  // two output frames per pass (fast-pathway stem)
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute((const void*)stem_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_done = true;
    }
    p.tchunks = (T + SP_TC - 1) / SP_TC;
    p.ntiles = N * ((p.Ho + SP_TH - 1) / SP_TH) * p.tilesW * p.tchunks;
    const int grid2 = p.ntiles < 256 ? p.ntiles : 256;  // one persistent 8-wave block per CU
    hipLaunchKernelGGL(stem_pair_kernel, dim3(grid2), dim3(512), stem_pair_smem(kT), st, p);
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  const size_t smem = stem_smem(CP, kT);
  int grid = p.ntiles < 512 ? p.ntiles : 512;  // 2 persistent blocks per CU
#define LAUNCH_STEM(NT_)                                                                         \
  do {                                                                                           \
    static bool attr_done = false;                                                               \
    if (!attr_done) {                                                                            \
      (void)hipFuncSetAttribute((const void*)stem_conv_kernel<NT_>,                              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);         \
      attr_done = true;                                                                          \
    }                                                                                            \
    hipLaunchKernelGGL((stem_conv_kernel<NT_>), dim3(grid), dim3(256), smem, st, p);             \
  } while (0)
  switch (CP / 16) {
    case 1: LAUNCH_STEM(1); break;
    case 2: LAUNCH_STEM(2); break;
    case 3: LAUNCH_STEM(3); break;
    default: LAUNCH_STEM(4); break;
  }
#undef LAUNCH_STEM
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// =============================================================================
// Stem weight gradient.  dW[co][dt][dh][dw][c] = sum_pos dY[pos][co] * X[pos @ tap][c]
// with the same frame-ring patch in LDS.  MFMA K = 32 output positions (two output rows of a
// tile); A = dY^T from an LDS image [m-tile][128 pos][16 couts] read with ds_read_b64_tr_b16;
// B = 16 consecutive (dw, c) elements = 4 pixels of one kernel row, read with the SAME
// transposing instruction straight from the raw patch (each position supplies its own
// 8-byte pixel address).  N-tiles (dt, dh, half-row) are dealt round-robin to the 4 waves;
// accumulators persist over all the block's work items; one fp32 slab per block, then the
// fixed-order slab reduce.  Output layout = the packed weight's: [Cout][kT][7][8][4].
// =============================================================================
typedef __attribute__((ext_vector_type(4))) short st_s16x4;
typedef __attribute__((address_space(3))) st_s16x4 st_lds_s16x4;
typedef __attribute__((ext_vector_type(8))) short st_s16x8;

__device__ __forceinline__ bf16x8 st_tr_pair(const char* p0, const char* p1) {
  const st_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((st_lds_s16x4*)p0);
  const st_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((st_lds_s16x4*)p1);
  const st_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

struct StemWgP {
  const uint16_t* x;   // [N][T][H][W][4]
  const uint16_t* dy;  // [N][T][Ho][Wo] rows of dy_ld
  float* slabs;        // [grid][Cout][Kpad]
  int N, T, H, W, Ho, Wo, Cout, kT, dy_ld;
  int tilesH, tilesW, tchunks, nitems;
};

template <int MT, int NTW>  // MT = padded Cout / 16 ; NTW = n-tiles per wave (ceil(kT*14 / 4))
__global__ __launch_bounds__(256) void stem_wgrad_kernel(StemWgP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch = smem;                                                 // [kT][21][40] x 8 B
  char* dyl = smem + ((p.kT * ST_FRAMEB + 15) & ~15);                 // [MT][128][16] bf16
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p4 = li & 3;
  const int pT = p.kT >> 1;
  const int ntiles_n = p.kT * 14;

  f32x4 acc[MT][NTW];
  int nt_dt[NTW], nt_off[NTW];  // per n-tile (dt, dh, half): frame index and byte offset in a frame
#pragma unroll
  for (int b = 0; b < NTW; ++b) {
#pragma unroll
    for (int a = 0; a < MT; ++a) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nt = wave + 4 * b;
    const int half = nt & 1, row = nt >> 1;
    nt_dt[b] = row / 7;
    nt_off[b] = (row - nt_dt[b] * 7) * ST_ROWB + (half * 4 + p4) * 8;
  }

  auto stage_frame = [&](int n, int ti, int hi0, int wi0) __attribute__((always_inline)) {
    const int slot = ((ti % p.kT) + p.kT) % p.kT;
    char* dst = patch + slot * ST_FRAMEB;
    const bool tin = (unsigned)ti < (unsigned)p.T;
    const uint16_t* src = p.x + (((long long)n * p.T + (tin ? ti : 0)) * p.H) * (long long)p.W * 4;
    uint2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * 256;
      const int r = i / ST_PW, c = i - r * ST_PW;
      const int hi = hi0 + r, wi = wi0 + c;
      v[u] = make_uint2(0u, 0u);
      if (i < ST_PH * ST_PW && tin && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
        v[u] = *(const uint2*)(src + ((long long)hi * p.W + wi) * 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * 256;
      if (i < ST_PH * ST_PW) *(uint2*)(dst + i * 8) = v[u];
    }
  };

  for (int item = blockIdx.x; item < p.nitems; item += gridDim.x) {
    int t = item;
    const int tc = t % p.tchunks;
    t /= p.tchunks;
    const int tw = t % p.tilesW;
    t /= p.tilesW;
    const int th = t % p.tilesH;
    const int n = t / p.tilesH;
    const int ho0 = th * ST_TH, wo0 = tw * ST_TW;
    const int hi0 = 2 * ho0 - 3, wi0 = 2 * wo0 - 3;
    const int to_beg = tc * ST_TC, to_end = min(p.T, to_beg + ST_TC);
    for (int to = to_beg; to < to_end; ++to) {
      __syncthreads();  // previous frame's readers are done
      if (to == to_beg) {
        for (int dt = 0; dt < p.kT; ++dt) stage_frame(n, to - pT + dt, hi0, wi0);
      } else {
        stage_frame(n, to + pT, hi0, wi0);
      }
      // dY tile -> [mt][pos][16] (zeros outside the image and beyond Cout)
      for (int i = tid; i < MT * 128 * 2; i += 256) {
        const int h8 = i & 1;            // which 8-channel half of the 16-wide m-tile
        const int pos = (i >> 1) & 127;
        const int mt = i >> 8;
        const int ho = ho0 + (pos >> 4), wo = wo0 + (pos & 15);
        const int c = mt * 16 + h8 * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ho < p.Ho && wo < p.Wo && c < p.Cout)
          v = *(const u32x4*)(p.dy + ((((long long)n * p.T + to) * p.Ho + ho) * p.Wo + wo) * p.dy_ld + c);
        *(u32x4*)(dyl + (mt * 128 + pos) * 32 + h8 * 16) = v;
      }
      __syncthreads();
      int slot0 = (to - pT) % p.kT;
      if (slot0 < 0) slot0 += p.kT;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        // this lane supplies position k = ks*32 + 8g + q (and k + 4)
        const int k = ks * 32 + 8 * g + q;
        bf16x8 af[MT];
#pragma unroll
        for (int a = 0; a < MT; ++a) {
          const char* ptr = dyl + (a * 128 + k) * 32 + p4 * 8;
          af[a] = st_tr_pair(ptr, ptr + 4 * 32);
        }
        const int rowk = (2 * (k >> 4)) * ST_ROWB + (2 * (k & 15)) * 8;  // patch offset of position k
#pragma unroll
        for (int b = 0; b < NTW; ++b) {
          if (wave + 4 * b < ntiles_n) {
            int sl = slot0 + nt_dt[b];
            if (sl >= p.kT) sl -= p.kT;
            const char* ptr = patch + sl * ST_FRAMEB + nt_off[b] + rowk;
            const bf16x8 bf = st_tr_pair(ptr, ptr + 4 * 16);  // +4 positions = +64 bytes
#pragma unroll
            for (int a = 0; a < MT; ++a)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf, acc[a][b], 0, 0, 0);
          }
        }
      }
    }
  }
  // D[row = g*4 + r -> cout][col = li -> k' column]
  const int Kpad = p.kT * 7 * 32;
  float* dst = p.slabs + (long long)blockIdx.x * p.Cout * Kpad;
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NTW; ++b) {
      const int nt = wave + 4 * b;
      if (nt < ntiles_n) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = a * 16 + g * 4 + r;
          if (co < p.Cout) dst[(long long)co * Kpad + nt * 16 + li] = acc[a][b][r];
        }
      }
    }
}

// Weight gradient of a stem with <= 8 output channels: the pair packing of stem_pair_kernel applied to dW.  The 16
// MFMA rows are (frame offset j) x (8 channels): the dY image holds output frames t0 and t0 + 1 side by side, one
// pass over the kT + 1 input frames of the pair feeds (kT + 1) * 14 n-tiles, and the accumulator of n-tile
// (dt', dh, half) carries dW[.][dt'] in rows j = 0 and dW[.][dt' - 1] in rows j = 1 -- written to two slabs per
// block, summed by the fixed-order slab reduce.  The next pair's two input frames and its dY image are fetched
// into registers before this pair's MFMAs (the kernel above waits out one memory latency per output frame).
template <int KT>
__global__ __launch_bounds__(256) void stem_wgrad_pair_kernel(StemWgP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KT1 = KT + 1, pT = KT >> 1;
  constexpr int NTN = KT1 * 14, NTW = (NTN + 3) / 4;
  char* patch = smem;                                           // [KT1][21][40] x 8 B
  char* dyl = smem + ((KT1 * ST_FRAMEB + 15) & ~15);            // [128 pos][(j, co)] bf16
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p4 = li & 3;

  f32x4 acc[NTW];
#pragma unroll
  for (int b = 0; b < NTW; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto slot_of = [](int ti) { return ((ti % KT1) + KT1) % KT1; };
  auto frame_load = [&](int n, int ti, int hi0, int wi0, uint2* v) __attribute__((always_inline)) {
    const bool tin = (unsigned)ti < (unsigned)p.T;
    const uint16_t* src = p.x + (((long long)n * p.T + (tin ? ti : 0)) * p.H) * (long long)p.W * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * 256;
      const int r = i / ST_PW, c = i - r * ST_PW;
      const int hi = hi0 + r, wi = wi0 + c;
      v[u] = make_uint2(0u, 0u);
      if (i < ST_PH * ST_PW && tin && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
        v[u] = *(const uint2*)(src + ((long long)hi * p.W + wi) * 4);
    }
  };
  auto frame_store = [&](int ti, const uint2* v) __attribute__((always_inline)) {
    char* dst = patch + slot_of(ti) * ST_FRAMEB;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * 256;
      if (i < ST_PH * ST_PW) *(uint2*)(dst + i * 8) = v[u];
    }
  };
  // thread -> (position, frame of the pair): 8 channels = 16 bytes of dY (zeros outside the image / the clip)
  auto dy_load = [&](int n, int to0, int ho0, int wo0) __attribute__((always_inline)) {
    const int jj = tid & 1, pos = tid >> 1;
    const int ho = ho0 + (pos >> 4), wo = wo0 + (pos & 15);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ho < p.Ho && wo < p.Wo && to0 + jj < p.T)
      v = *(const u32x4*)(p.dy + ((((long long)n * p.T + to0 + jj) * p.Ho + ho) * p.Wo + wo) * p.dy_ld);
    return v;
  };
  auto dy_store = [&](const u32x4& v) __attribute__((always_inline)) {
    *(u32x4*)(dyl + (tid >> 1) * 32 + (tid & 1) * 16) = v;
  };

  for (int item = blockIdx.x; item < p.nitems; item += gridDim.x) {
    int t = item;
    const int tc = t % p.tchunks;
    t /= p.tchunks;
    const int tw = t % p.tilesW;
    t /= p.tilesW;
    const int th = t % p.tilesH;
    const int n = t / p.tilesH;
    const int ho0 = th * ST_TH, wo0 = tw * ST_TW;
    const int hi0 = 2 * ho0 - 3, wi0 = 2 * wo0 - 3;
    const int to_beg = tc * SP_TC, to_end = min(p.T, to_beg + SP_TC);
    __syncthreads();  // the previous item's readers are done
    for (int f = 0; f < KT1; ++f) {
      uint2 v[4];
      frame_load(n, to_beg - pT + f, hi0, wi0, v);
      frame_store(to_beg - pT + f, v);
    }
    dy_store(dy_load(n, to_beg, ho0, wo0));
    __syncthreads();
    for (int to0 = to_beg; to0 < to_end; to0 += 2) {
      const bool more = to0 + 2 < to_end;
      uint2 nf0[4], nf1[4];
      u32x4 ndy = {0u, 0u, 0u, 0u};
      if (more) {
        frame_load(n, to0 - pT + KT1, hi0, wi0, nf0);
        frame_load(n, to0 - pT + KT1 + 1, hi0, wi0, nf1);
        ndy = dy_load(n, to0 + 2, ho0, wo0);
      }
      const int slot0 = slot_of(to0 - pT);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int k = ks * 32 + 8 * g + q;  // this lane supplies position k (and k + 4)
        const char* aptr = dyl + k * 32 + p4 * 8;
        const bf16x8 af = st_tr_pair(aptr, aptr + 4 * 32);
        const char* bbase = patch + (2 * (k >> 4)) * ST_ROWB + (2 * (k & 15)) * 8 + p4 * 8;
#pragma unroll
        for (int b = 0; b < NTW; ++b) {
          const int nt = wave + 4 * b;  // wave-uniform
          if (nt < NTN) {
            const int half = nt & 1, row = nt >> 1;
            const int dtp = row / 7, dh = row - dtp * 7;
            int sl = slot0 + dtp;
            if (sl >= KT1) sl -= KT1;
            const char* ptr = bbase + sl * ST_FRAMEB + dh * ST_ROWB + half * 32;
            const bf16x8 bf = st_tr_pair(ptr, ptr + 4 * 16);  // +4 positions = +64 bytes
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[b], 0, 0, 0);
          }
        }
      }
      __syncthreads();  // the pair's two oldest frames and the dY image are free
      if (more) {
        frame_store(to0 - pT + KT1, nf0);
        frame_store(to0 - pT + KT1 + 1, nf1);
        dy_store(ndy);
      }
      __syncthreads();
    }
  }
  // D[row = g*4 + r -> (j, co)][col = li]: rows j of n-tile (dt', dh, half) are dW[co][dt' - j][dh][half*16 + li]
  const int Kpad = KT * 7 * 32;
#pragma unroll
  for (int b = 0; b < NTW; ++b) {
    const int nt = wave + 4 * b;
    if (nt < NTN) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = g * 4 + r, j = m >> 3, co = m & 7;
        const int ntd = nt - 14 * j;  // the n-tile of (dt' - j, dh, half)
        if (ntd >= 0 && ntd < KT * 14 && co < p.Cout) {
          float* dst = p.slabs + ((long long)blockIdx.x * 2 + j) * p.Cout * Kpad;
          dst[(long long)co * Kpad + ntd * 16 + li] = acc[b][r];
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void stem_slab_reduce_kernel(const float* slabs, float* dw,
                                                              long long n, int S) {
  __shared__ float4 part[16][17];
  const long long n4 = n >> 2;
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long long i = (long long)blockIdx.x * 16 + col;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const int per = (S + 15) / 16;
    const int s0 = sl * per, s1 = min(S, s0 + per);
    for (int s = s0; s < s1; ++s) {
      const float4 v = *(const float4*)(slabs + (long long)s * n + i * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  part[sl][col] = acc;
  __syncthreads();
  if (sl == 0 && i < n4) {
    float4 t = part[0][col];
    for (int k = 1; k < 16; ++k) {
      const float4 v = part[k][col];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *(float4*)(dw + i * 4) = t;
  }
}

static int stem_wg_grid(int nitems) { return nitems < 512 ? nitems : 512; }

static bool stem_wg_pair(int T, int Cout, int kT) {
  static const int on = [] { const char* e = getenv("VS_STEM_PAIR"); return e ? atoi(e) : 1; }();
  return on && Cout == 8 && (kT == 3 || kT == 5) && T >= 2;
}

extern "C" size_t vs_stem_wgrad_workspace_bytes(int N, int T, int H, int W, int Cout, int kT) {
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const bool pair = stem_wg_pair(T, Cout, kT);
  const int tc = pair ? SP_TC : ST_TC;
  const int nitems = N * ((Ho + ST_TH - 1) / ST_TH) * ((Wo + ST_TW - 1) / ST_TW) * ((T + tc - 1) / tc);
  return (size_t)stem_wg_grid(nitems) * (pair ? 2 : 1) * Cout * kT * 7 * 32 * sizeof(float);
}

extern "C" int vs_stem_conv_wgrad(const void* dy, const void* x4, float* dwp, int N, int T, int H,
                                  int W, int Cout, int kT, int dy_ld, void* workspace,
                                  size_t ws_bytes, void* stream) {
  VS_CHECK_ARG(dy && x4 && dwp && workspace, "null tensor");
  VS_CHECK_ARG(Cout % 8 == 0 && Cout <= 64 && dy_ld % 8 == 0 && dy_ld >= Cout, "Cout in {8..64}, pitch % 8");
  VS_CHECK_ARG(kT == 1 || kT == 3 || kT == 5, "kT in {1,3,5}");
  StemWgP p;
  p.x = (const uint16_t*)x4;
  p.dy = (const uint16_t*)dy;
  p.slabs = (float*)workspace;
  p.N = N; p.T = T; p.H = H; p.W = W;
  p.Ho = (H + 6 - 7) / 2 + 1;
  p.Wo = (W + 6 - 7) / 2 + 1;
  p.Cout = Cout; p.kT = kT; p.dy_ld = dy_ld;
  p.tilesH = (p.Ho + ST_TH - 1) / ST_TH;
  p.tilesW = (p.Wo + ST_TW - 1) / ST_TW;
  p.tchunks = (T + ST_TC - 1) / ST_TC;
  p.nitems = N * p.tilesH * p.tilesW * p.tchunks;
  const int grid = stem_wg_grid(p.nitems);
  if (ws_bytes < vs_stem_wgrad_workspace_bytes(N, T, H, W, Cout, kT)) {
    vs_set_error("vs_stem_conv_wgrad: workspace too small");
    return VS_ERR_WORKSPACE;
  }
  const int MTv = (Cout + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  if (stem_wg_pair(T, Cout, kT)) {  // two output frames per pass (fast-pathway stem)
    p.tchunks = (T + SP_TC - 1) / SP_TC;
    p.nitems = N * p.tilesH * p.tilesW * p.tchunks;
    const int grid2 = stem_wg_grid(p.nitems);
    const size_t smem2 = (((size_t)(kT + 1) * ST_FRAMEB + 15) & ~(size_t)15) + (size_t)128 * 32;
    if (kT == 5) hipLaunchKernelGGL(stem_wgrad_pair_kernel<5>, dim3(grid2), dim3(256), smem2, st, p);
    else hipLaunchKernelGGL(stem_wgrad_pair_kernel<3>, dim3(grid2), dim3(256), smem2, st, p);
    VS_CHECK_LAUNCH();
    const long long n2 = (long long)Cout * kT * 7 * 32;
    hipLaunchKernelGGL(stem_slab_reduce_kernel, dim3((unsigned)((n2 / 4 + 15) / 16)), dim3(256), 0, st,
                       (const float*)workspace, dwp, n2, 2 * grid2);
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  const size_t smem = (((size_t)kT * ST_FRAMEB + 15) & ~(size_t)15) + (size_t)MTv * 128 * 32;
  const int ntw = (kT * 14 + 3) / 4;  // 4 (kT=1), 11 (kT=3), 18 (kT=5)
#define LAUNCH_SWG(MT_, NTW_)                                                                    \
  hipLaunchKernelGGL((stem_wgrad_kernel<MT_, NTW_>), dim3(grid), dim3(256), smem, st, p)
  if (MTv == 1 && ntw == 18) LAUNCH_SWG(1, 18);
  else if (MTv == 1 && ntw == 11) LAUNCH_SWG(1, 11);
  else if (MTv == 1) LAUNCH_SWG(1, 4);
  else if (MTv == 2 && ntw == 18) LAUNCH_SWG(2, 18);
  else if (MTv == 2 && ntw == 11) LAUNCH_SWG(2, 11);
  else if (MTv == 2) LAUNCH_SWG(2, 4);
  else if (MTv == 4 && ntw == 4) LAUNCH_SWG(4, 4);
  else if (MTv == 4 && ntw == 11) LAUNCH_SWG(4, 11);
  else if (MTv == 3 && ntw == 4) LAUNCH_SWG(3, 4);
  else {
    vs_set_error("vs_stem_conv_wgrad: unsupported (Cout, kT) = (%d, %d)", Cout, kT);
    return VS_ERR_UNSUPPORTED;
  }
#undef LAUNCH_SWG
  VS_CHECK_LAUNCH();
  const long long n = (long long)Cout * kT * 7 * 32;
  hipLaunchKernelGGL(stem_slab_reduce_kernel, dim3((unsigned)((n / 4 + 15) / 16)), dim3(256), 0, st,
                     (const float*)workspace, dwp, n, grid);
  VS_CHECK_LAUNCH();
  return VS_OK;
}
