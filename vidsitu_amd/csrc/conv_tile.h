// Shared between the convolution kernels (conv_igemm.hip, conv_halo.hip): the launch parameter block and the
// tile epilogue.  gfx950 only.
#pragma once
#include "common.h"

struct ConvP {
  const uint16_t* x;
  const uint16_t* w;
  uint16_t* y;
  const float* scale;
  const float* shift;
  const uint16_t* res;
  float* stats;
  int M, Ncols, K, Cg, g_ld;
  int Rt, Rh, Rw;  // row space per clip
  int Gt, Gh, Gw;  // gathered tensor dims per clip
  int kT, kH, kW;
  int mulT, mulH, mulW;
  int offT, offH, offW;
  int tmul;
  int shT, shH, shW;  // log2(stride) for the transposed gather
  int y_ld, res_ld, flags;
  int tilesM, tilesN;
  unsigned x_bytes, w_bytes;  // extents for the buffer resources (FAST path)
  int splitK;                 // > 1: blocks (tile, s) write fp32 partial tiles to `slab`
  float* slab;                // [splitK][M][Ncols]
  // In-launch split-K (ring launches): slab = [tile][splitK][fragment][256 threads] float4 partial accumulators and
  // sk_cnt one arrival counter per tile (zero between launches: the last arriver clears it).  NULL: the slab plan.
  int* sk_cnt;
  // MODE 2 (dgrad of a strided conv), stride classes: rows whose coordinate (r + pad) has the same
  // residues mod (sT, sH, sW) use the same subset of taps (dd == residue mod s); tiles never mix
  // classes, so a tile walks only ITS taps -- 9/4 instead of 9 for a 3x3 stride-(1,2,2) conv,
  // 1/4 of the rows (no tap at all) for a 1x1 stride-2 one.  ncls = 0: classes off.
  // M-tiles are numbered class-interleaved: logical tile u = (i, class u % ncls), i = u / ncls, valid while
  // i < cls_tiles[class] (slots beyond a smaller class's last tile exit at once).  A class-major numbering put
  // each class on its own pair of XCDs under the XCD-contiguous block order -- a 1x1 stride-2 dgrad (one class of
  // four has taps) then ran on 2 of the 8 XCDs.
  // An accumulating dgrad (residual == output) launches only the classes that have taps: nslots < ncls, cls_ids
  // lists them (a 1x1 stride-2 shortcut: one class of four).
  int ncls, nslots;
  int cls_tiles[16];          // M-tiles of each class
  int cls_ids[16];            // slot -> class
  // VS_CONV_BNBWD (dgrad whose output is the gradient dz behind a BN + ReLU unit): the epilogue also
  // emits that BN's backward partial sums per M-tile, stats[tm][0][c] = sum g, [1][c] = sum g * xhat with
  // xhat = (bny - mean) * invstd and g = dz where gamma * xhat + beta > 0, else 0 -- what
  // bn_bwd_reduce_kernel<2> computes in a pass of its own over dz and bny
  const uint16_t* bny;
  const float *bn_mean, *bn_invstd, *bn_gamma, *bn_beta;
  const uint8_t* bn_bits;  // the unit's ReLU mask as bits [rows][Ncols/8] (units with a residual input:
                           // the RESIDUAL epilogue), NULL: mask recomputed from gamma / beta
  int bny_ld;
  // RESIDUAL with a mask: the residual operand is an UNMASKED gradient dz and res_bits the ReLU mask of the
  // unit it belongs to (bits [rows][Ncols/8]); the epilogue adds dz where the bit is set.  Spares the
  // BN-backward apply kernel of a bottleneck's last unit the write of its masked copy of dz (`dres`).
  const uint8_t* res_bits;
  int dense;  // pointwise, unit stride: row m is position m of the gathered tensor (no row decode)
  int nclips;  // M / (Rt * Rh * Rw)
  // Order of the reduction (gathering launches with > 1 tap and Cg % 64 == 0): 0 = tap-major (k = tap * Cg + c, the
  // weights' own order), 1 = chunk-major (k-tile kt = 64-channel chunk kt / taps of tap kt % taps).  The taps of a
  // position re-read (nearly) the same activation rows: tap-major puts Cg / 64 k-tiles -- on the wide layers more
  // than an XCD's 4 MiB of L2 worth of streaming -- between two reads of a row, chunk-major one or two.
  int korder;
  // VS_CONV_BNBWD with RESIDUAL, second unit: a ResBlock's shortcut unit receives the same masked gradient as its c
  // unit (one sum(g), two sum(g * xhat)): stats2[tm][0][c] = sum g, [1][c] = sum g * (bny2 - mean2) * invstd2
  const uint16_t* bny2;
  const float *bn_mean2, *bn_invstd2;
  float* stats2;
  int bny2_ld;
  // Apply on load (train, the b -> c edge of a bottleneck): x is the PRODUCER unit's raw convolution output and the
  // operand is relu(x * in_scale[c] + in_shift[c]) rounded to bf16 -- what vs_bn_apply would have written, bit for
  // bit, formed on the A fragments after their LDS read.  NULL: x is used as it is.
  const float* in_scale;
  const float* in_shift;
#ifdef VS_STAMP
  unsigned long long* stamp;  // diagnostic build only (tools/launch_anatomy.py): [block][8] s_memrealtime stamps, or NULL
#endif
};
#define VS_CONV_BNBWD (1 << 20)

// ---- launch anatomy (diagnostic build, -DVS_STAMP; the shipped library holds none of this) -------------------------
// Phase stamps of a block, read with s_memrealtime (the 100 MHz clock every CU shares, so that block starts and ends
// line up across the chip), kept in scalar registers and stored by lane 0 of wave 0 at the very end -- no store, hence
// no vmcnt traffic, between the phases.  Indices: 0 block start, 1 tables / descriptors ready, 2 first tile landed in
// LDS, 3 main loop done, 4 epilogue staged, 5 last store issued, 6 block end, 7 = k-steps of the block.
#ifdef VS_STAMP
struct VsStamp { unsigned long long t[8]; };
__device__ __forceinline__ unsigned long long vs_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define VS_ST(st, i) ((st).t[i] = vs_now())
#define VS_STP(stp, i) do { if (stp) (stp)->t[i] = vs_now(); } while (0)
#define VS_ST_FLUSH(p, blk, st)                                                             \
  do {                                                                                      \
    if (threadIdx.x == 0 && (p).stamp) {                                                    \
      for (int i_ = 0; i_ < 8; ++i_) (p).stamp[(long long)(blk) * 8 + i_] = (st).t[i_];     \
    }                                                                                       \
  } while (0)
#else
struct VsStamp {};
#define VS_ST(st, i) do {} while (0)
#define VS_STP(stp, i) do {} while (0)
#define VS_ST_FLUSH(p, blk, st) do {} while (0)
#endif

#define VS_OOB 0x80000000u  // byte offset beyond any tensor: buffer_load returns zeros


__device__ __forceinline__ void mask8(float* f, unsigned bits) {
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = ((bits >> e) & 1u) ? f[e] : 0.f;
}
__device__ __forceinline__ uint4 mask8_bf16(uint4 v, unsigned bits) {
  v.x &= ((bits & 1u) ? 0x0000ffffu : 0u) | ((bits & 2u) ? 0xffff0000u : 0u);
  v.y &= ((bits & 4u) ? 0x0000ffffu : 0u) | ((bits & 8u) ? 0xffff0000u : 0u);
  v.z &= ((bits & 16u) ? 0x0000ffffu : 0u) | ((bits & 32u) ? 0xffff0000u : 0u);
  v.w &= ((bits & 64u) ? 0x0000ffffu : 0u) | ((bits & 128u) ? 0xffff0000u : 0u);
  return v;
}

// -----------------------------------------------------------------------------------------------
// Tile epilogue shared by the implicit-GEMM kernel and the halo-image kernel (conv_halo.hip): fp32
// accumulators of a BM x BN tile (4 waves, WM x WN, wave tile TM x TN) -> BN batch-statistic partials,
// affine, residual (optionally masked by bits), ReLU, bf16 stores as 16-byte channel vectors, and -- BNB --
// the BN-backward sums of the unit this dx belongs to.  `rowm(row)` maps a tile row to its output position
// (or -1).  `smem` is the block's staging area (free once the main loop is done), `statbuf` [2][4 * WM][BN] floats.
// -----------------------------------------------------------------------------------------------
// RAWSYNC: the barriers between the staging writes and reads order LDS traffic only (lgkmcnt + s_barrier).  A
// kernel that keeps asynchronous global->LDS copies in flight across the epilogue (conv_pw.hip) asks for it:
// __syncthreads() also waits vmcnt(0), i.e. one full memory latency for the copies issued a moment ago.
template <bool RAW>
__device__ __forceinline__ void tile_sync() {
  if constexpr (RAW) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  } else {
    __syncthreads();
  }
}

// 16-byte non-temporal load: the residual operand of an epilogue (the other branch's gradient / the block's input) is
// read here for the last time in the pass.
__device__ __forceinline__ uint4 ld_res16(const uint16_t* q) {
  const u32x4 v = __builtin_nontemporal_load((const u32x4*)q);
  return make_uint4(v[0], v[1], v[2], v[3]);
}

template <int BM, int BN, int WM, int WN, bool BNB, bool RAWSYNC = false, bool TWO = false, int EDBG = 0, typename RowMap>
__device__ __forceinline__ void conv_tile_epilogue(const ConvP& p, f32x4 (&acc)[BM / WM / 16][BN / WN / 16],
                                                   char* smem, float* statbuf, int tm, int n0, RowMap rowm,
                                                   VsStamp* stp = nullptr) {
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MR = TM / 16, NR = TN / 16;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 15, lq = lane >> 4;
  // (1) BN batch-statistic partials from the fp32 accumulators (tail rows are
  //     zero-filled, so they add nothing).
  if ((p.flags & VS_CONV_STATS) && !(EDBG & 1)) {  // EDBG: ablations (tools only, wrong results)
#pragma unroll
    for (int b = 0; b < NR; ++b) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int a = 0; a < MR; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[a][b][r];
          s += v;
          q += v * v;
        }
      // every lane quarter parks its partial in LDS and the column's owner adds the 4 x WM of them behind the
      // barrier the staging needs anyway -- no cross-lane exchange here (two dependent ds_bpermute round trips per
      // sum: ~0.35 us of a ~1.3 us epilogue in the ablation, profiles/r02_pw_ab.txt)
      const int col = wn * TN + b * 16 + lr;
      statbuf[(wm * 4 + lq) * BN + col] = s;
      statbuf[(WM * 4 + wm * 4 + lq) * BN + col] = q;
    }
  }
  constexpr int CPR = BN / 8;
  const bool has_res = (p.flags & VS_CONV_RESIDUAL) != 0;
  if (!has_res) {
    // (2a) no residual: affine + ReLU in registers, bf16 straight into an LDS tile, then
    //      whole 16-byte channel vectors are copied out (no unpack / repack pass).
    // Row pitch BN + 8 elements (+16 bytes): a wave's ds_write_b16 covers 16 columns x the four lane quarters' rows
    // q * 4 + r; with a 2 * BN-byte pitch (a multiple of 256 B) the four quarters hit the same 8 banks (4-way conflict:
    // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.24-0.33 on the pointwise launches, where the epilogue is most of
    // the LDS traffic); +16 bytes per row moves each quarter by 16 banks.  The 16-byte reads of the copy-out take a
    // whole row per 16 lanes either way.
    constexpr int EP = BN + 8;
    uint16_t* Eh = (uint16_t*)smem;
    const bool relu = (p.flags & VS_CONV_RELU) != 0;
    if (!(p.flags & (VS_CONV_AFFINE | VS_CONV_RELU))) {
      // raw outputs (every training-mode launch): no multiply-add, no clamp -- with 1..4 k-steps per tile the
      // epilogue's instruction count is a first-order cost
#pragma unroll
      for (int b = 0; b < NR; ++b) {
        const int col = wn * TN + b * 16 + lr;
#pragma unroll
        for (int a = 0; a < MR; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (EDBG & 2) asm volatile("" ::"v"(acc[a][b][r]));
            else Eh[(wm * TM + a * 16 + lq * 4 + r) * EP + col] = f32_to_bf16(acc[a][b][r]);
          }
      }
    } else {
#pragma unroll
    for (int b = 0; b < NR; ++b) {
      const int col = wn * TN + b * 16 + lr;
      float sc = 1.f, sh = 0.f;
      if ((p.flags & VS_CONV_AFFINE) && (n0 + col < p.Ncols)) {
        sc = p.scale[n0 + col];
        sh = p.shift[n0 + col];
      }
#pragma unroll
      for (int a = 0; a < MR; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wm * TM + a * 16 + lq * 4 + r;
          float v = acc[a][b][r] * sc + sh;
          if (relu) v = fmaxf(v, 0.f);
          Eh[row * EP + col] = f32_to_bf16(v);
        }
    }
    }
    tile_sync<RAWSYNC>();
    VS_STP(stp, 4);
    if constexpr (BNB) {
      // copy-out + the consumer BN's backward sums: thread = one 8-channel column x (256 / CPR) row
      // lanes; the saved conv outputs of all its rows are requested before the first use
      constexpr int RL = 256 / CPR, IT = BM / RL;
      static_assert(256 % CPR == 0 && BM % RL == 0, "tile shape");
      const int c8 = tid % CPR, rl = tid / CPR;
      const int n = n0 + c8 * 8;
      const bool nok = n < p.Ncols;
      const int nn = nok ? n : 0;
      float mu[8], is[8], ga[8], be[8], sg[8], sx[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        mu[e] = p.bn_mean[nn + e];
        is[e] = p.bn_invstd[nn + e];
        ga[e] = p.bn_gamma[nn + e];
        be[e] = p.bn_beta[nn + e];
        sg[e] = 0.f;
        sx[e] = 0.f;
      }
      uint4 yv4[IT];
      int mm[IT];
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int row = rl + i * RL;
        const int m = rowm(row);
        mm[i] = nok ? m : -1;
        yv4[i] = *(const uint4*)(p.bny + (long long)(mm[i] >= 0 ? mm[i] : 0) * p.bny_ld + nn);
      }
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int row = rl + i * RL;
        const uint4 v = *(const uint4*)(Eh + row * EP + c8 * 8);
        if (mm[i] >= 0) {
          *(uint4*)(p.y + (long long)mm[i] * p.y_ld + n) = v;
          float g[8], yv[8];
          unpack8_bf16(v, g);
          unpack8_bf16(yv4[i], yv);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            g[e] = ((yv[e] - mu[e]) * is[e] * ga[e] + be[e]) > 0.f ? g[e] : 0.f;
            sg[e] += g[e];
            sx[e] += g[e] * (yv[e] - mu[e]) * is[e];
          }
        }
      }
      float* red = (float*)(smem + BM * EP * 2);  // behind the bf16 tile
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[tid * 16 + e] = sg[e];
        red[tid * 16 + 8 + e] = sx[e];
      }
      tile_sync<RAWSYNC>();
      if (tid < BN && n0 + tid < p.Ncols) {  // fixed order over the row lanes
        const int cc = tid >> 3, e = tid & 7;
        float ts = 0.f, tq = 0.f;
        for (int r = 0; r < RL; ++r) {
          ts += red[(r * CPR + cc) * 16 + e];
          tq += red[(r * CPR + cc) * 16 + 8 + e];
        }
        float* dst = p.stats + (long long)tm * 2 * p.Ncols;
        dst[n0 + tid] = ts;
        dst[p.Ncols + n0 + tid] = tq;
      }
    } else {
    // every LDS read of the thread's vectors is issued before the first store (a read -> store loop was a chain of
    // LDS latencies: ~0.2 us of a ~1.3 us epilogue, which on 1..4-step layers is a third of the block)
    constexpr int CO = (BM * CPR + 255) / 256;
    uint4 cv[CO];
    int cm[CO];
#pragma unroll
    for (int u = 0; u < CO; ++u) {
      const int idx = tid + u * 256;
      const int row = idx / CPR, c8 = idx - row * CPR;
      const bool in = idx < BM * CPR;
      const int m = in ? rowm(row) : -1;
      cm[u] = (m >= 0 && n0 + c8 * 8 < p.Ncols) ? m : -1;
      cv[u] = *(const uint4*)(Eh + (in ? row : 0) * EP + c8 * 8);
    }
#pragma unroll
    for (int u = 0; u < CO; ++u) {
      const int c8 = (tid + u * 256) % CPR;
      if (cm[u] >= 0) *(uint4*)(p.y + (long long)cm[u] * p.y_ld + n0 + c8 * 8) = cv[u];
    }
    }
  } else {
    // (2b) residual add: fp32 tile through LDS, residual read as 16-byte vectors
    float* E = (float*)smem;
#pragma unroll
    for (int b = 0; b < NR; ++b) {
      const int col = wn * TN + b * 16 + lr;
      float sc = 1.f, sh = 0.f;
      if ((p.flags & VS_CONV_AFFINE) && (n0 + col < p.Ncols)) {
        sc = p.scale[n0 + col];
        sh = p.shift[n0 + col];
      }
#pragma unroll
      for (int a = 0; a < MR; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wm * TM + a * 16 + lq * 4 + r;
          E[row * BN + col] = acc[a][b][r] * sc + sh;
        }
    }
    tile_sync<RAWSYNC>();
    VS_STP(stp, 4);
    if constexpr (BNB) {
      // residual add + copy-out + the consumer BN's backward sums (ReLU mask from the unit's bit mask):
      // same thread layout as the no-residual variant above
      constexpr int RL = 256 / CPR, IT = BM / RL;
      static_assert(256 % CPR == 0 && BM % RL == 0, "tile shape");
      const int c8 = tid % CPR, rl = tid / CPR;
      const int n = n0 + c8 * 8;
      const bool nok = n < p.Ncols;
      const int nn = nok ? n : 0;
      const int bpr = p.Ncols >> 3;
      constexpr bool two = TWO;  // also the shortcut unit's sums (same gradient, same mask); a template flag: the
      // 54 extra registers of this path must not count against every other BN-sums launch's occupancy
      float mu[8], is[8], sg[8], sx[8], mu2[8], is2[8], sx2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        mu[e] = p.bn_mean[nn + e];
        is[e] = p.bn_invstd[nn + e];
        mu2[e] = two ? p.bn_mean2[nn + e] : 0.f;
        is2[e] = two ? p.bn_invstd2[nn + e] : 0.f;
        sg[e] = 0.f;
        sx[e] = 0.f;
        sx2[e] = 0.f;
      }
      uint4 yv4[IT], rv4[IT], yw4[IT];
      unsigned bits[IT], rbits[IT];
      int mm[IT];
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int row = rl + i * RL;
        const int m = rowm(row);
        mm[i] = nok ? m : -1;
        const long long mc = mm[i] >= 0 ? mm[i] : 0;
        yv4[i] = *(const uint4*)(p.bny + mc * p.bny_ld + nn);
        yw4[i] = two ? *(const uint4*)(p.bny2 + mc * p.bny2_ld + nn) : make_uint4(0u, 0u, 0u, 0u);
        rv4[i] = ld_res16(p.res + mc * p.res_ld + nn);
        bits[i] = p.bn_bits[mc * bpr + (nn >> 3)];
        rbits[i] = p.res_bits ? (unsigned)p.res_bits[mc * bpr + (nn >> 3)] : 0xffu;
      }
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int row = rl + i * RL;
        const float4 v0 = *(const float4*)(E + row * BN + c8 * 8);
        const float4 v1 = *(const float4*)(E + row * BN + c8 * 8 + 4);
        if (mm[i] >= 0) {
          float v[8], rf[8], g[8], yv[8];
          v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w;
          v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
          unpack8_bf16(rv4[i], rf);
          mask8(rf, rbits[i]);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rf[e];
          const uint4 o = pack8_bf16(v);
          *(uint4*)(p.y + (long long)mm[i] * p.y_ld + n) = o;
          unpack8_bf16(o, g);  // the sums see dz as stored
          unpack8_bf16(yv4[i], yv);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            g[e] = ((bits[i] >> e) & 1u) ? g[e] : 0.f;
            sg[e] += g[e];
            sx[e] += g[e] * (yv[e] - mu[e]) * is[e];
          }
          if (two) {
            float yw[8];
            unpack8_bf16(yw4[i], yw);
#pragma unroll
            for (int e = 0; e < 8; ++e) sx2[e] += g[e] * (yw[e] - mu2[e]) * is2[e];
          }
        }
      }
      tile_sync<RAWSYNC>();  // every thread is done with the fp32 tile: its space holds the row-lane sums
      float* red = (float*)smem;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[tid * 16 + e] = sg[e];
        red[tid * 16 + 8 + e] = sx[e];
      }
      tile_sync<RAWSYNC>();
      float ts_keep = 0.f;
      if (tid < BN && n0 + tid < p.Ncols) {  // fixed order over the row lanes
        const int cc = tid >> 3, e = tid & 7;
        float ts = 0.f, tq = 0.f;
        for (int r = 0; r < RL; ++r) {
          ts += red[(r * CPR + cc) * 16 + e];
          tq += red[(r * CPR + cc) * 16 + 8 + e];
        }
        float* dst = p.stats + (long long)tm * 2 * p.Ncols;
        dst[n0 + tid] = ts;
        dst[p.Ncols + n0 + tid] = tq;
        ts_keep = ts;
      }
      if (two) {  // the second unit: the same sum(g), its own sum(g * xhat) through the same staging space
        tile_sync<RAWSYNC>();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 16 + 8 + e] = sx2[e];
        tile_sync<RAWSYNC>();
        if (tid < BN && n0 + tid < p.Ncols) {
          const int cc = tid >> 3, e = tid & 7;
          float tq = 0.f;
          for (int r = 0; r < RL; ++r) tq += red[(r * CPR + cc) * 16 + 8 + e];
          float* dst = p.stats2 + (long long)tm * 2 * p.Ncols;
          dst[n0 + tid] = ts_keep;
          dst[p.Ncols + n0 + tid] = tq;
        }
      }
    } else {
    // the thread's residual vectors (global) and mask bytes are all requested before the first use: the
    // load -> add -> store loop was a chain of memory latencies per tile
    constexpr int CO = (BM * CPR + 255) / 256;
    uint4 rv[CO];
    unsigned rb[CO];
    int cm[CO];
#pragma unroll
    for (int u = 0; u < CO; ++u) {
      const int idx = tid + u * 256;
      const int row = idx / CPR, c8 = idx - row * CPR;
      const int m = idx < BM * CPR ? rowm(row) : -1;
      const int n = n0 + c8 * 8;
      cm[u] = (m >= 0 && n < p.Ncols) ? m : -1;
      const long long mc = cm[u] >= 0 ? cm[u] : 0;
      const int nc = cm[u] >= 0 ? n : 0;
      rv[u] = ld_res16(p.res + mc * p.res_ld + nc);
      rb[u] = p.res_bits ? (unsigned)p.res_bits[mc * (p.Ncols >> 3) + (nc >> 3)] : 0xffu;
    }
#pragma unroll
    for (int u = 0; u < CO; ++u) {
      const int idx = tid + u * 256;
      const int row = idx / CPR, c8 = idx - row * CPR;
      if (cm[u] >= 0) {
        float v[8];
        const float4 v0 = *(const float4*)(E + row * BN + c8 * 8);
        const float4 v1 = *(const float4*)(E + row * BN + c8 * 8 + 4);
        v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w;
        v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
        float rf[8];
        unpack8_bf16(rv[u], rf);
        if (p.res_bits) mask8(rf, rb[u]);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rf[e];
        if (p.flags & VS_CONV_RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *(uint4*)(p.y + (long long)cm[u] * p.y_ld + n0 + c8 * 8) = pack8_bf16(v);
      }
    }
    }
  }
  VS_STP(stp, 5);
  if ((p.flags & VS_CONV_STATS) && tid < BN && n0 + tid < p.Ncols) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < 4 * WM; ++w) {  // fixed order: wave rows, lane quarters
      s += statbuf[w * BN + tid];
      q += statbuf[(4 * WM + w) * BN + tid];
    }
    float* dst = p.stats + (long long)tm * 2 * p.Ncols;
    dst[n0 + tid] = s;
    dst[p.Ncols + n0 + tid] = q;
  }
}

// ---- halo-image kernel (conv_halo.hip): tile geometry, planned on the host ----
struct HaloGeo {
  int kind;           // 0: taps along T, 1: taps in the H/W plane
  int G, D1, D2, O1, O2;
  int O2p;            // output row pitch along axis 2 (>= O2: lines padded to 16 rows, kind 1)
  int k1, k2, p1, p2;
  int flip;           // dgrad
  int T, H, W, HW;    // extents of the (equal) input / output position grids
  int per;            // groups per clip (kind 0) / line chunks per frame (kind 1)
  int ngroups;        // groups in the whole tensor
  int RA;             // G * D1 * D2
  int rows;           // G * O1 * O2 (valid tile rows)
  int tilesM, tilesN;
  int mrw, nrw;       // wave tile in 16-row / 16-column units
};


// Geometry of the halo kernel for this launch (mode = kernel MODE, dgrad: mirrored taps, flags = desc.flags),
// or false when the implicit-GEMM kernel runs it.
bool vs_halo_plan(const ConvP& p, int mode, int dgrad, int flags, HaloGeo* out);
int vs_halo_launch(const ConvP& p, const HaloGeo& g, hipStream_t st);
void vs_halo_variant(const HaloGeo& g, int* depth, int* taps_unrolled);

// ---- persistent pointwise kernel (conv_pw.hip): shallow-K 1x1x1 convs, weight slice resident in LDS ----
struct PwGeo {
  int bn;         // columns per weight slice (64 / 128)
  int nsl;        // weight slices
  int gx;         // tile lists per XCD
  int ngroups;    // tile lists (8 * gx): list q walks row tiles q, q + ngroups, ...
  int tilesM;     // 64-row tiles (= rows of the batch-statistic partials)
  int nk;         // 64-wide k-chunks
  int nslot;      // activation ring slots (nslot - 1 chunks in flight)
  int epi_bytes;  // epilogue staging area
  int smem;       // dynamic LDS of the launch
  int dense;      // rows are consecutive positions of the gathered tensor
  int dbg;        // VS_PW_DBG ablations (wrong results; tools only)
};
bool vs_pw_plan(const ConvP& p, int mode, int flags, PwGeo* out);
int vs_pw_launch(const ConvP& p, const PwGeo& g, hipStream_t st);
bool vs_pw_aol_ok(const PwGeo& g, const ConvP& p);

// ---- deep-pipeline kernel (conv_deep.hip): 256 x 256 x 64 tile, 8 waves, sub-buffer ring 7 phases deep ----
struct DeepGeo {
  int tilesM, tilesN;  // 256 x 256 tiles (tilesM = rows of the batch-statistic / BN-backward partials)
  int smem;            // dynamic LDS of the launch
};
bool vs_deep_plan(const ConvP& p, int mode, int flags, DeepGeo* out);
int vs_deep_launch(const ConvP& p, int mode, const DeepGeo& g, hipStream_t st);

