// Implicit-GEMM Conv3d forward / data-gradient on bf16 MFMA (gfx950).
//
// Replaces nn.Conv3d (cuDNN fwd / bwd-data) of the SlowFast trunk reached from
// vidsitu_code/mdl_sf_base.py:22-33.  GEMM view (channels-last activations):
//   rows    m = (n, to, ho, wo)           output positions      (fwd)
//   cols    n = cout
//   K       k = (dt, dh, dw, cin)         gathered on the fly from x
//   A[m][k] = x[n, to*s-p+dt, ..][cin]    16-byte units = 8 consecutive channels
//   B[n][k] = w[cout][dt][dh][dw][cin]    K-contiguous ("B^T" layout)
// dgrad is the same kernel with the gather transposed (rows = input positions,
// gathered tensor = dy, stride becomes a divisibility test).
//
// Tile: BM x BN x 64, 256 threads (4 waves, WM x WN), v_mfma_f32_16x16x32_bf16,
// register-staged global->LDS double buffering (loads of tile k+1 are issued
// before the MFMAs of tile k and written to LDS after them), XOR-swizzled
// 128-byte LDS rows (conflict-free ds_read_b128 fragment reads), fp32 epilogue
// staged through LDS so stores are whole 16-byte channel vectors.
#include "common.h"

#include <mutex>

#include "conv_tile.h"

template <int BM, int BN, int NSTAGE = 2>
struct ConvSmem {
  static constexpr int STAGE = (BM + BN) * 128;
  static constexpr int EPI = BM * BN * 4;
  static constexpr int MAIN = (NSTAGE * STAGE > EPI) ? NSTAGE * STAGE : EPI;
};

// FAST (taps <= 32, every conv of the trunk but the Cin-padded stems): per row a bitmask of
// valid taps and a base byte offset are computed ONCE; per k-step a load is
// `valid ? base + delta[k] : OOB` through a buffer resource whose bounds check supplies the
// zero padding -- no branches, no 64-bit address math in the loop.
// !FAST: generic per-load coordinate tests (any tap count).
// DBG (diagnostic builds only, wrong results): 1 = no global loads in the loop, 2 = no MFMAs,
// 3 = no LDS stores in the loop, 4 = no LDS fragment reads (operands stay whatever they were).
// NS (FAST only): 0 = register-staged pipeline; >= 2 = LDS-DMA ring of NS stages
// (`buffer_load_dwordx4 ... lds`, NS-1 tiles in flight, one raw barrier per k-step).
// BNB: the dgrad variant that also emits the consumer BN's backward sums (VS_CONV_BNBWD); a template flag
// so that the extra epilogue registers do not count against every other launch's occupancy.
// AOL (MODE 0 ring launches only): apply on load, see ConvP::in_scale -- the constants of the input channels live in
// LDS behind the statistic rows, every A fragment goes through aol_frag_k after its LDS read.
template <int BM, int BN, int WM, int WN, int MODE, bool FAST, int DBG = 0, int NS = 0, bool BNB = false,
          bool BNB2 = false, bool AOL = false>
__device__ __forceinline__ void conv_igemm_body(const ConvP& p, const int blk, const int nblk) {
  // (blk / nblk: this block's index and the block count of THIS convolution's grid -- blockIdx.x / gridDim.x of the
  //  stand-alone launch, a sub-range of the grid in the dgrad + wgrad pair launch of conv_pair.hip)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int AI = BM / 32;
  constexpr int BJ = (BN + 31) / 32;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MR = TM / 16, NR = TN / 16;
  constexpr int STAGE = ConvSmem<BM, BN>::STAGE;
  constexpr int MAIN = ConvSmem<BM, BN, (NS > 2 ? NS : 2)>::MAIN;
  static_assert(NS == 0 || (FAST && NS >= 2 && BN >= 32), "LDS-DMA ring needs the FAST gather");
  static_assert(!AOL || (MODE == 0 && NS >= 2), "apply on load: pointwise ring launches");
  static_assert(WM * WN == 4, "4 waves");
  static_assert(TM % 16 == 0 && TN % 16 == 0, "wave tile");

  const int tid = threadIdx.x;
  [[maybe_unused]] VsStamp vst;
#ifdef VS_STAMP
  for (int i_ = 0; i_ < 8; ++i_) vst.t[i_] = 0ull;
#endif
  VS_ST(vst, 0);
  // XCD-aware remap: blocks b, b+8, ... share an XCD (speed only); give each
  // XCD a contiguous run of tiles so neighbouring N-tiles re-read A from its L2.
  int swz;
  {
    const int nwg = nblk, bid = blk;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // (uniform integer divisions by launch parameters cost ~35 scalar instructions each, 64-bit ones ~100: with
  //  1..4 k-steps per block the prologue was longer than the main loop -- the common cases take no division)
  const int ksplit = (p.splitK == 1) ? 0 : swz % p.splitK;  // the splits of one tile are neighbours (same XCD run)
  const int tile_id = (p.splitK == 1) ? swz : swz / p.splitK;
  int tn = 0, tm = tile_id;
  if (p.tilesN == 2) {
    tn = tile_id & 1;
    tm = tile_id >> 1;
  } else if (p.tilesN == 4) {
    tn = tile_id & 3;
    tm = tile_id >> 2;
  } else if (p.tilesN != 1) {
    tn = tile_id % p.tilesN;
    tm = tile_id / p.tilesN;
  }
  const int n0 = tn * BN;
  const int kc = tid & 7, lrow = tid >> 3;

  // ---- stride class of this tile (MODE 2 only) ----
  const bool cls = (MODE == 2) && p.ncls > 0;
  int m0 = tm * BM;
  int Mloc = p.M;                                // rows of the row space this tile indexes
  int cT = p.Rt, cH = p.Rh, cW = p.Rw;           // extents of that row space
  int r0T = 0, r0H = 0, r0W = 0, stT = 1, stH = 1, stW = 1;  // r = r0 + st * i
  int qT = 0, qH = 0, qW = 0, ntT = p.kT, ntH = p.kH, ntW = p.kW;  // taps: dd = q + st * i
  if (cls) {
    stT = 1 << p.shT; stH = 1 << p.shH; stW = 1 << p.shW;
    // (strides are powers of two: residues, first rows and counts are masks and shifts -- as written with % and /
    //  on run-time values this block was ~15 scalar divisions, ~500 instructions in front of a 1..2-step main loop)
    int slot, ti;
    fast_divmod(tm, p.nslots, 1.0f / (float)p.nslots, ti, slot);  // tm < 2^24
    ti = __builtin_amdgcn_readfirstlane(ti);
    slot = __builtin_amdgcn_readfirstlane(slot);
    const int cq = p.cls_ids[slot];
    if (ti >= p.cls_tiles[cq]) {  // empty slot of the class-interleaved numbering
      if (BNB && tid < BN && n0 + tid < p.Ncols) {
        float* dst = p.stats + (long long)tm * 2 * p.Ncols;
        dst[n0 + tid] = 0.f;
        dst[p.Ncols + n0 + tid] = 0.f;
      }
      return;
    }
    m0 = ti * BM;
    qW = cq & (stW - 1); qH = (cq >> p.shW) & (stH - 1); qT = cq >> (p.shW + p.shH);
    auto first = [](int q, int off, int st) { return (q - off) & (st - 1); };  // two's complement: also for q < off
    auto count = [](int R, int r0, int st, int sh) { return r0 < R ? (R - r0 + st - 1) >> sh : 0; };
    r0T = first(qT, p.offT, stT); r0H = first(qH, p.offH, stH); r0W = first(qW, p.offW, stW);
    cT = count(p.Rt, r0T, stT, p.shT); cH = count(p.Rh, r0H, stH, p.shH); cW = count(p.Rw, r0W, stW, p.shW);
    Mloc = p.nclips * cT * cH * cW;
    ntT = qT < p.kT ? (p.kT - qT + stT - 1) >> p.shT : 0;
    ntH = qH < p.kH ? (p.kH - qH + stH - 1) >> p.shH : 0;
    ntW = qW < p.kW ? (p.kW - qW + stW - 1) >> p.shW : 0;
  }
  const int Keff = cls ? ntT * ntH * ntW * p.Cg : p.K;  // this tile's reduction length
  const int K8 = Keff >> 3;

  float* statbuf = (float*)(smem + MAIN);               // [2][4 * WM][BN]
  int* rowpos = (int*)(smem + MAIN + 8 * WM * BN * 4);  // [BM] output position of each tile row
  int2* ktab = (int2*)(smem + MAIN + 8 * WM * BN * 4 + (MODE == 2 ? BM * 4 : 0));  // [K8]

  if (MODE != 0) {
    const int C8 = p.Cg >> 3;
    const float rcpC8 = 1.0f / (float)C8, rcpkW = 1.0f / (float)p.kW, rcpkH = 1.0f / (float)p.kH;
    const int K8pad = (((Keff + 63) >> 6) + (NS > 0 ? NS : 0)) << 3;  // + the ring's run-ahead + the entry read one k-step early
    for (int k8 = tid; k8 < K8pad; k8 += 256) {
      if (k8 >= K8) {  // K tail: tap 31 is never valid (FAST requires <= 31 taps)
        ktab[k8] = FAST ? make_int2(31, 0) : make_int2(0, 0);
        continue;
      }
      int tap, c8;
      if (p.korder && !cls) {  // chunk-major: k-tile kt = chunk kt / taps of tap kt % taps (Cg % 64 == 0)
        const int ntaps = p.kT * p.kH * p.kW;
        int kt = k8 >> 3, chunk;
        fast_divmod(kt, ntaps, 1.0f / (float)ntaps, chunk, tap);
        c8 = chunk * 8 + (k8 & 7);
      } else {
        fast_divmod(k8, C8, rcpC8, tap, c8);  // k8 < 2^24: float-reciprocal splits (three integer divisions per entry before)
      }
      int dw, dh, dt;
      if (cls) {  // enumerate only the taps of this class
        int iw, t2, ih, it;
        fast_divmod(tap, ntW, 1.0f / (float)ntW, t2, iw);
        fast_divmod(t2, ntH, 1.0f / (float)ntH, it, ih);
        dw = qW + stW * iw; dh = qH + stH * ih; dt = qT + stT * it;
        tap = (dt * p.kH + dh) * p.kW + dw;
      } else {
        int t2;
        fast_divmod(tap, p.kW, rcpkW, t2, dw);
        fast_divmod(t2, p.kH, rcpkH, dt, dh);
      }
      if (FAST) {
        long long dpos;
        if (MODE == 1) dpos = (((long long)dt * p.Gh + dh) * p.Gw + dw) * p.tmul;
        else dpos = -(((long long)(dt >> p.shT) * p.Gh + (dh >> p.shH)) * p.Gw + (dw >> p.shW));
        // .x = tap (5 bits, the row's validity bit) | 16-byte unit of this k inside a weight row
        ktab[k8] = make_int2(tap | ((tap * C8 + c8) << 5), (int)((dpos * p.g_ld + c8 * 8) * 2));
      } else {
        ktab[k8] = make_int2(dt | (dh << 8) | (dw << 16), c8 * 8);
      }
    }
  }

  // ---- per-thread row state (this thread always loads 16-byte unit kc of rows lrow+32i)
  int rT[AI], rH[AI], rW[AI];
  long long rbase[AI];
  unsigned roff[AI], vmask[AI];
  // row -> (clip, t, h, w): float-reciprocal division (exact below 2^24 rows, two fix-ups) instead of three
  // ~40-instruction integer divisions per row
  const bool small_rows = Mloc < (1 << 24);
  const float rcpW = 1.0f / (float)cW, rcpH = 1.0f / (float)cH, rcpT = 1.0f / (float)cT;
  // One row's state: position decode, base byte offset, bitmask of valid taps (FAST), rowpos (MODE 2).
  auto decode_row = [&](const int m, const int slot, const bool write_rowpos, int& oT, int& oH, int& oW, long long& obase,
                        unsigned& ooff, unsigned& omask) __attribute__((always_inline)) {
    oT = oH = oW = -(1 << 20);
    obase = -1;
    ooff = VS_OOB;
    omask = 0u;
    if (MODE == 2 && write_rowpos) rowpos[slot] = -1;
    if (m < Mloc) {
      int rw, t1, rh, t2, rt, n;
      if (MODE == 0 && p.dense) {  // rows are consecutive positions of the gathered tensor: nothing to decode
        rw = m; rh = 0; rt = 0; n = 0;
      } else if (small_rows) {
        fast_divmod(m, cW, rcpW, t1, rw);
        fast_divmod(t1, cH, rcpH, t2, rh);
        fast_divmod(t2, cT, rcpT, n, rt);
      } else {
        rw = m % cW; t1 = m / cW;
        rh = t1 % cH; t2 = t1 / cH;
        rt = t2 % cT; n = t2 / cT;
      }
      rw = r0W + stW * rw;  // (identity unless the tile belongs to a stride class)
      rh = r0H + stH * rh;
      rt = r0T + stT * rt;
      if (MODE == 2 && write_rowpos)
        rowpos[slot] = ((n * p.Rt + rt) * p.Rh + rh) * p.Rw + rw;
      if (MODE == 0) {
        const long long pos = p.dense ? (long long)m :
            ((long long)(n * p.Gt + rt * p.mulT) * p.Gh + rh * p.mulH) * p.Gw + rw * p.mulW;
        obase = pos * p.g_ld;
        ooff = (unsigned)(pos * p.g_ld * 2);
        omask = 1u;
      } else {
        const int ct = rt * p.mulT + p.offT, ch = rh * p.mulH + p.offH, cw = rw * p.mulW + p.offW;
        obase = (long long)n * p.Gt * p.Gh * p.Gw;
        oT = ct;
        oH = ch;
        oW = cw;
        if (FAST) {
          long long pos0;
          if (MODE == 1) pos0 = obase + ((long long)ct * p.Gh + ch) * p.Gw + cw;
          else pos0 = obase + ((long long)(ct >> p.shT) * p.Gh + (ch >> p.shH)) * p.Gw + (cw >> p.shW);
          ooff = (unsigned)(pos0 * p.g_ld * 2);  // exact modulo 2^32 whenever the tap is valid
          // per-axis validity first (kT + kH + kW tests), then one AND per tap
          auto axis_mask = [&](int c, int kk, int sh, int G) {
            unsigned mm = 0u;
            for (int dd = 0; dd < kk; ++dd) {
              int v = c + p.tmul * dd;
              bool ok = true;
              if (MODE == 2) {
                ok = (v & ((1 << sh) - 1)) == 0;
                v >>= sh;
              }
              ok = ok && ((unsigned)v < (unsigned)G);
              mm |= (ok ? 1u : 0u) << dd;
            }
            return mm;
          };
          const unsigned mt = axis_mask(ct, p.kT, p.shT, p.Gt), mh = axis_mask(ch, p.kH, p.shH, p.Gh),
                         mw = axis_mask(cw, p.kW, p.shW, p.Gw);
          unsigned mk = 0u;
          int tap = 0;
          for (int dt = 0; dt < p.kT; ++dt)
            for (int dh = 0; dh < p.kH; ++dh) {
              const unsigned th = (mt >> dt) & (mh >> dh) & 1u;
              mk |= (th ? mw : 0u) << tap;
              tap += p.kW;
            }
          omask = mk;
        }
      }
    }
  };
  // Round 6 (profiles/r06_launch_anatomy.txt): the eight threads that fetch the eight 16-byte units of a row all
  // decoded that row, AI rows each -- 2.9-4.2 us of prologue in front of a 9-19 us block on the multi-tap layers
  // (1.0 us on a pointwise one).  The gathering launches now decode ONE row per thread into an LDS table (aliased
  // on the statistic rows, which only the epilogue uses) and every thread picks up its AI entries behind the barrier
  // the k-table needs anyway: AI x fewer decode instructions per wave, same integers.
  constexpr bool ROWTAB = FAST && !AOL && BM <= 256;
  static_assert(8 * WM * BN * 4 >= BM * 8, "the row table fits the statistic rows it is aliased on");
  const bool use_rowtab = ROWTAB && !(MODE == 0 && p.dense);
  uint2* rowtab = (uint2*)statbuf;  // [BM] (roff, vmask)
  if (use_rowtab) {
    if (tid < BM) {
      int t_, h_, w_;
      long long b_;
      unsigned o_, k_;
      decode_row(m0 + tid, tid, true, t_, h_, w_, b_, o_, k_);
      rowtab[tid] = make_uint2(o_, k_);
    }
#pragma unroll
    for (int i = 0; i < AI; ++i) {  // (filled behind the barrier below)
      rT[i] = rH[i] = rW[i] = 0;
      rbase[i] = -1;
      roff[i] = VS_OOB;
      vmask[i] = 0u;
    }
  } else {
#pragma unroll
    for (int i = 0; i < AI; ++i)
      decode_row(m0 + lrow + 32 * i, lrow + 32 * i, kc == 0, rT[i], rH[i], rW[i], rbase[i], roff[i], vmask[i]);
  }
  // weight rows of this thread (byte offsets; OOB beyond Ncols)
  unsigned boff[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int row = lrow + 32 * j, n = n0 + row;
    boff[j] = (row < BN && n < p.Ncols) ? (unsigned)((long long)n * p.K * 2) : VS_OOB;
  }
  const __amdgpu_buffer_rsrc_t xsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);

  u32x4 ra0[AI], rb0[BJ], ra1[AI], rb1[BJ];  // two tiles of loads in flight (prefetch distance 2)
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  auto gload = [&](int kt, u32x4* ra, u32x4* rb) __attribute__((always_inline)) {
    const int k8 = kt * 8 + kc;
    const bool kval = k8 < K8;
    if (DBG == 1 && kt > 1) return;
    if (FAST) {
      // branch-free: every predicate is folded into the byte offset (OOB -> zeros)
      int2 e = make_int2(0, k8 * 16);
      if (MODE != 0) e = ktab[k8];  // table is padded to whole k-steps (tap 31 = never valid)
      const unsigned kbit = (MODE == 0) ? (unsigned)kval : 1u;
      const unsigned wk = (MODE == 0) ? (unsigned)k8 : ((unsigned)e.x >> 5);
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const unsigned ok = kbit & (vmask[i] >> (e.x & 31)) & 1u;
        const unsigned off = ok ? roff[i] + (unsigned)e.y : VS_OOB;
        ra[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, off, 0, 0));
      }
#pragma unroll
      for (int j = 0; j < BJ; ++j) {
        const unsigned ok = (unsigned)kval & (unsigned)(boff[j] != VS_OOB);
        const unsigned off = ok ? boff[j] + wk * 16u : VS_OOB;
        rb[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, off, 0, 0));
      }
      return;
    }
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        ra[i] = zero4;
        if (kval && rbase[i] >= 0) ra[i] = *(const u32x4*)(p.x + rbase[i] + k8 * 8);
      }
    } else {
      int2 e = make_int2(0, 0);
      if (kval) e = ktab[k8];
      const int dt = (e.x & 0xff) * p.tmul, dh = ((e.x >> 8) & 0xff) * p.tmul,
                dw = ((e.x >> 16) & 0xff) * p.tmul;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        int ti = rT[i] + dt, hi = rH[i] + dh, wi = rW[i] + dw;
        bool ok = kval;
        if (MODE == 2) {
          ok = ok && (((ti & ((1 << p.shT) - 1)) | (hi & ((1 << p.shH) - 1)) |
                       (wi & ((1 << p.shW) - 1))) == 0);
          ti >>= p.shT;
          hi >>= p.shH;
          wi >>= p.shW;
        }
        ok = ok && ((unsigned)ti < (unsigned)p.Gt) && ((unsigned)hi < (unsigned)p.Gh) &&
             ((unsigned)wi < (unsigned)p.Gw);
        ra[i] = zero4;
        if (ok) {
          const long long pos = rbase[i] + (long long)((ti * p.Gh + hi) * p.Gw + wi);
          ra[i] = *(const u32x4*)(p.x + pos * p.g_ld + e.y);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int row = lrow + 32 * j;
      const int n = n0 + row;
      rb[j] = zero4;
      if (kval && row < BN && n < p.Ncols) rb[j] = *(const u32x4*)(p.w + (long long)n * p.K + k8 * 8);
    }
  };

  auto sstore = [&](int buf, const u32x4* ra, const u32x4* rb) __attribute__((always_inline)) {
    if (DBG == 3) {
      asm volatile("" ::"v"(ra[0]), "v"(rb[0]));  // keep the loads alive
      return;
    }
    char* A = smem + buf * STAGE;
    char* B = A + BM * 128;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int row = lrow + 32 * i;
      *(u32x4*)(A + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int row = lrow + 32 * j;
      if (row < BN) *(u32x4*)(B + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4)) = rb[j];
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 15, lq = lane >> 4;

  f32x4 acc[MR][NR];
#pragma unroll
  for (int a = 0; a < MR; ++a)
#pragma unroll
    for (int b = 0; b < NR; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // straight-line MFMA chains (no conditional exit: a branch here makes the compiler shuttle
  // every accumulator between AGPRs and VGPRs each k-step); a K tail multiplies staged zeros.
  // AOL: [2][AOL_K] scale / shift of input channel k behind the statistic rows (MODE 0: no tables there); aol_k0 =
  // first channel of the k-tile `compute` is about to multiply
  constexpr int AOL_K = 512;
  float* aol_tab = (float*)(smem + MAIN + 8 * WM * BN * 4);
  int aol_k0 = 0;
  if (AOL) {
    for (int i = tid; i < AOL_K; i += 256) {
      aol_tab[i] = i < p.K ? p.in_scale[i] : 0.f;
      aol_tab[AOL_K + i] = i < p.K ? p.in_shift[i] : 0.f;
    }  // (published by the ring loop's first barrier)
  }
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const char* A = smem + buf * STAGE;
    const char* B = A + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[MR], bfr[NR];
      const int ch = ks * 4 + lq;
#pragma unroll
      for (int a = 0; a < MR; ++a) {
        const int row = wm * TM + a * 16 + lr;
        if (DBG == 4) af[a] = __builtin_bit_cast(bf16x8, (u32x4){(unsigned)row, (unsigned)ch, 1u, 2u});
        else af[a] = *(const bf16x8*)(A + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
        if (AOL) {  // rows past M were zero-filled and have to stay zero (the statistic partials sum every tile row)
          const bf16x8 t = aol_frag_k(af[a], aol_tab + aol_k0 + ch * 8, aol_tab + AOL_K + aol_k0 + ch * 8);
          af[a] = (m0 + row < p.M) ? t : __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
        }
      }
#pragma unroll
      for (int b = 0; b < NR; ++b) {
        const int row = wn * TN + b * 16 + lr;
        if (DBG == 4) bfr[b] = __builtin_bit_cast(bf16x8, (u32x4){(unsigned)row, (unsigned)ch, 3u, 4u});
        else bfr[b] = *(const bf16x8*)(B + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
      }
      if (DBG == 2) {
#pragma unroll
        for (int a = 0; a < MR; ++a) asm volatile("" ::"v"(af[a]));
#pragma unroll
        for (int b = 0; b < NR; ++b) asm volatile("" ::"v"(bfr[b]));
      } else {
#pragma unroll
        for (int a = 0; a < MR; ++a)
#pragma unroll
          for (int b = 0; b < NR; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
      }
    }
  };

  if (MODE == 2 && cls && Keff == 0) {
    // no tap reaches this stride class (e.g. odd positions of a 1x1 stride-2 conv): the gradient
    // is zero, the tile is a copy of the residual (or zeros) -- no staging, no MFMA, no LDS tile.
    // In place (residual == output, an accumulating dgrad): nothing to do at all.
    if ((p.flags & VS_CONV_RESIDUAL) && p.res == p.y && p.res_bits == nullptr) {
      if (BNB && tid < BN && n0 + tid < p.Ncols) {
        float* dst = p.stats + (long long)tm * 2 * p.Ncols;
        dst[n0 + tid] = 0.f;
        dst[p.Ncols + n0 + tid] = 0.f;
      }
      return;
    }
    __syncthreads();  // rowpos
    constexpr int CPRz = BN / 8;
    for (int idx = tid; idx < BM * CPRz; idx += 256) {
      const int row = idx / CPRz, c8 = idx - row * CPRz;
      const int m = rowpos[row], n = n0 + c8 * 8;
      if (m >= 0 && n < p.Ncols) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (p.flags & VS_CONV_RESIDUAL) {
          v = *(const uint4*)(p.res + (long long)m * p.res_ld + n);
          if (p.res_bits) v = mask8_bf16(v, p.res_bits[(long long)m * (p.Ncols >> 3) + (n >> 3)]);
        }
        *(uint4*)(p.y + (long long)m * p.y_ld + n) = v;
      }
    }
    if (BNB && tid < BN && n0 + tid < p.Ncols) {  // dz = 0: the tile adds nothing
      float* dst = p.stats + (long long)tm * 2 * p.Ncols;
      dst[n0 + tid] = 0.f;
      dst[p.Ncols + n0 + tid] = 0.f;
    }
    return;
  }
  const int nk_all = (Keff + 63) >> 6;
  int kbeg = 0, nk = nk_all;  // this block's k-steps
  if (p.splitK != 1) {
    kbeg = (int)((unsigned)(nk_all * ksplit) / (unsigned)p.splitK);
    nk = (int)((unsigned)(nk_all * (ksplit + 1)) / (unsigned)p.splitK) - kbeg;
  }
  __syncthreads();  // ktab (and the row table) visible
  if (use_rowtab) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const uint2 e = rowtab[lrow + 32 * i];
      roff[i] = e.x;
      vmask[i] = e.y;
    }
    // (the epilogue's first statistic writes come behind the main loop's barriers: no reader of the table is left by then)
  }
  VS_ST(vst, 1);
#ifdef VS_STAMP
  vst.t[7] = (unsigned long long)nk;
  unsigned long long vs_first = ~0ull;
#endif
  if constexpr (NS > 0) {
    // LDS-DMA ring.  One wave-instruction copies 64 x 16 B = eight 128-byte tile rows straight
    // into LDS (destination = wave-uniform base + lane * 16, no VGPRs, no ds_write); the XOR
    // swizzle of the image goes on the SOURCE: the thread that fills physical unit kc of row r
    // fetches logical unit kc ^ ((r >> 1) & 7).  Every thread issues exactly L loads per tile
    // (masked ones point out of range: the buffer range check writes zeros), so a counted
    // vmcnt retires tile kt while tiles kt+1 .. kt+D-1 stay in flight across the raw barrier.
    constexpr int D = NS - 1;
    constexpr int L = AI + BJ;
    static_assert((D - 1) * L <= 63, "vmcnt range");
    const int kce = kc ^ ((lrow >> 1) & 7);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The copies are issued from inline asm: a builtin LDS-DMA makes hipcc treat every later
    // ds_read as dependent on it and wait vmcnt(0), which serialises the ring.  hipcc has no
    // VMEM of its own between the first copy and the explicit vmcnt(0) that ends the loop, and
    // it does not use M0 in this kernel (written and read inside one statement).
    typedef __attribute__((address_space(3))) char* lds_ptr_t;
    const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem + (unsigned)wv * 1024u;
    auto rsrc_words = [](const void* base, unsigned bytes) __attribute__((always_inline)) {
      const unsigned long a = (unsigned long)base;
      return (i32x4){(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
    };
    const i32x4 xdesc = rsrc_words(p.x, p.x_bytes), wdesc = rsrc_words(p.w, p.w_bytes);
    auto dma16 = [](const i32x4& desc, unsigned lds_addr, unsigned voff) __attribute__((always_inline)) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                   :
                   : "s"(lds_addr), "v"(voff), "s"(desc)
                   : "memory");
    };
    // (e: the k-table entry of (kt, this thread's unit) for the gathering modes -- handed in, so that the ring loop can read
    //  the NEXT k-step's entry behind this step's copies instead of in front of them: one LDS round trip per k-step off
    //  the copy-issue path)
    auto dma_e = [&](int kt, int stage, int2 e) __attribute__((always_inline)) {
      const int k8 = kt * 8 + kce;
      const bool kval = k8 < K8;
      if (MODE == 0) e = make_int2(0, k8 * 16);
      const unsigned kbit = (MODE == 0) ? (unsigned)kval : 1u;
      const unsigned wk = (MODE == 0) ? (unsigned)k8 : ((unsigned)e.x >> 5);
      const unsigned A = lds0 + (unsigned)(stage * STAGE);
      const unsigned B = A + BM * 128;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const unsigned ok = kbit & (vmask[i] >> (e.x & 31)) & 1u;
        const unsigned off = ok ? roff[i] + (unsigned)e.y : VS_OOB;
        dma16(xdesc, A + i * 4096, off);
      }
#pragma unroll
      for (int j = 0; j < BJ; ++j) {
        const unsigned ok = (unsigned)kval & (unsigned)(boff[j] != VS_OOB);
        const unsigned off = ok ? boff[j] + wk * 16u : VS_OOB;
        dma16(wdesc, B + j * 4096, off);
      }
    };
    auto dma = [&](int kt, int stage) __attribute__((always_inline)) {
      int2 e = make_int2(0, 0);
      if (MODE != 0) e = ktab[kt * 8 + kce];
      dma_e(kt, stage, e);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) dma(kbeg + d, d);
    int2 e_next = make_int2(0, 0);
    if (MODE != 0) e_next = ktab[(kbeg + D) * 8 + kce];
    int st_c = 0, st_l = D;  // stage computed / stage refilled this step
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"((D - 1) * L) : "memory");  // my part of tile kt landed
      __builtin_amdgcn_s_barrier();  // everyone's did; stage st_l (tile kt-1) is no longer read
      __builtin_amdgcn_sched_barrier(0);
#ifdef VS_STAMP
      { const unsigned long long now_ = vs_now(); vs_first = now_ < vs_first ? now_ : vs_first; }  // (branch-free: the first k-step's)
#endif
      // (one straight-line body: a second, "fragment reads first" order of this step lived here behind a run-time
      //  switch until round 4 -- measured useless in round 2 (profiles/r02_ring_frags_first.txt), and its mere presence
      //  made hipcc move all 64 accumulator registers at the top of every k-step)
      dma_e(kbeg + kt + D, st_l, e_next);
      if (MODE != 0) e_next = ktab[(kbeg + kt + D + 1) * 8 + kce];  // (the table is padded one k-step past the ring's run-ahead)
      __builtin_amdgcn_sched_barrier(0);
      if (AOL) aol_k0 = (kbeg + kt) * 64;
      compute(st_c);
      __builtin_amdgcn_sched_barrier(0);
      st_c = (st_c + 1 == NS) ? 0 : st_c + 1;
      st_l = (st_l + 1 == NS) ? 0 : st_l + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // run-ahead tiles (never read) before LDS reuse
    __syncthreads();
#ifdef VS_STAMP
    vst.t[2] = vs_first;
#endif
  } else {
  // Software pipeline, prefetch distance 2: while tile kt is multiplied, tile kt+1 sits in one
  // register set (loaded during step kt-1, written to LDS at the end of step kt) and the loads
  // of tile kt+2 are issued into the other set.  With one block per CU a single tile in flight
  // left every k-step waiting out a full memory latency.  Loads past the last tile of the
  // block's range are real data of the next split (or K-tail zeros) and are never stored.
  // Loop body is branch-free (unrolled by 2).
  gload(kbeg, ra0, rb0);
  sstore(0, ra0, rb0);
  gload(kbeg + 1, ra1, rb1);
  __syncthreads();
  VS_ST(vst, 2);
  int kt = 0;
  for (; kt + 2 < nk; kt += 2) {
    gload(kbeg + kt + 2, ra0, rb0);
    __builtin_amdgcn_sched_barrier(0);
    compute(0);
    __builtin_amdgcn_sched_barrier(0);
    sstore(1, ra1, rb1);  // tile kt+1
    __syncthreads();
    gload(kbeg + kt + 3, ra1, rb1);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    __builtin_amdgcn_sched_barrier(0);
    sstore(0, ra0, rb0);  // tile kt+2
    __syncthreads();
  }
  // remaining: nk - kt is 1 or 2; buffer 0 holds tile kt, set 1 holds tile kt+1 (if any)
  compute(0);
  if (kt + 1 < nk) {
    sstore(1, ra1, rb1);
    __syncthreads();
    compute(1);
  }
  __syncthreads();
  }

  VS_ST(vst, 3);
  bool in_launch = false;
  if constexpr (NS >= 2 && !AOL) {
    if (p.splitK > 1 && p.sk_cnt != nullptr) {
      // In-launch split-K: the S blocks of a tile (neighbours in the block order, same XCD run) each store their fp32
      // partial accumulators -- in register order, one 4-KiB row of float4 per (a, b) fragment, nothing staged -- and
      // take a ticket from the tile's arrival counter.  The block that draws ticket S - 1 sums the partials of splits
      // 0 .. S - 1 IN THAT ORDER (its own included, re-read: the sum does not depend on who arrived last, so the result
      // is bitwise the same from run to run), clears the counter for the next launch and runs the fused epilogue.
      // sk_cnt has to be zero before the first launch (vs_conv_workspace_bytes).
      in_launch = true;
      // Partials and ticket without a cache-wide fence: an agent-scope release / acquire fence on gfx950 is
      // buffer_wbl2 + buffer_inv over the XCD's whole L2 (~100 us per launch, measured: every split launch took
      // 100-135 us whatever its size).  Instead every partial word is an agent-scope relaxed atomic store / load
      // (global_store / global_load with sc1: written through to, and read from, the point the XCDs share); an explicit
      // s_waitcnt vmcnt(0) in every wave retires the stores before the barrier behind which the ticket is drawn.
      constexpr int Q = MR * NR;
      float* part = p.slab + ((long long)tile_id * p.splitK) * (Q * 1024);  // [split][fragment][component][256 threads]
      float* mine = part + (long long)ksplit * (Q * 1024) + tid;
#pragma unroll
      for (int a = 0; a < MR; ++a)
#pragma unroll
        for (int b = 0; b < NR; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __hip_atomic_store(mine + ((a * NR + b) * 4 + r) * 256, acc[a][b][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // Retire this wave's partial stores BEFORE the barrier: a workgroup-scope barrier on gfx950 does not wait for
      // vmcnt outside tgsplit mode (the ISA of this kernel showed `s_waitcnt vmcnt(63)` in front of s_barrier), so
      // without the explicit wait the ticket could be drawn -- and the last arriver start summing -- while partial
      // words were still in flight.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int* flag = (int*)smem;
      if (tid == 0) flag[0] = atomicAdd(p.sk_cnt + tile_id, 1);
      __syncthreads();
      const int ticket = flag[0];
      if (ticket != p.splitK - 1) return;
      if (tid == 0) p.sk_cnt[tile_id] = 0;
      auto load_part = [&](int s, int q) __attribute__((always_inline)) {
        const float* ps = part + (long long)s * (Q * 1024) + q * 1024 + tid;
        f32x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = __hip_atomic_load(ps + r * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return t;
      };
      // (whole-vector assignments and adds: with per-component updates of acc here, hipcc treated three of the four
      //  lanes of every accumulator as undefined in the MAIN LOOP -- three quarters of every ring launch's outputs garbage)
#pragma unroll
      for (int a = 0; a < MR; ++a)
#pragma unroll
        for (int b = 0; b < NR; ++b) acc[a][b] = load_part(0, a * NR + b);
      for (int s = 1; s < p.splitK; ++s) {
#pragma unroll
        for (int a = 0; a < MR; ++a)
#pragma unroll
          for (int b = 0; b < NR; ++b) acc[a][b] = acc[a][b] + load_part(s, a * NR + b);
      }
      __syncthreads();  // flag word: the epilogue reuses the tile memory
    }
  }
  if (p.splitK > 1 && !in_launch) {
    // split-K: raw fp32 partial tile -> slab[ksplit]; BN statistics, affine, residual, ReLU and
    // the bf16 store happen in conv_splitk_epilogue_kernel after the fixed-order slab sum
    float* E = (float*)smem;
#pragma unroll
    for (int b = 0; b < NR; ++b) {
      const int col = wn * TN + b * 16 + lr;
#pragma unroll
      for (int a = 0; a < MR; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) E[(wm * TM + a * 16 + lq * 4 + r) * BN + col] = acc[a][b][r];
    }
    __syncthreads();
    float* dst = p.slab + (long long)ksplit * p.M * p.Ncols;
    constexpr int QPR = BN / 4;
    for (int idx = tid; idx < BM * QPR; idx += 256) {
      const int row = idx / QPR, c4 = idx - row * QPR;
      const int m = (MODE == 2) ? rowpos[row] : (m0 + row < p.M ? m0 + row : -1), n = n0 + c4 * 4;
      if (m >= 0 && n < p.Ncols)
        *(float4*)(dst + (long long)m * p.Ncols + n) = *(const float4*)(E + row * BN + c4 * 4);
    }
    return;
  }

  // ---------------- epilogue ----------------
#ifdef VS_STAMP
  VsStamp* const stp_ = &vst;
#else
  VsStamp* const stp_ = nullptr;
#endif
  conv_tile_epilogue<BM, BN, WM, WN, BNB, false, BNB2>(p, acc, smem, statbuf, tm, n0, [&](int row) {
    return (MODE == 2) ? rowpos[row] : (m0 + row < p.M ? m0 + row : -1);
  }, stp_);
  VS_ST(vst, 6);
  VS_ST_FLUSH(p, blk, vst);
}

template <int BM, int BN, int WM, int WN, int MODE, bool FAST, int DBG = 0, int NS = 0, bool BNB = false,
          bool BNB2 = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p) {
  conv_igemm_body<BM, BN, WM, WN, MODE, FAST, DBG, NS, BNB, BNB2>(p, blockIdx.x, gridDim.x);
}

// Apply on load (ConvP::in_scale): the pointwise ring launch whose A fragments go through the producer's BN + ReLU.
template <int BM, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__(256) void conv_igemm_aol_kernel(ConvP p) {
  conv_igemm_body<BM, BN, WM, WN, 0, true, 0, NS, false, false, true>(p, blockIdx.x, gridDim.x);
}

// Sum of the split-K slabs (fixed order) + the whole fused epilogue.  Block = 64 rows x all
// columns; thread = one 8-channel chunk column x (256 / chunks) row lanes.
__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(
    const float* slab, int S, uint16_t* y, const float* scale, const float* shift,
    const uint16_t* res, float* stats, int M, int N, int y_ld, int res_ld, int flags) {
  __shared__ float red[2][256][8];
  const int cpr = N >> 3;
  const int ncol = cpr < 256 ? cpr : 256;
  const int rl = 256 / ncol;
  const int col = threadIdx.x % ncol, lane_r = threadIdx.x / ncol;
  const int r0 = blockIdx.x * 64;
  for (int cb = col; cb < cpr; cb += ncol) {
    const int c = cb * 8;
    float sg[8], sq[8], sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sg[e] = sq[e] = 0.f;
      sc[e] = (flags & VS_CONV_AFFINE) ? scale[c + e] : 1.f;
      sh[e] = (flags & VS_CONV_AFFINE) ? shift[c + e] : 0.f;
    }
    for (int r = lane_r; r < 64 && lane_r < rl; r += rl) {  // (256 % ncol leftover threads idle)
      const int m = r0 + r;
      if (m >= M) break;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
      for (int s = 0; s < S; ++s) {
        const float* src = slab + ((long long)s * M + m) * N + c;
        const float4 a = *(const float4*)src, b = *(const float4*)(src + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
        v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sg[e] += v[e];
        sq[e] += v[e] * v[e];
        v[e] = v[e] * sc[e] + sh[e];
      }
      if (flags & VS_CONV_RESIDUAL) {
        float rf[8];
        unpack8_bf16(*(const uint4*)(res + (long long)m * res_ld + c), rf);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rf[e];
      }
      if (flags & VS_CONV_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      *(uint4*)(y + (long long)m * y_ld + c) = pack8_bf16(v);
    }
    if (flags & VS_CONV_STATS) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[0][threadIdx.x][e] = sg[e];
        red[1][threadIdx.x][e] = sq[e];
      }
      __syncthreads();
      if (lane_r == 0) {
        for (int r = 1; r < rl; ++r) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            sg[e] += red[0][r * ncol + col][e];
            sq[e] += red[1][r * ncol + col][e];
          }
        }
        float* dst = stats + (long long)blockIdx.x * 2 * N;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          dst[c + e] = sg[e];
          dst[N + c + e] = sq[e];
        }
      }
    }
  }
}

// =============================================================================
// Direct kernel for small-channel layers (fast pathway: N <= 32, K <= 192, taps <= 31).
// Those GEMMs are a few MFMAs per 16 rows, so the tiled kernel above is all prologue /
// barrier / epilogue.  Here there is no block-level cooperation at all: the whole weight
// matrix sits in registers as B fragments, every lane fetches its A fragment (16 bytes =
// 8 channels of one tap of its row) straight into the MFMA operand with a bounds-checked
// buffer_load, each wave streams `tiles_per_wave` consecutive 16-row tiles, and the only
// LDS use is a 512-byte per-wave transpose so that stores are whole channel vectors.
// =============================================================================
// Row decode of the register-resident kernels: output position m -> byte offset of its tap-0 input row and the
// bitmask of taps that fall inside the input (MODE 0: pointwise, 1: forward / unit-stride dgrad, 2: strided dgrad).
template <int MODE>
__device__ __forceinline__ void direct_decode_row(const ConvP& p, int m, float rcpW, float rcpH, float rcpT,
                                                  unsigned& roff, unsigned& vmask) {
  int rw, t1, rh, t2, rt, n;
  fast_divmod(m, p.Rw, rcpW, t1, rw);
  fast_divmod(t1, p.Rh, rcpH, t2, rh);
  fast_divmod(t2, p.Rt, rcpT, n, rt);
  if (MODE == 0) {
    const long long pos =
        ((long long)(n * p.Gt + rt * p.mulT) * p.Gh + rh * p.mulH) * p.Gw + rw * p.mulW;
    roff = (unsigned)(pos * p.g_ld * 2);
    vmask = 1u;
  } else {
    const int ct = rt * p.mulT + p.offT, ch = rh * p.mulH + p.offH, cw = rw * p.mulW + p.offW;
    long long pos0 = (long long)n * p.Gt * p.Gh * p.Gw;
    if (MODE == 1) pos0 += ((long long)ct * p.Gh + ch) * p.Gw + cw;
    else pos0 += ((long long)(ct >> p.shT) * p.Gh + (ch >> p.shH)) * p.Gw + (cw >> p.shW);
    roff = (unsigned)(pos0 * p.g_ld * 2);
    auto axis_mask = [&](int c, int kk, int shf, int G) {
      unsigned mm = 0u;
      for (int dd = 0; dd < kk; ++dd) {
        int v = c + p.tmul * dd;
        bool ok = true;
        if (MODE == 2) {
          ok = (v & ((1 << shf) - 1)) == 0;
          v >>= shf;
        }
        ok = ok && ((unsigned)v < (unsigned)G);
        mm |= (ok ? 1u : 0u) << dd;
      }
      return mm;
    };
    const unsigned mt = axis_mask(ct, p.kT, p.shT, p.Gt), mh = axis_mask(ch, p.kH, p.shH, p.Gh),
                   mw = axis_mask(cw, p.kW, p.shW, p.Gw);
    int tap = 0;
    unsigned vm = 0u;
    for (int dt = 0; dt < p.kT; ++dt)
      for (int dh = 0; dh < p.kH; ++dh) {
        const unsigned th = (mt >> dt) & (mh >> dh) & 1u;
        vm |= (th ? mw : 0u) << tap;
        tap += p.kW;
      }
    vmask = vm;
  }
}

// The whole weight matrix of a register-resident kernel as MFMA B fragments, and per k-step the tap its 8-channel
// group belongs to (ktap; 31 = none) with the byte offset of that group from a row's tap-0 address (kdel).
template <int NT, int KS, int MODE>
__device__ __forceinline__ void direct_load_weights(const ConvP& p, const __amdgpu_buffer_rsrc_t wsrc, int lq, int lr,
                                                    bf16x8 (&bfr)[KS][NT], int (&ktap)[KS], int (&kdel)[KS]) {
  const int K8 = p.K >> 3, C8 = p.Cg >> 3;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int k8 = ks * 4 + lq;
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int n = b * 16 + lr;
      const unsigned off = (k8 < K8 && n < p.Ncols) ? (unsigned)((n * p.K + k8 * 8) * 2) : VS_OOB;
      bfr[ks][b] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wsrc, off, 0, 0));
    }
    ktap[ks] = 31;  // never valid
    kdel[ks] = 0;
    if (k8 < K8) {
      if (MODE == 0) {
        ktap[ks] = 0;
        kdel[ks] = k8 * 16;
      } else {
        const int tap = k8 / C8, c8 = k8 - tap * C8;
        const int dw = tap % p.kW, t2 = tap / p.kW;
        const int dh = t2 % p.kH, dt = t2 / p.kH;
        long long dpos;
        if (MODE == 1) dpos = (((long long)dt * p.Gh + dh) * p.Gw + dw) * p.tmul;
        else dpos = -(((long long)(dt >> p.shT) * p.Gh + (dh >> p.shH)) * p.Gw + (dw >> p.shW));
        ktap[ks] = tap;
        kdel[ks] = (int)((dpos * p.g_ld + c8 * 8) * 2);
      }
    }
  }
}

// TB = 16-row tiles a wave keeps in flight: the gathers of all TB tiles are issued before the first MFMA.  One
// tile per round trip moved 256..1024 unique bytes per wave per memory latency (~1 TB/s over the chip for the
// 8-channel layers of the fast pathway); see profiles/r02_direct_tb.txt for the sweep.
// BNB (dgrad launches): the copy-out also emits the BN-backward sums of the unit this dx is the complete dz of --
// the two variants of conv_tile_epilogue (no residual: mask recomputed from gamma / beta; residual: the unit's bit
// mask) -- one partial row per block (fixed order: a lane's rows in order, then waves x lanes through LDS).
template <int NT, int KS, int MODE, int TB, bool BNB = false>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvP p, int tiles_per_wave) {
  __shared__ __attribute__((aligned(16))) uint16_t tbuf[4][TB * 16 * 16 * NT];
  __shared__ float sstat[2][4][16 * NT];
  __shared__ float bnred[BNB ? 4 * 64 * 16 : 1];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lq = lane >> 4;
  const __amdgpu_buffer_rsrc_t xsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);

  // B fragments: B[k = ks*32 + 8*lq + j][n = nt*16 + lr], resident for the whole kernel
  bf16x8 bfr[KS][NT];
  int ktap[KS], kdel[KS];
  direct_load_weights<NT, KS, MODE>(p, wsrc, lq, lr, bfr, ktap, kdel);
  const float rcpW = 1.0f / (float)p.Rw, rcpH = 1.0f / (float)p.Rh, rcpT = 1.0f / (float)p.Rt;
  float ssum[NT], ssq[NT];
#pragma unroll
  for (int b = 0; b < NT; ++b) ssum[b] = ssq[b] = 0.f;
  float sc[NT], sh[NT];
#pragma unroll
  for (int b = 0; b < NT; ++b) {
    const int col = b * 16 + lr;
    sc[b] = 1.f;
    sh[b] = 0.f;
    if ((p.flags & VS_CONV_AFFINE) && col < p.Ncols) {
      sc[b] = p.scale[col];
      sh[b] = p.shift[col];
    }
  }
  const bool relu = (p.flags & VS_CONV_RELU) != 0, has_res = (p.flags & VS_CONV_RESIDUAL) != 0;
  // BNB: this lane's fixed 8-channel chunk (64 % chunks-per-row == 0) and the unit's per-channel constants
  constexpr int CPRW_ = 2 * NT;
  const int bn_c8 = lane % CPRW_, bn_n = bn_c8 * 8;
  const bool bn_ok = BNB && bn_n < p.Ncols;
  float bmu[8], bis[8], bga[8], bbe[8], bsg[8], bsx[8];
  if (BNB) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = bn_ok ? bn_n + e : 0;
      bmu[e] = p.bn_mean[c];
      bis[e] = p.bn_invstd[c];
      bga[e] = has_res ? 0.f : p.bn_gamma[c];
      bbe[e] = has_res ? 0.f : p.bn_beta[c];
      bsg[e] = 0.f;
      bsx[e] = 0.f;
    }
  }
  uint16_t* tb = tbuf[wave];
  const int tile0 = (blockIdx.x * 4 + wave) * tiles_per_wave;

  for (int it = 0; it < tiles_per_wave; it += TB) {
    if ((tile0 + it) * 16 >= p.M) break;
    // Row decode ONCE per row: lane l decodes row l of the batch (TB * 16 <= 64 rows: position -> (clip, t, h, w),
    // base byte offset, bitmask of valid taps -- ~80 vector instructions), and the 4 lanes x TB tiles that need a
    // row fetch its two words with a cross-lane read.  Decoding per (tile, lane) repeated every row four times
    // over and made the kernel instruction-issue bound.
    unsigned roff_own = VS_OOB, vmask_own = 0u;
    {
      const int m = (tile0 + it) * 16 + lane;
      if (lane < TB * 16 && m < p.M && it + (lane >> 4) < tiles_per_wave)
        direct_decode_row<MODE>(p, m, rcpW, rcpH, rcpT, roff_own, vmask_own);
    }
    unsigned roff[TB], vmask[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      roff[u] = (unsigned)__shfl((int)roff_own, u * 16 + lr, 64);
      vmask[u] = (unsigned)__shfl((int)vmask_own, u * 16 + lr, 64);
    }
    bf16x8 af[TB][KS];
#pragma unroll
    for (int u = 0; u < TB; ++u)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const unsigned ok = (vmask[u] >> ktap[ks]) & 1u;
        const unsigned off = ok ? roff[u] + (unsigned)kdel[ks] : VS_OOB;
        af[u][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xsrc, off, 0, 0));
      }
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      f32x4 acc[NT];
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int b = 0; b < NT; ++b)
          acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u][ks], bfr[ks][b], acc[b], 0, 0, 0);
      // D[row = lq*4 + r][col = lr]; rows beyond M were zero-filled
#pragma unroll
      for (int b = 0; b < NT; ++b) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = acc[b][r];
          ssum[b] += a;
          ssq[b] += a * a;
          float v = a * sc[b] + sh[b];
          if (relu && !has_res) v = fmaxf(v, 0.f);
          tb[(u * 16 + lq * 4 + r) * (16 * NT) + b * 16 + lr] = f32_to_bf16(v);
        }
      }
    }
    // the wave's own (TB*16) x (16*NT) bf16 tile -> 16-byte channel vectors (wave-local LDS hand-off)
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's ds_writes have landed
    __builtin_amdgcn_wave_barrier();
    constexpr int CPRW = 2 * NT;  // chunks per row
    const int m0 = (tile0 + it) * 16;
    const int nrows = 16 * min(TB, tiles_per_wave - it);
#pragma unroll
    for (int c = lane; c < TB * 16 * CPRW; c += 64) {
      const int row = c / CPRW, c8 = c - row * CPRW;
      const int mm = m0 + row, n = c8 * 8;
      if (row < nrows && mm < p.M && n < p.Ncols) {
        uint4 v = *(const uint4*)(tb + row * (16 * NT) + c8 * 8);
        if (has_res) {
          float f[8], g[8];
          unpack8_bf16(v, f);
          unpack8_bf16(*(const uint4*)(p.res + (long long)mm * p.res_ld + n), g);
          if (p.res_bits) mask8(g, p.res_bits[(long long)mm * (p.Ncols >> 3) + (n >> 3)]);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            f[e] += g[e];
            if (relu) f[e] = fmaxf(f[e], 0.f);
          }
          v = pack8_bf16(f);
        }
        *(uint4*)(p.y + (long long)mm * p.y_ld + n) = v;
        if (BNB) {  // the sums see dz as stored
          float gq[8], yv[8];
          unpack8_bf16(v, gq);
          unpack8_bf16(*(const uint4*)(p.bny + (long long)mm * p.bny_ld + n), yv);
          const unsigned bits = has_res ? (unsigned)p.bn_bits[(long long)mm * (p.Ncols >> 3) + (n >> 3)] : 0u;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xh = (yv[e] - bmu[e]) * bis[e];
            const bool on = has_res ? ((bits >> e) & 1u) != 0 : (xh * bga[e] + bbe[e]) > 0.f;
            const float ge = on ? gq[e] : 0.f;
            bsg[e] += ge;
            bsx[e] += ge * xh;
          }
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();  // reads done before the next batch overwrites tb
  }
  if (BNB) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bnred[(wave * 64 + lane) * 16 + e] = bsg[e];
      bnred[(wave * 64 + lane) * 16 + 8 + e] = bsx[e];
    }
    __syncthreads();
    if (tid < p.Ncols) {  // fixed order: waves, then the lanes that own this channel's chunk
      const int cc = tid >> 3, e = tid & 7;
      float ts = 0.f, tq = 0.f;
      for (int w = 0; w < 4; ++w)
        for (int l = cc; l < 64; l += CPRW_) {
          ts += bnred[(w * 64 + l) * 16 + e];
          tq += bnred[(w * 64 + l) * 16 + 8 + e];
        }
      float* dst = p.stats + (long long)blockIdx.x * 2 * p.Ncols;
      dst[tid] = ts;
      dst[p.Ncols + tid] = tq;
    }
    return;
  }
  if (p.flags & VS_CONV_STATS) {
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      float s = ssum[b], q = ssq[b];
      s += __shfl_xor(s, 16, 64);
      q += __shfl_xor(q, 16, 64);
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 32, 64);
      if (lq == 0) {
        sstat[0][wave][b * 16 + lr] = s;
        sstat[1][wave][b * 16 + lr] = q;
      }
    }
    __syncthreads();
    if (tid < p.Ncols) {
      float* dst = p.stats + (long long)blockIdx.x * 2 * p.Ncols;
      dst[tid] = sstat[0][0][tid] + sstat[0][1][tid] + sstat[0][2][tid] + sstat[0][3][tid];
      dst[p.Ncols + tid] = sstat[1][0][tid] + sstat[1][1][tid] + sstat[1][2][tid] + sstat[1][3][tid];
    }
  }
}

// 16-row tiles per wave (x 4 waves x 16 rows = rows per block): 8 (512 rows).  VS_DIRECT_TPW_BLOCKS=N (experiment, round 4):
// fewer tiles per wave where 8 leave fewer than N blocks -- measured neutral at N = 512 over the small-channel layers
// (0.316 vs 0.317 ms forward, 200 704-row layers with 784 instead of 392 blocks), 10 % slower at 1024, and 0.4 % slower
// in the step (more BN partial rows): off by default (profiles/r04_direct_tpw.txt).
static int direct_tpw(long long M) {
  static const long long want = [] { const char* e = getenv("VS_DIRECT_TPW_BLOCKS"); return e ? atoll(e) : 0ll; }();
  int tpw = 8;
  while (want > 0 && tpw > 2 && (M + 64 * tpw - 1) / (64 * tpw) < want) tpw >>= 1;
  return tpw;
}
#define VS_DIRECT_TPW direct_tpw(p.M)

static int direct_blocks(long long M) {
  const int tpw = direct_tpw(M);
  return (int)((M + 16 * 4 * tpw - 1) / (16 * 4 * tpw));
}

template <int NT, int KS, int TB>
static int launch_direct_tb(const ConvP& p, int mode, hipStream_t st) {
  const int grid = direct_blocks(p.M);
  if (mode == 0)
    hipLaunchKernelGGL((conv_direct_kernel<NT, KS, 0, TB>), dim3(grid), dim3(256), 0, st, p, VS_DIRECT_TPW);
  else if (mode == 1)
    hipLaunchKernelGGL((conv_direct_kernel<NT, KS, 1, TB>), dim3(grid), dim3(256), 0, st, p, VS_DIRECT_TPW);
  else
    hipLaunchKernelGGL((conv_direct_kernel<NT, KS, 2, TB>), dim3(grid), dim3(256), 0, st, p, VS_DIRECT_TPW);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// tiles in flight per wave: VS_DIRECT_TB=1|2|4 overrides the rule (A/B runs)
static int direct_tb(int nt, int ks) {
  static const int forced = [] {
    const char* e = getenv("VS_DIRECT_TB");
    return e ? atoi(e) : 0;
  }();
  if (forced == 1 || forced == 2 || forced == 4) return forced;
  return (nt * ks <= 6) ? 4 : 2;
}

template <int NT, int KS>
static int launch_direct_ks(const ConvP& p, int mode, hipStream_t st) {
  if (p.flags & VS_CONV_BNBWD) {  // dgrad + the producer's BN-backward sums: two tiles in flight (register budget)
    const int grid = direct_blocks(p.M);
    if (mode == 0)
      hipLaunchKernelGGL((conv_direct_kernel<NT, KS, 0, 2, true>), dim3(grid), dim3(256), 0, st, p, VS_DIRECT_TPW);
    else if (mode == 1)
      hipLaunchKernelGGL((conv_direct_kernel<NT, KS, 1, 2, true>), dim3(grid), dim3(256), 0, st, p, VS_DIRECT_TPW);
    else
      hipLaunchKernelGGL((conv_direct_kernel<NT, KS, 2, 2, true>), dim3(grid), dim3(256), 0, st, p, VS_DIRECT_TPW);
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  switch (direct_tb(NT, KS)) {
    case 1: return launch_direct_tb<NT, KS, 1>(p, mode, st);
    case 2: return launch_direct_tb<NT, KS, 2>(p, mode, st);
    default: return launch_direct_tb<NT, KS, 4>(p, mode, st);
  }
}

template <int NT>
static int launch_direct(const ConvP& p, int mode, hipStream_t st) {
  switch ((p.K + 31) / 32) {
    case 1: return launch_direct_ks<NT, 1>(p, mode, st);
    case 2: return launch_direct_ks<NT, 2>(p, mode, st);
    case 3: return launch_direct_ks<NT, 3>(p, mode, st);
    case 4: return launch_direct_ks<NT, 4>(p, mode, st);
    case 5: return launch_direct_ks<NT, 5>(p, mode, st);
    default: return launch_direct_ks<NT, 6>(p, mode, st);
  }
}


// =============================================================================
// Evaluation only: conv b and conv c of a fast-pathway bottleneck in ONE launch (SURVEY.md 7 "Layout": the 8 / 16
// channel inner tensor never leaves the CU).  Stage 1 is the register-resident kernel above for conv b (folded BN +
// ReLU, bf16 like the stored tensor would be); the wave's transposed b tile in LDS IS the A operand of conv c: lane
// (row lr, k-group lq) reads its 16 bytes = 8 inner channels of its row, one v_mfma_f32_16x16x32_bf16 per 16 output
// channels of c against conv c's weights (<= 8 fragments, resident), then folded BN + residual + ReLU in fp32 (one
// rounding, as conv_tile_epilogue does) through a per-wave fp32 LDS tile so that stores are whole channel vectors.
// The residual rows of all TB tiles are fetched with the gathers, before the first MFMA.
// =============================================================================
struct BcP {
  const uint16_t* w2;   // conv c weights [N2][K2 = conv b's Cout] bf16
  unsigned w2_bytes;
  int N2;               // conv c's output channels (<= 16 * NT2)
  const float* scale2;  // folded BN of conv c
  const float* shift2;
  const uint16_t* res;  // block input / shortcut rows [M][res_ld], or null
  int res_ld;
  uint16_t* y2;         // block output rows [M][y2_ld]
  int y2_ld;
  int relu2;
};

template <int NT, int KS, int MODE, int TB, int NT2>
__global__ __launch_bounds__(256) void conv_direct_bc_kernel(ConvP p, BcP q, int tiles_per_wave) {
  constexpr int S2 = 16 * NT2 + 4;   // fp32 row pitch of the c tile (+4: the four row groups land in two bank halves)
  constexpr int CPR2 = 2 * NT2;      // 8-channel chunks per output row
  constexpr int CH2 = (16 * CPR2 + 63) / 64;  // chunks per lane and 16-row tile
  __shared__ __attribute__((aligned(16))) uint16_t tbuf[4][TB * 16 * 16 * NT];
  __shared__ __attribute__((aligned(16))) float tbuf2[4][16 * S2];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lq = lane >> 4;
  const __amdgpu_buffer_rsrc_t xsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2src =
      __builtin_amdgcn_make_buffer_rsrc((void*)q.w2, 0, (int)q.w2_bytes, 0x00020000);

  bf16x8 bfr[KS][NT];
  int ktap[KS], kdel[KS];
  direct_load_weights<NT, KS, MODE>(p, wsrc, lq, lr, bfr, ktap, kdel);
  // conv c: B2[k = 8*lq + j][n = b*16 + lr], K2 = p.Ncols <= 16 * NT (one k-step of 32, zero beyond K2)
  bf16x8 w2fr[NT2];
  float sc2[NT2], sh2[NT2];
#pragma unroll
  for (int b = 0; b < NT2; ++b) {
    const int n = b * 16 + lr;
    const unsigned off = (lq * 8 < p.Ncols && n < q.N2) ? (unsigned)((n * p.Ncols + lq * 8) * 2) : VS_OOB;
    w2fr[b] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w2src, off, 0, 0));
    sc2[b] = n < q.N2 ? q.scale2[n] : 0.f;
    sh2[b] = n < q.N2 ? q.shift2[n] : 0.f;
  }
  const float rcpW = 1.0f / (float)p.Rw, rcpH = 1.0f / (float)p.Rh, rcpT = 1.0f / (float)p.Rt;
  float sc[NT], sh[NT];
#pragma unroll
  for (int b = 0; b < NT; ++b) {
    const int col = b * 16 + lr;
    sc[b] = 1.f;
    sh[b] = 0.f;
    if (col < p.Ncols) {
      sc[b] = p.scale[col];
      sh[b] = p.shift[col];
    }
  }
  const bool relu = (p.flags & VS_CONV_RELU) != 0;
  uint16_t* tb = tbuf[wave];
  float* tb2 = tbuf2[wave];
  const int tile0 = (blockIdx.x * 4 + wave) * tiles_per_wave;

  for (int it = 0; it < tiles_per_wave; it += TB) {
    if ((tile0 + it) * 16 >= p.M) break;
    const int m0 = (tile0 + it) * 16;
    unsigned roff_own = VS_OOB, vmask_own = 0u;
    {
      const int m = m0 + lane;
      if (lane < TB * 16 && m < p.M && it + (lane >> 4) < tiles_per_wave)
        direct_decode_row<MODE>(p, m, rcpW, rcpH, rcpT, roff_own, vmask_own);
    }
    unsigned roff[TB], vmask[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      roff[u] = (unsigned)__shfl((int)roff_own, u * 16 + lr, 64);
      vmask[u] = (unsigned)__shfl((int)vmask_own, u * 16 + lr, 64);
    }
    bf16x8 af[TB][KS];
#pragma unroll
    for (int u = 0; u < TB; ++u)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const unsigned ok = (vmask[u] >> ktap[ks]) & 1u;
        const unsigned off = ok ? roff[u] + (unsigned)kdel[ks] : VS_OOB;
        af[u][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xsrc, off, 0, 0));
      }
    // the residual chunks this lane adds in the copy-out of every tile of the batch
    const int nrows = 16 * min(TB, tiles_per_wave - it);
    uint4 resv[TB][CH2];
#pragma unroll
    for (int u = 0; u < TB; ++u)
#pragma unroll
      for (int j = 0; j < CH2; ++j) {
        const int c = lane + j * 64;
        const int row = c / CPR2, n = (c - row * CPR2) * 8;
        const int mm = m0 + u * 16 + row;
        resv[u][j] = make_uint4(0u, 0u, 0u, 0u);
        if (q.res && c < 16 * CPR2 && u * 16 + row < nrows && mm < p.M && n < q.N2)
          resv[u][j] = *(const uint4*)(q.res + (long long)mm * q.res_ld + n);
      }
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      f32x4 acc[NT];
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int b = 0; b < NT; ++b)
          acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u][ks], bfr[ks][b], acc[b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[b][r] * sc[b] + sh[b];
          if (relu) v = fmaxf(v, 0.f);
          tb[(u * 16 + lq * 4 + r) * (16 * NT) + b * 16 + lr] = f32_to_bf16(v);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's ds_writes have landed
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      // conv c on the tile: A2[row lr][k = 8*lq + j] = the b tile's row, 16 bytes per lane (columns >= Ncols are zero:
      // zero weight columns, unit scale, zero shift in stage 1)
      uint4 a2w = make_uint4(0u, 0u, 0u, 0u);
      if (lq * 8 < 16 * NT) a2w = *(const uint4*)(tb + (u * 16 + lr) * (16 * NT) + lq * 8);
      const bf16x8 a2 = __builtin_bit_cast(bf16x8, a2w);
#pragma unroll
      for (int b = 0; b < NT2; ++b) {
        f32x4 c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, w2fr[b], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) tb2[(lq * 4 + r) * S2 + b * 16 + lr] = c2[r] * sc2[b] + sh2[b];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < CH2; ++j) {
        const int c = lane + j * 64;
        const int row = c / CPR2, n = (c - row * CPR2) * 8;
        const int mm = m0 + u * 16 + row;
        if (c < 16 * CPR2 && u * 16 + row < nrows && mm < p.M && n < q.N2) {
          const float4 lo = *(const float4*)(tb2 + row * S2 + n), hi = *(const float4*)(tb2 + row * S2 + n + 4);
          float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          if (q.res) {
            float g[8];
            unpack8_bf16(resv[u][j], g);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += g[e];
          }
          if (q.relu2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
          }
          *(uint4*)(q.y2 + (long long)mm * q.y2_ld + n) = pack8_bf16(f);
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();  // reads done before the next tile overwrites tb2 (and the next batch tb)
    }
  }
}

template <int KS, int NT2>
static int launch_direct_bc_ks(const ConvP& p, const BcP& q, hipStream_t st) {
  const int grid = direct_blocks(p.M);
  if (direct_tb(1, KS) == 4 && NT2 <= 2)
    hipLaunchKernelGGL((conv_direct_bc_kernel<1, KS, 1, 4, NT2>), dim3(grid), dim3(256), 0, st, p, q, VS_DIRECT_TPW);
  else
    hipLaunchKernelGGL((conv_direct_bc_kernel<1, KS, 1, 2, NT2>), dim3(grid), dim3(256), 0, st, p, q, VS_DIRECT_TPW);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// 32 inner channels, 9 taps (fast res4: K = 288 = 9 k-steps, 72 weight registers per lane): two tiles in flight,
// fewer tiles per wave than the 8 / 16-channel layers (a quarter of their rows)
static int launch_direct_bc32(const ConvP& p, const BcP& q, hipStream_t st) {
  static const int tpw = [] {
    const char* e = getenv("VS_BC32_TPW");
    const int v = e ? atoi(e) : 4;
    return v < 2 ? 2 : (v & ~1);
  }();
  const int grid = (int)((p.M + 64ll * tpw - 1) / (64ll * tpw));
  hipLaunchKernelGGL((conv_direct_bc_kernel<2, 9, 1, 2, 8>), dim3(grid), dim3(256), 0, st, p, q, tpw);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

template <int NT2>
static int launch_direct_bc(const ConvP& p, const BcP& q, hipStream_t st) {
  switch ((p.K + 31) / 32) {
    case 1: return launch_direct_bc_ks<1, NT2>(p, q, st);
    case 2: return launch_direct_bc_ks<2, NT2>(p, q, st);
    case 3: return launch_direct_bc_ks<3, NT2>(p, q, st);
    case 4: return launch_direct_bc_ks<4, NT2>(p, q, st);
    case 5: return launch_direct_bc_ks<5, NT2>(p, q, st);
    default: return launch_direct_bc_ks<6, NT2>(p, q, st);
  }
}

// ---- debug / cross-check kernel: one thread per output element, same gather ---
__global__ void conv_naive_kernel(ConvP p, int transposed) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)p.M * p.Ncols) return;
  const int n = (int)(idx % p.Ncols);
  const int m = (int)(idx / p.Ncols);
  int rw = m % p.Rw, t1 = m / p.Rw;
  int rh = t1 % p.Rh, t2 = t1 / p.Rh;
  int rt = t2 % p.Rt, cn = t2 / p.Rt;
  float acc = 0.f;
  for (int dt = 0; dt < p.kT; ++dt)
    for (int dh = 0; dh < p.kH; ++dh)
      for (int dw = 0; dw < p.kW; ++dw) {
        int ti = rt * p.mulT + p.offT + p.tmul * dt;
        int hi = rh * p.mulH + p.offH + p.tmul * dh;
        int wi = rw * p.mulW + p.offW + p.tmul * dw;
        if (transposed) {
          if ((ti & ((1 << p.shT) - 1)) | (hi & ((1 << p.shH) - 1)) | (wi & ((1 << p.shW) - 1)))
            continue;
          ti >>= p.shT;
          hi >>= p.shH;
          wi >>= p.shW;
        }
        if ((unsigned)ti >= (unsigned)p.Gt || (unsigned)hi >= (unsigned)p.Gh ||
            (unsigned)wi >= (unsigned)p.Gw)
          continue;
        const long long pos = ((long long)(cn * p.Gt + ti) * p.Gh + hi) * p.Gw + wi;
        const uint16_t* xp = p.x + pos * p.g_ld;
        const int tap = (dt * p.kH + dh) * p.kW + dw;
        const uint16_t* wp = p.w + (long long)n * p.K + tap * p.Cg;
        for (int c = 0; c < p.Cg; ++c) acc += bf16_to_f32(xp[c]) * bf16_to_f32(wp[c]);
      }
  float v = acc;
  if (p.flags & VS_CONV_AFFINE) v = v * p.scale[n] + p.shift[n];
  if (p.flags & VS_CONV_RESIDUAL) {
    const bool on = !p.res_bits || ((p.res_bits[(long long)m * (p.Ncols >> 3) + (n >> 3)] >> (n & 7)) & 1);
    if (on) v += bf16_to_f32(p.res[(long long)m * p.res_ld + n]);
  }
  if (p.flags & VS_CONV_RELU) v = fmaxf(v, 0.f);
  p.y[(long long)m * p.y_ld + n] = f32_to_bf16(v);
}

// ------------------------------ host side ------------------------------------
struct TileCfg {
  int bm, bn;
};

// Tile + staging plan, from the (tile x ring) sweep of every conv GEMM of the batch-8 train step
// (tools/autotune_conv.py, profiles/r01_autotune_ring.txt):
//  * shallow K (<= 2 k-steps): nothing to pipeline -> register staging, 64-row tiles;
//  * deep K, >= 256 128x128 tiles: 128x128 with a 2-stage ring (64 KB -> 2 blocks per CU);
//  * deep K, fewer tiles (slow s4 / s5 at batch 8): 64x128 with a 3-stage ring (72 KB, 2 blocks
//    per CU, 2 tiles in flight per block) -- these layers are latency / bytes-in-flight bound;
//  * <= 64 output channels: 64x64 (HBM-bound), ring once K is deep enough to matter.
static TileCfg pick_tile(long long M, int Ncols, int K, int* ring) {
  TileCfg c;
  const int nk = (K + 63) / 64;
  *ring = 0;
  if (Ncols > 64) {
    c.bn = 128;
    const long long t128 = ((M + 127) / 128) * ((Ncols + 127) / 128);
    if (nk <= 2) {
      c.bm = 64;
    } else if (t128 >= 256) {
      c.bm = 128;
      *ring = 2;
    } else {
      c.bm = 64;
      *ring = 3;
    }
  } else if (Ncols >= 32) {
    c.bn = 64;
    c.bm = 64;
    if (nk >= 8) *ring = 2;
  } else {
    c.bn = 16;
    c.bm = 256;
  }
  return c;
}

// conv_pair.hip: between vs_conv_pair_begin / _end a 128 x 128 ring-2 tile launch is recorded instead of issued
static bool pair_take_dgrad(const ConvP& p, int grid, size_t smem, int mode, bool bnb, hipStream_t st);
#ifndef VS_CONV_PAIR_TU  // compiled on its own (not through conv_pair.hip): nothing is ever recorded
static bool pair_take_dgrad(const ConvP&, int, size_t, int, bool, hipStream_t) { return false; }
#endif

template <int BM, int BN, int WM, int WN, int MODE, int NS, bool BNB = false, bool BNB2 = false>
static int launch_one(const ConvP& p, int grid, size_t smem, hipStream_t st) {
  if constexpr (BM == 128 && BN == 128 && WM == 2 && WN == 2 && NS == 2 && !BNB2) {
    if (p.sk_cnt == nullptr && pair_take_dgrad(p, grid, smem, MODE, BNB, st)) return VS_OK;
  }
  static bool attr_done = false;  // dynamic LDS above 64 KiB needs an explicit opt-in, once per kernel
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<BM, BN, WM, WN, MODE, true, 0, NS, BNB, BNB2>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, MODE, true, 0, NS, BNB, BNB2>), dim3(grid), dim3(256),
                     smem, st, p);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// tiles for which the BN-backward-sums dgrad is instantiated (what pick_tile produces for >= 32 columns)
static bool bnb_tile(int bm, int bn) { return (bn == 128 && (bm == 128 || bm == 64)) || (bm == 64 && bn == 64); }

template <int BM, int BN, int WM, int WN, int NS>
static int launch_mode(const ConvP& p, int mode, int grid, size_t smem, hipStream_t st) {
  if (p.flags & VS_CONV_BNBWD) {
    if constexpr ((BN == 128 && (BM == 128 || BM == 64)) || (BM == 64 && BN == 64)) {
      if (p.bny2) {  // + the shortcut unit's sums (unit-stride dgrads only: dgrad_impl checks)
        if (mode == 0) return launch_one<BM, BN, WM, WN, 0, NS, true, true>(p, grid, smem, st);
        return launch_one<BM, BN, WM, WN, 1, NS, true, true>(p, grid, smem, st);
      }
      if (mode == 0) return launch_one<BM, BN, WM, WN, 0, NS, true>(p, grid, smem, st);
      if (mode == 1) return launch_one<BM, BN, WM, WN, 1, NS, true>(p, grid, smem, st);
      return launch_one<BM, BN, WM, WN, 2, NS, true>(p, grid, smem, st);
    } else {
      vs_set_error("conv: BN-backward sums are not built for the %dx%d tile", BM, BN);
      return VS_ERR_UNSUPPORTED;
    }
  }
  if (mode == 0) return launch_one<BM, BN, WM, WN, 0, NS>(p, grid, smem, st);
  if (mode == 1) return launch_one<BM, BN, WM, WN, 1, NS>(p, grid, smem, st);
  return launch_one<BM, BN, WM, WN, 2, NS>(p, grid, smem, st);
}

static size_t conv_smem_bytes(int bm, int bn, int wm, int ns, int mode, int K) {
  const size_t stage = (size_t)(bm + bn) * 128, epi = (size_t)bm * bn * 4;
  const size_t ring = (size_t)(ns > 2 ? ns : 2) * stage;
  const size_t tab = mode ? (size_t)(((K + 63) >> 6) + (ns > 0 ? ns : 0)) * 64 : 0;
  const size_t rowpos = mode == 2 ? (size_t)bm * 4 : 0;
  return (ring > epi ? ring : epi) + (size_t)8 * wm * bn * 4 + rowpos + tab;
}

// ring: 0 = register-staged pipeline, 2..4 = LDS-DMA ring with that many stages
template <int BM, int BN, int WM, int WN>
static int launch_cfg(const ConvP& p, int mode, int ring, hipStream_t st) {
  const int grid = p.tilesM * p.tilesN * p.splitK;
  const bool fast = p.kT * p.kH * p.kW <= 31;
  if (!fast || BN < 32) ring = 0;
  while (ring >= 2 && conv_smem_bytes(BM, BN, WM, ring, mode, p.K) > 160 * 1024) --ring;
  if (ring < 2) ring = 0;
  const size_t smem = conv_smem_bytes(BM, BN, WM, ring, mode, p.K);
  const int dbg = (p.flags >> 12) & 7;
  if (dbg && BM == 128 && BN == 128 && mode == 1 && fast) {
    static bool dbg_attr = false;
    if (!dbg_attr) {
      (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<128, 128, 2, 2, 1, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<128, 128, 2, 2, 1, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<128, 128, 2, 2, 1, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<128, 128, 2, 2, 1, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      dbg_attr = true;
    }
    const size_t sm = conv_smem_bytes(128, 128, 2, 0, 1, p.K);
    if (dbg == 1) hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 2, 1, true, 1>), dim3(grid), dim3(256), sm, st, p);
    else if (dbg == 2) hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 2, 1, true, 2>), dim3(grid), dim3(256), sm, st, p);
    else if (dbg == 3) hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 2, 1, true, 3>), dim3(grid), dim3(256), sm, st, p);
    else hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 2, 1, true, 4>), dim3(grid), dim3(256), sm, st, p);
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  if (!fast) {  // > 31 taps: only the Cin-padded stems of configurations without a stem kernel
    if (mode != 1) {
      vs_set_error("conv: more than 31 taps is only supported for the forward gather");
      return VS_ERR_UNSUPPORTED;
    }
    static bool slow_attr = false;
    if (!slow_attr) {
      (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<BM, BN, WM, WN, 1, false>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      slow_attr = true;
    }
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, 1, false>), dim3(grid), dim3(256), smem,
                       st, p);
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  if constexpr (BN >= 32) {
    if (ring == 2) return launch_mode<BM, BN, WM, WN, 2>(p, mode, grid, smem, st);
    if (ring == 3) return launch_mode<BM, BN, WM, WN, 3>(p, mode, grid, smem, st);
    if constexpr ((BM + BN) * 128 * 4 <= 144 * 1024) {
      if (ring == 4) return launch_mode<BM, BN, WM, WN, 4>(p, mode, grid, smem, st);
    }
  }
  return launch_mode<BM, BN, WM, WN, 0>(p, mode, grid, smem, st);
}

// tile configs addressable through vs_conv_desc.flags bits 8..11 (value = id + 1; 0 = heuristic)
static const TileCfg kTileTable[] = {{128, 128}, {64, 128}, {128, 64}, {64, 64},
                                     {256, 32},  {256, 16}, {256, 128}, {128, 256}};
static const int kNumTileCfgs = 8;

struct ConvPlan {
  TileCfg tile;
  int S;        // split-K factor (1 = none)
  bool direct;  // register-resident small-channel kernel
  int ring;     // 0 = register-staged pipeline, 2..4 = LDS-DMA ring stages
  bool in_launch;  // S > 1: partial accumulators summed by the last block of each tile inside the launch (no second kernel)
};

// VS_CONV_SPLITK_IL: 0 = only where the descriptor asks for it (VS_CONV_SPLITK_IL in flags; default), 1 = where the plan
// below would pick it, 2 = wherever the shape is eligible.  OFF by default: measured at 8 clips per GPU
// (profiles/r04_splitk_in_launch.txt) it loses on every layer of the step -- a 128 x 128 fp32 partial is 64 KiB written
// through and read back per block, the operand traffic of four k-steps, and the layers it would fill the chip for are
// already on the halo-image / deep kernels.
static int splitk_il_mode() {
  static const int m = [] { const char* e = getenv("VS_CONV_SPLITK_IL"); return e ? atoi(e) : 0; }();
  return m;
}

// Few-tile, deep-K layers (slow s4 / s5 at batch 8: 100-400 tiles for 256 CUs) leave each CU
// with at most one block, which is bound by bytes-in-flight / latency; splitting K puts 2-4
// blocks on every CU (fp32 slabs, fixed-order sum in conv_splitk_epilogue_kernel).
static ConvPlan plan_conv(long long M, int Ncols, int K, int taps, int flags) {
  ConvPlan pl;
  pl.S = 1;
  pl.direct = false;
  pl.in_launch = false;
  const int fring = (flags >> 16) & 7;  // VS_CONV_RING: 1 = register-staged, 2..4 = ring stages
  pl.ring = fring >= 2 ? (fring > 4 ? 4 : fring) : 0;
  const int forced = (flags >> 8) & 0xf;
  if (forced >= 1 && forced <= kNumTileCfgs) {
    pl.tile = kTileTable[forced - 1];
    return pl;
  }
  // (K <= 192: s3's fast-pathway conv a, 64 -> 16 [3,1,1], 24.1 us on the 256 x 16 tile, joins the register-resident kernel)
  if (Ncols <= 32 && K <= 192 && taps <= 31 && M >= 64) {
    pl.direct = true;
    pl.tile = kTileTable[5];
    return pl;
  }
  int hring = 0;
  pl.tile = pick_tile(M, Ncols, K, &hring);
  if (fring == 0 && taps <= 31) pl.ring = hring;
  const int nk = (K + 63) / 64;
  const long long t128 = ((M + 127) / 128) * ((Ncols + 127) / 128);
  // measured on MI355X at batch 8: the slab write + re-read costs more than the extra occupancy
  // buys (s4.a 46 -> 58 us), so the split plan is opt-in (VS_CONV_SPLITK) until batches grow
  // In-launch split-K (the 128 x 128 tile on the 2-stage ring): under-tiled deep reductions get S blocks per tile, 2 per
  // CU; the tile's last arriver sums the S partial accumulators in split order and runs the ordinary epilogue.
  {
    const int il = (flags & VS_CONV_NOSPLITK_IL) ? 0 : ((flags & VS_CONV_SPLITK_IL) ? 2 : splitk_il_mode());
    static const int il_tiles = [] { const char* e = getenv("VS_CONV_SPLITK_IL_TILES"); return e ? atoi(e) : 224; }();
    static const int il_nk = [] { const char* e = getenv("VS_CONV_SPLITK_IL_NK"); return e ? atoi(e) : 32; }();
    const bool shape_ok = Ncols >= 128 && Ncols % 8 == 0 && taps <= 31 && nk >= 8 && fring == 0 && forced == 0;
    if (il && shape_ok && !(flags & VS_CONV_SPLITK) && (il == 2 || (t128 <= il_tiles && nk >= il_nk))) {
      long long S = (512 + t128 / 2) / t128;  // ~2 blocks per CU
      if (S > nk / 4) S = nk / 4;             // >= 4 k-steps per block
      if (S > 8) S = 8;
      if (S >= 2) {
        pl.tile = kTileTable[0];
        pl.ring = 2;
        pl.S = (int)S;
        pl.in_launch = true;
        return pl;
      }
    }
  }
  if ((flags & VS_CONV_SPLITK) && Ncols >= 128 && Ncols % 8 == 0 && taps <= 31 && t128 < 384 && nk >= 16) {
    pl.tile = kTileTable[0];
    pl.ring = fring >= 2 ? pl.ring : 0;
    long long S = (512 + t128 - 1) / t128;
    if (S > nk / 8) S = nk / 8;
    if (S > 4) S = 4;
    if (S < 1) S = 1;
    pl.S = (int)S;
  }
  return pl;
}

// in-launch plan: [counters: one int per tile, 4 KiB aligned][tile][S][128 x 128 fp32]; tile count with room for the
// class-interleaved numbering of strided data gradients (setup_stride_classes: at most 16 partly filled tiles more)
static size_t il_counter_bytes(long long M, int Ncols) {
  const size_t tiles = (size_t)((M + 127) / 128 + 16) * (size_t)((Ncols + 127) / 128);
  return (tiles * sizeof(int) + 4095) & ~(size_t)4095;
}
static size_t plan_ws_bytes(const ConvPlan& pl, long long M, int Ncols) {
  if (pl.S > 1 && pl.in_launch) {
    const size_t tiles = (size_t)((M + 127) / 128 + 16) * (size_t)((Ncols + 127) / 128);
    return il_counter_bytes(M, Ncols) + tiles * (size_t)pl.S * 128 * 128 * sizeof(float);
  }
  return pl.S > 1 ? il_counter_bytes(M, Ncols) + (size_t)pl.S * M * Ncols * sizeof(float) : 0;  // (same head: one buffer)
}

// Stride classes of the transposed gather (MODE 2): tiles are laid out class by class; sets
// p.ncls / p.cls_tiles / p.tilesM.
static void setup_stride_classes(ConvP& p, int bm, int mode, int flags) {
  p.ncls = p.nslots = 0;
  if (mode == 2 && p.kT * p.kH * p.kW <= 31 && !(flags & VS_CONV_NOCLASS)) {
    const int sT = 1 << p.shT, sH = 1 << p.shH, sW = 1 << p.shW;
    const int ncls = sT * sH * sW;
    if (ncls > 1 && ncls <= 16) {
      const int nb = p.M / (p.Rt * p.Rh * p.Rw);
      auto cnt = [](int R, int q, int off, int st) {
        const int r0 = ((q - off) % st + st) % st;
        return r0 < R ? (R - r0 + st - 1) / st : 0;
      };
      // accumulating in place: the tiles of classes no tap reaches would only copy the residual onto itself
      const bool inplace = (p.flags & VS_CONV_RESIDUAL) && p.res != nullptr && p.res == p.y && p.res_bits == nullptr &&
                           !(p.flags & VS_CONV_BNBWD);
      int tmax = 0, nslots = 0;
      for (int q = 0; q < ncls; ++q) {
        const int qw = q % sW, qh = (q / sW) % sH, qt = q / (sW * sH);
        const long long rows = (long long)nb * cnt(p.Rt, qt, p.offT, sT) * cnt(p.Rh, qh, p.offH, sH) *
                               cnt(p.Rw, qw, p.offW, sW);
        p.cls_tiles[q] = (int)((rows + bm - 1) / bm);
        const bool has_taps = qt < p.kT && qh < p.kH && qw < p.kW;
        if (inplace && !has_taps) continue;
        p.cls_ids[nslots++] = q;
        if (p.cls_tiles[q] > tmax) tmax = p.cls_tiles[q];
      }
      p.ncls = ncls;
      p.nslots = nslots;
      p.tilesM = tmax * nslots;
    }
  }
}

// the tile-kernel launch apply on load is built for: pointwise, the 128 x 128 tile on the 2-stage ring, no split-K
static bool aol_tile_ok(const ConvPlan& pl, int mode, int K) {
  return mode == 0 && !pl.direct && pl.S == 1 && pl.tile.bm == 128 && pl.tile.bn == 128 && pl.ring == 2 && K <= 512;
}

static int launch_conv(ConvP& p, int mode, int naive, int flags, void* ws, size_t ws_bytes,
                       hipStream_t st) {
  {
    // chunk-major reduction (ConvP::korder) where a tap spans at least VS_CONV_KORDER_MIN k-tiles (default 2; 0 = never)
    static const int kmin = [] { const char* e = getenv("VS_CONV_KORDER_MIN"); return e ? atoi(e) : 2; }();
    const int taps = p.kT * p.kH * p.kW;
    p.korder = (kmin > 0 && mode != 0 && taps > 1 && taps <= 31 && p.Cg % 64 == 0 && p.Cg / 64 >= kmin) ? 1 : 0;
  }
  p.splitK = 1;
  p.slab = nullptr;
  p.sk_cnt = nullptr;
  if (naive) {
    const long long total = (long long)p.M * p.Ncols;
    hipLaunchKernelGGL(conv_naive_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       p, mode == 2 ? 1 : 0);
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  {
    HaloGeo hg;
    if (flags & VS_CONV_BNB2) flags |= VS_CONV_NOHALO | VS_CONV_NOPW;  // two-unit sums: the tile kernel's variant only
    if (vs_halo_plan(p, mode, p.tmul < 0, flags, &hg)) {  // unit-stride [kT,1,1] / [1,kH,kW]: halo-image kernel
      p.tilesM = hg.tilesM;
      p.tilesN = hg.tilesN;
      return vs_halo_launch(p, hg, st);
    }
    PwGeo pg;
    if (vs_pw_plan(p, mode, flags, &pg)) {  // shallow-K pointwise: persistent weight-resident kernel
      p.tilesM = pg.tilesM;
      p.tilesN = pg.nsl;
      return vs_pw_launch(p, pg, st);
    }
    DeepGeo dg;
    if (vs_deep_plan(p, mode, flags, &dg)) {  // wide, deep reductions that fill the chip with 256 x 256 tiles
      p.tilesM = dg.tilesM;
      p.tilesN = dg.tilesN;
      return vs_deep_launch(p, mode, dg, st);
    }
  }
  const ConvPlan pl = plan_conv(p.M, p.Ncols, p.K, p.kT * p.kH * p.kW, flags);
  if (p.in_scale) {  // apply on load: vs_conv_aol_ok told the caller which launches exist
    if (!aol_tile_ok(pl, mode, p.K)) {
      vs_set_error("conv: apply on load is not built for this plan (ask vs_conv_aol_ok)");
      return VS_ERR_UNSUPPORTED;
    }
    p.tilesM = (p.M + 127) / 128;
    p.tilesN = (p.Ncols + 127) / 128;
    static std::once_flag attr;
    std::call_once(attr, [] {
      (void)hipFuncSetAttribute((const void*)conv_igemm_aol_kernel<128, 128, 2, 2, 2>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const size_t smem = conv_smem_bytes(128, 128, 2, 2, 0, p.K) + 2 * 512 * sizeof(float);
    hipLaunchKernelGGL((conv_igemm_aol_kernel<128, 128, 2, 2, 2>), dim3(p.tilesM * p.tilesN), dim3(256), smem, st, p);
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  if (pl.direct)
    return p.Ncols <= 16 ? launch_direct<1>(p, mode, st) : launch_direct<2>(p, mode, st);
  const TileCfg c = pl.tile;
  p.tilesM = (p.M + c.bm - 1) / c.bm;
  p.tilesN = (p.Ncols + c.bn - 1) / c.bn;
  setup_stride_classes(p, c.bm, mode, flags);
  if (pl.S > 1 && pl.in_launch) {
    if (ws == nullptr || ws_bytes < plan_ws_bytes(pl, p.M, p.Ncols)) {
      vs_set_error("conv: in-launch split-K workspace too small (%zu < %zu)", ws_bytes, plan_ws_bytes(pl, p.M, p.Ncols));
      return VS_ERR_WORKSPACE;
    }
    if ((size_t)p.tilesM * p.tilesN * sizeof(int) > il_counter_bytes(p.M, p.Ncols)) {
      vs_set_error("conv: in-launch split-K: %d x %d tiles exceed the counter block", p.tilesM, p.tilesN);
      return VS_ERR_WORKSPACE;
    }
    p.splitK = pl.S;
    p.sk_cnt = (int*)ws;
    p.slab = (float*)((char*)ws + il_counter_bytes(p.M, p.Ncols));
    return launch_cfg<128, 128, 2, 2>(p, mode, 2, st);
  }
  if (pl.S > 1) {
    if (p.res_bits) {
      vs_set_error("conv: the split-K plan has no masked-residual epilogue");
      return VS_ERR_UNSUPPORTED;
    }
    if (ws == nullptr || ws_bytes < plan_ws_bytes(pl, p.M, p.Ncols)) {
      vs_set_error("conv: split-K workspace too small (%zu < %zu)", ws_bytes,
                   plan_ws_bytes(pl, p.M, p.Ncols));
      return VS_ERR_WORKSPACE;
    }
    p.splitK = pl.S;
    p.slab = (float*)((char*)ws + il_counter_bytes(p.M, p.Ncols));  // (the head of the buffer holds the in-launch plan's counters)
    const int user_flags = p.flags;
    p.flags &= ~(VS_CONV_STATS | VS_CONV_AFFINE | VS_CONV_RESIDUAL | VS_CONV_RELU);
    const int rc = launch_cfg<128, 128, 2, 2>(p, mode, pl.ring, st);
    if (rc) return rc;
    hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3((p.M + 63) / 64), dim3(256), 0, st,
                       (const float*)p.slab, pl.S, p.y, p.scale, p.shift, p.res, p.stats, p.M, p.Ncols,
                       p.y_ld, p.res_ld, user_flags);
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  if (c.bm == 128 && c.bn == 128) return launch_cfg<128, 128, 2, 2>(p, mode, pl.ring, st);
  if (c.bm == 64 && c.bn == 128) return launch_cfg<64, 128, 1, 4>(p, mode, pl.ring, st);
  if (c.bm == 128 && c.bn == 64) return launch_cfg<128, 64, 2, 2>(p, mode, pl.ring, st);
  if (c.bm == 64 && c.bn == 64) return launch_cfg<64, 64, 2, 2>(p, mode, pl.ring, st);
  if (c.bm == 256 && c.bn == 32) return launch_cfg<256, 32, 4, 1>(p, mode, pl.ring, st);
  if (c.bm == 256 && c.bn == 128) return launch_cfg<256, 128, 4, 1>(p, mode, pl.ring, st);
  if (c.bm == 128 && c.bn == 256) return launch_cfg<128, 256, 1, 4>(p, mode, pl.ring, st);
  return launch_cfg<256, 16, 4, 1>(p, mode, pl.ring, st);
}

static int ilog2_exact(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return ((1 << s) == v) ? s : -1;
}

static int check_desc(const vs_conv_desc* d) {
  VS_CHECK_ARG(d != nullptr, "null desc");
  VS_CHECK_ARG(d->Cin % 8 == 0 && d->Cout % 8 == 0, "Cin and Cout must be multiples of 8");
  VS_CHECK_ARG(d->x_ld % 8 == 0 && d->y_ld % 8 == 0, "row pitches must be multiples of 8");
  VS_CHECK_ARG(d->x_ld >= d->Cin && d->y_ld >= d->Cout, "row pitch smaller than channels");
  VS_CHECK_ARG(d->kT <= 7 && d->kH <= 7 && d->kW <= 7, "kernel extent > 7");
  VS_CHECK_ARG(d->To == (d->Ti + 2 * d->pT - d->kT) / d->sT + 1, "To inconsistent");
  VS_CHECK_ARG(d->Ho == (d->Hi + 2 * d->pH - d->kH) / d->sH + 1, "Ho inconsistent");
  VS_CHECK_ARG(d->Wo == (d->Wi + 2 * d->pW - d->kW) / d->sW + 1, "Wo inconsistent");
  VS_CHECK_ARG((long long)d->N * d->Ti * d->Hi * d->Wi < (1ll << 31), "too many positions");
  VS_CHECK_ARG((long long)d->kT * d->kH * d->kW * (d->Cin > d->Cout ? d->Cin : d->Cout) <= 8 * 2048,
               "K too large for the LDS tap table");
  return VS_OK;
}

static int fill_dgrad_params(ConvP& p, const vs_conv_desc* d);

// ConvP of a forward launch (everything but the tensor pointers and byte extents); returns the kernel MODE.
#ifdef VS_STAMP
static unsigned long long* g_vs_stamp = nullptr;
// diagnostic build only: every later convolution launch stamps its blocks' phases into buf[block][8] (NULL: off)
extern "C" int vs_stamp_attach(void* buf) {
  g_vs_stamp = (unsigned long long*)buf;
  return VS_OK;
}
#endif

static int fill_fwd_params(ConvP& p, const vs_conv_desc* d) {
#ifdef VS_STAMP
  p.stamp = g_vs_stamp;
#endif
  p.x = p.w = nullptr;
  p.y = nullptr;
  p.scale = p.shift = nullptr;
  p.res = nullptr;
  p.stats = nullptr;
  p.bny = nullptr;
  p.bn_mean = p.bn_invstd = p.bn_gamma = p.bn_beta = nullptr;
  p.bn_bits = nullptr;
  p.bny_ld = 0;
  p.res_bits = nullptr;
  p.bny2 = nullptr; p.bn_mean2 = p.bn_invstd2 = nullptr; p.stats2 = nullptr; p.bny2_ld = 0;
  p.in_scale = p.in_shift = nullptr;
  p.M = d->N * d->To * d->Ho * d->Wo;
  p.nclips = d->N;
  p.Ncols = d->Cout;
  p.K = d->kT * d->kH * d->kW * d->Cin;
  p.Cg = d->Cin;
  p.g_ld = d->x_ld;
  p.Rt = d->To; p.Rh = d->Ho; p.Rw = d->Wo;
  p.Gt = d->Ti; p.Gh = d->Hi; p.Gw = d->Wi;
  p.kT = d->kT; p.kH = d->kH; p.kW = d->kW;
  p.mulT = d->sT; p.mulH = d->sH; p.mulW = d->sW;
  p.offT = -d->pT; p.offH = -d->pH; p.offW = -d->pW;
  p.tmul = 1;
  p.shT = p.shH = p.shW = 0;
  p.x_bytes = p.w_bytes = 0;
  p.y_ld = d->y_ld;
  p.res_ld = d->res_ld;
  p.flags = d->flags & 0x70ff;  // epilogue bits + debug ablation
  p.tilesM = p.tilesN = 0;
  p.ncls = p.nslots = 0;
  const bool pointwise = (d->kT * d->kH * d->kW == 1) && d->pT == 0 && d->pH == 0 && d->pW == 0;
  p.dense = (pointwise && d->sT == 1 && d->sH == 1 && d->sW == 1) ? 1 : 0;
  return pointwise ? 0 : 1;
}

extern "C" int vs_conv_stats_rows(const vs_conv_desc* d) {
  {
    ConvP p;
    HaloGeo hg;
    const int mode = fill_fwd_params(p, d);
    if (!(d->flags & VS_CONV_NAIVE) && vs_halo_plan(p, mode, 0, d->flags, &hg)) return hg.tilesM;
    PwGeo pg;
    p.flags |= d->flags & VS_CONV_RESIDUAL;
    if (vs_pw_plan(p, mode, d->flags, &pg)) return pg.tilesM;
    DeepGeo dg;
    if (!(d->flags & VS_CONV_NAIVE) && vs_deep_plan(p, mode, d->flags, &dg)) return dg.tilesM;
  }
  const long long M = (long long)d->N * d->To * d->Ho * d->Wo;
  const ConvPlan pl = plan_conv(M, d->Cout, d->kT * d->kH * d->kW * d->Cin, d->kT * d->kH * d->kW,
                                d->flags);
  if (pl.direct) return direct_blocks(M);
  if (pl.S > 1 && !pl.in_launch) return (int)((M + 63) / 64);
  return (int)((M + pl.tile.bm - 1) / pl.tile.bm);
}

extern "C" int vs_conv_plan(const vs_conv_desc* d, int dgrad, int* out) {
  VS_CHECK_ARG(d != nullptr && out != nullptr, "null argument");
  const int taps = d->kT * d->kH * d->kW;
  {
    ConvP p;
    HaloGeo hg;
    const int mode = dgrad ? fill_dgrad_params(p, d) : fill_fwd_params(p, d);
    if (mode >= 0 && !(d->flags & VS_CONV_NAIVE) && vs_halo_plan(p, mode, dgrad, d->flags, &hg)) {
      out[0] = 32 * hg.mrw;
      out[1] = 32 * hg.nrw;
      vs_halo_variant(hg, &out[2], &out[3]);  // weight-ring depth, unrolled taps (0 = generic)
      out[4] = 2;  // halo-image kernel
      return VS_OK;
    }
    PwGeo pg;
    if (mode >= 0 && vs_pw_plan(p, mode, d->flags, &pg)) {
      out[0] = 64;
      out[1] = pg.bn;
      out[2] = pg.nslot;  // activation ring slots
      out[3] = 1;
      out[4] = 3;  // persistent pointwise kernel
      return VS_OK;
    }
    DeepGeo dg;
    if (mode >= 0 && !(d->flags & VS_CONV_NAIVE) && vs_deep_plan(p, mode, d->flags, &dg)) {
      out[0] = 256;
      out[1] = 256;
      out[2] = 8;  // sub-buffers of the ring (7 in flight)
      out[3] = 1;
      out[4] = 4;  // deep-pipeline kernel
      return VS_OK;
    }
  }
  ConvPlan pl;
  if (dgrad) {
    const long long M = (long long)d->N * d->Ti * d->Hi * d->Wi;
    pl = plan_conv(M, d->Cin, taps * d->Cout, taps, d->flags);
  } else {
    const long long M = (long long)d->N * d->To * d->Ho * d->Wo;
    pl = plan_conv(M, d->Cout, taps * d->Cin, taps, d->flags);
  }
  out[0] = pl.tile.bm;
  out[1] = pl.tile.bn;
  out[2] = (taps <= 31 && pl.tile.bn >= 32 && !pl.direct) ? pl.ring : 0;
  out[3] = pl.S;
  out[4] = pl.direct ? 1 : (pl.S > 1 && pl.in_launch ? 5 : 0);  // 5: the tile kernel with the in-launch split-K sum
  return VS_OK;
}

extern "C" size_t vs_conv_workspace_bytes(const vs_conv_desc* d, int dgrad) {
  const int taps = d->kT * d->kH * d->kW;
  if (dgrad) {
    const long long M = (long long)d->N * d->Ti * d->Hi * d->Wi;
    return plan_ws_bytes(plan_conv(M, d->Cin, taps * d->Cout, taps, d->flags), M, d->Cin);
  }
  const long long M = (long long)d->N * d->To * d->Ho * d->Wo;
  return plan_ws_bytes(plan_conv(M, d->Cout, taps * d->Cin, taps, d->flags), M, d->Cout);
}

extern "C" int vs_conv_fwd(const void* x, const void* w, void* y, const vs_conv_desc* d,
                           const float* scale, const float* shift, const void* residual,
                           float* stats_partial, void* workspace, size_t ws_bytes, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  VS_CHECK_ARG(x && w && y, "null tensor");
  VS_CHECK_ARG(!(d->flags & VS_CONV_AFFINE) || (scale && shift), "AFFINE needs scale/shift");
  VS_CHECK_ARG(!(d->flags & VS_CONV_RESIDUAL) || residual, "RESIDUAL needs residual");
  VS_CHECK_ARG(!(d->flags & VS_CONV_STATS) || stats_partial, "STATS needs stats_partial");
  ConvP p;
  const int mode = fill_fwd_params(p, d);
  p.x = (const uint16_t*)x;
  p.w = (const uint16_t*)w;
  p.y = (uint16_t*)y;
  p.scale = scale;
  p.shift = shift;
  p.res = (const uint16_t*)residual;
  p.stats = stats_partial;
  {
    const long long xb = (long long)d->N * d->Ti * d->Hi * d->Wi * d->x_ld * 2;
    const long long wb = (long long)d->Cout * p.K * 2;
    VS_CHECK_ARG(xb < (1ll << 31) && wb < (1ll << 31), "tensor larger than 2 GiB");
    p.x_bytes = (unsigned)xb;
    p.w_bytes = (unsigned)wb;
  }
  return launch_conv(p, mode, (d->flags & VS_CONV_NAIVE) != 0, d->flags, workspace, ws_bytes,
                     (hipStream_t)stream);
}

// Apply on load (train, the b -> c edge of a bottleneck): which forward convolutions can take their input as the
// producer unit's RAW output plus that unit's BN scale / shift (vs_conv_fwd_aol), i.e. for which descriptors the plan
// lands on a kernel with the fragment transform: 1x1x1, unit stride, Cin <= 512, on the persistent pointwise kernel
// (128-column variant) or the 128 x 128 tile on the two-stage ring.
extern "C" int vs_conv_aol_ok(const vs_conv_desc* d) {
  if (check_desc(d) != VS_OK) return 0;
  if (d->flags & (VS_CONV_NAIVE | VS_CONV_AFFINE | VS_CONV_RESIDUAL | VS_CONV_RELU)) return 0;
  ConvP p;
  const int mode = fill_fwd_params(p, d);
  if (mode != 0 || !p.dense || p.K > 512 || p.K % 8) return 0;
  PwGeo pg;
  if (vs_pw_plan(p, mode, d->flags, &pg)) return vs_pw_aol_ok(pg, p) ? 1 : 0;
  // The apply-on-load launch never runs on the deep-pipeline kernel, but vs_conv_stats_rows (which sees no in_scale)
  // would size the statistic rows for it (256-row tiles) while the 128-row tile kernel writes twice as many rows:
  // a descriptor the deep plan accepts is refused here -- callers pass VS_CONV_NODEEP so that the rows query and the
  // launch agree.
  DeepGeo dg;
  if (vs_deep_plan(p, mode, d->flags, &dg)) return 0;
  const ConvPlan pl = plan_conv(p.M, p.Ncols, p.K, 1, d->flags);
  return aol_tile_ok(pl, mode, p.K) ? 1 : 0;
}

// y = conv1x1(relu(x * in_scale[c] + in_shift[c]), w) with x the producer's raw convolution output: the operand is
// what vs_bn_apply(x, in_scale, in_shift, relu) would have written, bit for bit, never stored.  Epilogue: VS_CONV_STATS
// only (the train-mode c unit).  Replaces `b_relu(b_bn(.))` -> `c(.)` of slowfast BottleneckTransform.forward.
extern "C" int vs_conv_fwd_aol(const void* x, const void* w, void* y, const vs_conv_desc* d, const float* in_scale,
                               const float* in_shift, float* stats_partial, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  VS_CHECK_ARG(x && w && y && in_scale && in_shift, "null tensor");
  VS_CHECK_ARG(vs_conv_aol_ok(d), "apply on load is not built for this convolution: ask vs_conv_aol_ok");
  VS_CHECK_ARG(!(d->flags & VS_CONV_STATS) || stats_partial, "STATS needs stats_partial");
  ConvP p;
  const int mode = fill_fwd_params(p, d);
  p.x = (const uint16_t*)x;
  p.w = (const uint16_t*)w;
  p.y = (uint16_t*)y;
  p.stats = stats_partial;
  p.in_scale = in_scale;
  p.in_shift = in_shift;
  {
    const long long xb = (long long)d->N * d->Ti * d->Hi * d->Wi * d->x_ld * 2;
    const long long wb = (long long)d->Cout * p.K * 2;
    VS_CHECK_ARG(xb < (1ll << 31) && wb < (1ll << 31), "tensor larger than 2 GiB");
    p.x_bytes = (unsigned)xb;
    p.w_bytes = (unsigned)wb;
  }
  return launch_conv(p, mode, 0, d->flags, nullptr, 0, (hipStream_t)stream);
}

// conv b -> conv c of a bottleneck in one launch (evaluation): which (descriptor of conv b, width of conv c) pairs
// the register-resident kernel takes.  conv b: not pointwise, 8 / 16 output channels with K <= 192 or 32 output channels
// with 256 < K <= 288 (3 x 3 on 32 channels), folded BN + ReLU and nothing else in its epilogue; conv c: 1x1x1 unit
// stride on conv b's output, <= 64 (<= 128) output channels.
extern "C" int vs_conv_fwd_bc_fusable(const vs_conv_desc* d, int cout_c) {
  if (check_desc(d) != VS_OK) return 0;
  const int taps = d->kT * d->kH * d->kW;
  const long long M = (long long)d->N * d->To * d->Ho * d->Wo;
  if (taps == 1 || taps > 31 || d->Cout % 8 != 0 || cout_c % 8 != 0 || cout_c < 8 || M < 64) return 0;
  if ((d->flags & (VS_CONV_RESIDUAL | VS_CONV_STATS | VS_CONV_NAIVE)) || !(d->flags & VS_CONV_AFFINE)) return 0;
  const int K = taps * d->Cin;
  if (d->Cout == 32) return (K > 256 && K <= 288 && cout_c <= 128) ? 1 : 0;  // launch_direct_bc32
  if (d->Cout > 16 || cout_c > 64) return 0;
  const ConvPlan pl = plan_conv(M, d->Cout, K, taps, d->flags & ~0xf00);
  return pl.direct ? 1 : 0;
}

extern "C" int vs_conv_fwd_bc(const void* x, const void* w_b, const vs_conv_desc* d, const float* scale_b,
                              const float* shift_b, const void* w_c, int cout_c, const float* scale_c,
                              const float* shift_c, const void* residual, int res_ld, void* y, int y_ld,
                              int relu_c, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  VS_CHECK_ARG(x && w_b && w_c && y && scale_b && shift_b && scale_c && shift_c, "null tensor");
  VS_CHECK_ARG(vs_conv_fwd_bc_fusable(d, cout_c), "not a fusable (conv b, conv c) pair: ask vs_conv_fwd_bc_fusable");
  VS_CHECK_ARG(y_ld >= cout_c && y_ld % 8 == 0 && (!residual || (res_ld >= cout_c && res_ld % 8 == 0)), "row pitch");
  ConvP p;
  const int mode = fill_fwd_params(p, d);
  VS_CHECK_ARG(mode == 1, "conv b must not be pointwise");
  p.x = (const uint16_t*)x;
  p.w = (const uint16_t*)w_b;
  p.scale = scale_b;
  p.shift = shift_b;
  {
    const long long xb = (long long)d->N * d->Ti * d->Hi * d->Wi * d->x_ld * 2;
    const long long wb = (long long)d->Cout * p.K * 2;
    VS_CHECK_ARG(xb < (1ll << 31) && wb < (1ll << 31), "tensor larger than 2 GiB");
    p.x_bytes = (unsigned)xb;
    p.w_bytes = (unsigned)wb;
  }
  BcP q;
  q.w2 = (const uint16_t*)w_c;
  q.w2_bytes = (unsigned)((long long)cout_c * d->Cout * 2);
  q.N2 = cout_c;
  q.scale2 = scale_c;
  q.shift2 = shift_c;
  q.res = (const uint16_t*)residual;
  q.res_ld = res_ld;
  q.y2 = (uint16_t*)y;
  q.y2_ld = y_ld;
  q.relu2 = relu_c;
  if (d->Cout == 32) return launch_direct_bc32(p, q, (hipStream_t)stream);
  return cout_c <= 32 ? launch_direct_bc<2>(p, q, (hipStream_t)stream) : launch_direct_bc<4>(p, q, (hipStream_t)stream);
}

// ConvP of a dgrad launch (everything but the tensor pointers); returns the kernel MODE or < 0.
static int fill_dgrad_params(ConvP& p, const vs_conv_desc* d) {
  const int shT = ilog2_exact(d->sT), shH = ilog2_exact(d->sH), shW = ilog2_exact(d->sW);
  if (shT < 0 || shH < 0 || shW < 0) return -1;
#ifdef VS_STAMP
  p.stamp = g_vs_stamp;
#endif
  p.scale = p.shift = nullptr;
  p.stats = nullptr;
  p.bny = nullptr;
  p.bn_mean = p.bn_invstd = p.bn_gamma = p.bn_beta = nullptr;
  p.bn_bits = nullptr;
  p.bny_ld = 0;
  p.res_bits = nullptr;
  p.bny2 = nullptr; p.bn_mean2 = p.bn_invstd2 = nullptr; p.stats2 = nullptr; p.bny2_ld = 0;
  p.in_scale = p.in_shift = nullptr;
  p.M = d->N * d->Ti * d->Hi * d->Wi;
  p.nclips = d->N;
  p.Ncols = d->Cin;
  p.K = d->kT * d->kH * d->kW * d->Cout;
  p.Cg = d->Cout;
  p.g_ld = d->y_ld;
  p.Rt = d->Ti; p.Rh = d->Hi; p.Rw = d->Wi;
  p.Gt = d->To; p.Gh = d->Ho; p.Gw = d->Wo;
  p.kT = d->kT; p.kH = d->kH; p.kW = d->kW;
  p.mulT = p.mulH = p.mulW = 1;
  p.offT = d->pT; p.offH = d->pH; p.offW = d->pW;
  p.tmul = -1;
  p.shT = shT; p.shH = shH; p.shW = shW;
  p.y_ld = d->x_ld;
  p.res_ld = d->res_ld;
  p.flags = d->flags & (VS_CONV_NAIVE | VS_CONV_RESIDUAL);
  p.tilesM = p.tilesN = 0;
  const bool unit_stride = d->sT == 1 && d->sH == 1 && d->sW == 1;
  const bool pointwise =
      unit_stride && (d->kT * d->kH * d->kW == 1) && d->pT == 0 && d->pH == 0 && d->pW == 0;
  p.dense = pointwise ? 1 : 0;
  return pointwise ? 0 : (unit_stride ? 1 : 2);
}

static int dgrad_impl(const void* dy, const void* wt, void* dx, const vs_conv_desc* d, const void* residual,
                      const uint8_t* residual_bits,
                      void* workspace, size_t ws_bytes, void* stream, const void* bn_y, int bn_y_ld,
                      const uint8_t* relu_bits, const float* mean, const float* invstd, const float* gamma,
                      const float* beta, float* stats_partial, const void* bn_y2 = nullptr, int bn_y2_ld = 0,
                      const float* mean2 = nullptr, const float* invstd2 = nullptr,
                      float* stats_partial2 = nullptr) {
  int rc = check_desc(d);
  if (rc) return rc;
  VS_CHECK_ARG(dy && wt && dx, "null tensor");
  VS_CHECK_ARG(!(d->flags & VS_CONV_RESIDUAL) || residual, "RESIDUAL needs residual");
  ConvP p;
  const int mode = fill_dgrad_params(p, d);
  VS_CHECK_ARG(mode >= 0, "strides must be powers of two");
  p.x = (const uint16_t*)dy;
  p.w = (const uint16_t*)wt;
  p.y = (uint16_t*)dx;
  p.res = (const uint16_t*)residual;
  VS_CHECK_ARG(!residual_bits || (residual && d->Cin % 8 == 0), "a residual mask needs a residual");
  p.res_bits = residual_bits;
  {
    const long long xb = (long long)d->N * d->To * d->Ho * d->Wo * d->y_ld * 2;
    const long long wb = (long long)d->Cin * p.K * 2;
    VS_CHECK_ARG(xb < (1ll << 31) && wb < (1ll << 31), "tensor larger than 2 GiB");
    p.x_bytes = (unsigned)xb;
    p.w_bytes = (unsigned)wb;
  }
  if (stats_partial) {
    VS_CHECK_ARG(bn_y && mean && invstd && bn_y_ld >= d->Cin && bn_y_ld % 8 == 0,
                 "BN-backward sums need the unit's saved conv output and its mean / invstd");
    const bool has_res = (d->flags & VS_CONV_RESIDUAL) != 0;
    VS_CHECK_ARG(has_res ? relu_bits != nullptr : (relu_bits == nullptr && gamma && beta),
                 "BN-backward sums: RESIDUAL pairs with the unit's bit mask, no residual with gamma / beta");
    VS_CHECK_ARG(vs_conv_dgrad_bnstats_rows(d) > 0,
                 "this dgrad cannot emit BN-backward sums (vs_conv_dgrad_bnstats_rows == 0)");
    p.flags |= VS_CONV_BNBWD;
    p.stats = stats_partial;
    p.bny = (const uint16_t*)bn_y;
    p.bny_ld = bn_y_ld;
    p.bn_mean = mean;
    p.bn_invstd = invstd;
    p.bn_gamma = gamma;
    p.bn_beta = beta;
    p.bn_bits = relu_bits;
    if (stats_partial2) {
      VS_CHECK_ARG(has_res && bn_y2 && mean2 && invstd2 && bn_y2_ld >= d->Cin && bn_y2_ld % 8 == 0,
                   "second BN unit: RESIDUAL form only, with its saved conv output, mean and invstd");
      VS_CHECK_ARG(mode != 2, "second BN unit: unit-stride data gradients only");
      p.bny2 = (const uint16_t*)bn_y2;
      p.bny2_ld = bn_y2_ld;
      p.bn_mean2 = mean2;
      p.bn_invstd2 = invstd2;
      p.stats2 = stats_partial2;
    }
  }
  return launch_conv(p, mode, (d->flags & VS_CONV_NAIVE) != 0, d->flags, workspace, ws_bytes,
                     (hipStream_t)stream);
}

extern "C" int vs_conv_dgrad_bnstats_rows(const vs_conv_desc* d) {
  if (d == nullptr || (d->flags & VS_CONV_NAIVE)) return 0;
  ConvP p;
  const int mode = fill_dgrad_params(p, d);
  if (mode < 0) return 0;
  // a strided dgrad with a residual: tiles of stride classes no tap reaches copy the residual without
  // passing through the epilogue
  if ((d->flags & VS_CONV_RESIDUAL) && mode == 2) return 0;
  {
    HaloGeo hg;
    const int pf = (d->flags & VS_CONV_BNB2) ? (d->flags | VS_CONV_NOHALO | VS_CONV_NOPW) : d->flags;
    if (vs_halo_plan(p, mode, 1, pf, &hg)) return hg.tilesM;
    PwGeo pg;
    p.flags |= VS_CONV_BNBWD;
    if (vs_pw_plan(p, mode, pf, &pg)) return pg.tilesM;
    DeepGeo dg;
    if (vs_deep_plan(p, mode, pf, &dg)) return dg.tilesM;
  }
  const ConvPlan pl = plan_conv(p.M, p.Ncols, p.K, p.kT * p.kH * p.kW, d->flags);
  if (pl.direct) {
    // The small-channel kernel can emit them too (one partial row per block), but it is bound by instruction
    // issue, not by bandwidth: with the mask / sum arithmetic in its copy-out the step is 0.1-0.2 ms SLOWER
    // than the plain launch plus the separate (bandwidth-bound) reduce pass -- 13.05 / 13.19 vs 12.83 / 13.08 ms,
    // A/B on one box.  Opt-in: VS_CONV_DIRECTBNB in desc.flags, or VS_DIRECT_BNB=1.
    static const int on = [] { const char* e = getenv("VS_DIRECT_BNB"); return e ? atoi(e) : 0; }();
    return (on || (d->flags & VS_CONV_DIRECTBNB)) ? direct_blocks(p.M) : 0;
  }
  if ((pl.S > 1 && !pl.in_launch) || !bnb_tile(pl.tile.bm, pl.tile.bn) || p.kT * p.kH * p.kW > 31) return 0;
  p.tilesM = (p.M + pl.tile.bm - 1) / pl.tile.bm;
  setup_stride_classes(p, pl.tile.bm, mode, d->flags);
  return p.tilesM;
}

extern "C" int vs_conv_dgrad(const void* dy, const void* wt, void* dx, const vs_conv_desc* d,
                             const void* residual, void* workspace, size_t ws_bytes, void* stream) {
  return dgrad_impl(dy, wt, dx, d, residual, nullptr, workspace, ws_bytes, stream, nullptr, 0, nullptr, nullptr,
                    nullptr, nullptr, nullptr, nullptr);
}

extern "C" int vs_conv_dgrad_ex(const void* dy, const void* wt, void* dx, const vs_conv_desc* d,
                                const vs_dgrad_epilogue* ep, void* workspace, size_t ws_bytes, void* stream) {
  VS_CHECK_ARG(ep != nullptr, "null epilogue description");
  VS_CHECK_ARG(!ep->stats_partial || ep->bn_y, "stats_partial needs the BN unit's operands");
  VS_CHECK_ARG(!ep->stats_partial2 || ep->stats_partial, "stats_partial2 rides on stats_partial");
  return dgrad_impl(dy, wt, dx, d, ep->residual, ep->residual_bits, workspace, ws_bytes, stream, ep->bn_y,
                    ep->bn_y_ld, ep->relu_bits, ep->mean, ep->invstd, ep->gamma, ep->beta, ep->stats_partial,
                    ep->bn_y2, ep->bn_y2_ld, ep->mean2, ep->invstd2, ep->stats_partial2);
}

extern "C" int vs_conv_dgrad_bnstats(const void* dy, const void* wt, void* dx, const vs_conv_desc* d,
                                     const void* residual, const void* bn_y, int bn_y_ld,
                                     const uint8_t* relu_bits, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, float* stats_partial,
                                     void* workspace, size_t ws_bytes, void* stream) {
  VS_CHECK_ARG(stats_partial, "null stats_partial");
  return dgrad_impl(dy, wt, dx, d, residual, nullptr, workspace, ws_bytes, stream, bn_y, bn_y_ld, relu_bits, mean,
                    invstd, gamma, beta, stats_partial);
}

// w [Cout][taps][Cin] -> wt [Cin][taps][Cout]
__global__ void weight_transpose_kernel(const uint16_t* w, uint16_t* wt, int Cout, int taps,
                                        int Cin) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)Cout * taps * Cin;
  if (idx >= total) return;
  const int co = (int)(idx % Cout);
  const long long r = idx / Cout;
  const int tap = (int)(r % taps);
  const int ci = (int)(r / taps);
  wt[idx] = w[((long long)co * taps + tap) * Cin + ci];
}

extern "C" int vs_weight_transpose(const void* w, void* wt, int Cout, int taps, int Cin,
                                   void* stream) {
  VS_CHECK_ARG(w && wt && Cout > 0 && taps > 0 && Cin > 0, "bad args");
  const long long total = (long long)Cout * taps * Cin;
  hipLaunchKernelGGL(weight_transpose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const uint16_t*)w, (uint16_t*)wt, Cout, taps, Cin);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// All dgrad weight images of a model in ONE launch.  table[i] = {element offset (same in
// src and dst arenas), Cout, taps, Cin, first flat index}; entries sorted by first index.
template <typename T>
__global__ void weight_transpose_batched_kernel(const T* src, T* dst, const long long* table, int n,
                                                long long total) {
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (table[mid * 5 + 4] <= idx) lo = mid; else hi = mid - 1;
    }
    const long long off = table[lo * 5 + 0];
    const int Cout = (int)table[lo * 5 + 1], taps = (int)table[lo * 5 + 2], Cin = (int)table[lo * 5 + 3];
    const long long e = idx - table[lo * 5 + 4];
    const int co = (int)(e % Cout);
    const long long r = e / Cout;
    const int tap = (int)(r % taps);
    const int ci = (int)(r / taps);
    dst[off + e] = src[off + ((long long)co * taps + tap) * Cin + ci];
  }
}

extern "C" int vs_weight_transpose_batched(const void* src, void* dst, const int64_t* table, int n,
                                           int64_t total, void* stream) {
  VS_CHECK_ARG(src && dst && table && n > 0 && total > 0, "bad args");
  long long grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(weight_transpose_batched_kernel<uint16_t>, dim3((unsigned)grid), dim3(256), 0,
                     (hipStream_t)stream, (const uint16_t*)src, (uint16_t*)dst,
                     (const long long*)table, n, (long long)total);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// The same for fp32 nn.Linear weights (taps = 1: [N][K] -> [K][N]), the operand of the
// weight-streaming dx = dy @ W kernel: all linears of a model in one launch.
extern "C" int vs_transpose_f32_batched(const float* src, float* dst, const int64_t* table, int n,
                                        int64_t total, void* stream) {
  VS_CHECK_ARG(src && dst && table && n > 0 && total > 0, "bad args");
  long long grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(weight_transpose_batched_kernel<float>, dim3((unsigned)grid), dim3(256), 0,
                     (hipStream_t)stream, src, dst, (const long long*)table, n, (long long)total);
  VS_CHECK_LAUNCH();
  return VS_OK;
}


// Tiled form of the two batched transposes (the per-element kernel above reads with a stride of
// taps*Cin elements per lane: 0.5 TB/s on the bf16 conv weights, 1.4 TB/s on the fp32 linears): a block
// moves one TS x TS tile of one (entry, tap) through LDS -- reads contiguous along Cin, writes contiguous
// along Cout -- and finds its entry by ONE binary search over tile_first[] (first tile of every entry).
template <typename T, int TS>
__global__ __launch_bounds__(256) void weight_transpose_tiled_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                                     const long long* __restrict__ table,
                                                                     const long long* __restrict__ tile_first,
                                                                     int n) {
  __shared__ T tile[TS][TS + (sizeof(T) == 2 ? 2 : 1)];
  const long long b = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_first[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const long long off = table[lo * 5 + 0];
  const int Cout = (int)table[lo * 5 + 1], taps = (int)table[lo * 5 + 2], Cin = (int)table[lo * 5 + 3];
  const int tco = (Cout + TS - 1) / TS, tci = (Cin + TS - 1) / TS;
  long long t = b - tile_first[lo];
  const int ic = (int)(t % tci);
  t /= tci;
  const int oc = (int)(t % tco);
  const int tap = (int)(t / tco);
  constexpr int RY = 256 / TS;  // thread rows per pass
  const int tx = threadIdx.x % TS, ty = threadIdx.x / TS;
#pragma unroll
  for (int r = 0; r < TS; r += RY) {
    const int co = oc * TS + ty + r, ci = ic * TS + tx;
    if (co < Cout && ci < Cin) tile[ty + r][tx] = __builtin_nontemporal_load(src + off + ((long long)co * taps + tap) * Cin + ci);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < TS; r += RY) {
    const int ci = ic * TS + ty + r, co = oc * TS + tx;
    // (both sides non-temporal: 1.1 GB per step of layout traffic beside the forward pass, read again only by the data
    //  gradients of the backward pass)
    if (ci < Cin && co < Cout) __builtin_nontemporal_store(tile[tx][ty + r], dst + off + ((long long)ci * taps + tap) * Cout + co);
  }
}

/* Tiled batched transpose: table as vs_weight_transpose_batched; tile_first[i] = first tile of entry i
 * with tiles(i) = taps * ceil(Cout / TS) * ceil(Cin / TS), TS = 64 for 2-byte and 32 for 4-byte elements. */
extern "C" int vs_weight_transpose_tiled(const void* src, void* dst, const int64_t* table,
                                         const int64_t* tile_first, int n, int64_t total_tiles, int elem_bytes,
                                         void* stream) {
  VS_CHECK_ARG(src && dst && table && tile_first && n > 0 && total_tiles > 0, "bad args");
  VS_CHECK_ARG(elem_bytes == 2 || elem_bytes == 4, "2- or 4-byte elements");
  if (elem_bytes == 2)
    hipLaunchKernelGGL((weight_transpose_tiled_kernel<uint16_t, 64>), dim3((unsigned)total_tiles), dim3(256), 0,
                       (hipStream_t)stream, (const uint16_t*)src, (uint16_t*)dst, (const long long*)table,
                       (const long long*)tile_first, n);
  else
    hipLaunchKernelGGL((weight_transpose_tiled_kernel<float, 32>), dim3((unsigned)total_tiles), dim3(256), 0,
                       (hipStream_t)stream, (const float*)src, (float*)dst, (const long long*)table,
                       (const long long*)tile_first, n);
  VS_CHECK_LAUNCH();
  return VS_OK;
}
