// Halo-image convolution kernel (gfx950): unit-stride Conv3d forward / data-gradient whose taps lie
// along ONE axis group -- [kT,1,1] (the bottleneck's conv a in slow s4 / s5) or [1,kH,kW] (conv b) --
// i.e. 58 % of SlowFast-R50's multiply-adds (vidsitu_code/mdl_sf_base.py:22-33).
//
// The implicit-GEMM kernel (conv_igemm.hip) stages one [BM x 64] activation tile PER TAP: a 3x3 conv moves
// every input row through L2 -> LDS nine times, and at batch 8 those layers (100-800 tiles on 256 CUs) are
// bound by exactly that per-CU fill rate.  Here a tile is a patch of positions that is
// closed under the taps -- a group of S spatial positions x all T frames for [kT,1,1], LH lines x W columns
// of one frame for [1,kH,kW] -- and its activation rows INCLUDING the halo are staged ONCE per 64-channel
// chunk as an "image" in LDS; every tap then reads its A fragments from that image at a shifted row
// (per-lane row address + a scalar tap offset), the zero padding being image rows filled by out-of-range
// buffer loads.  Only the weights are re-staged per tap.  Fill bytes per FLOP drop ~3x (224 x 64 tile of a
// 3x3 conv: 36 + 9 x 8 KB per 64-channel chunk instead of 9 x 36 KB).
//
//   image row  j = (g * D1 + i1) * D2 + i2      g: group in the tile, (i1, i2): position incl. halo
//   output row r = (g * O1 + o1) * O2 + o2
//   A fragment row of output row r at tap (d1, d2) = abase(r) + d1 * D2 + d2,  abase(r) = (g*D1 + o1)*D2 + o2
//   dgrad = the same with the taps mirrored (same-padded unit-stride convs are their own transpose shape).
//
// 512 threads: waves 0-3 compute (2 (M) x 2 (N), wave tile MRW x 16 rows by NRW x 16 columns,
// v_mfma_f32_16x16x32_bf16), waves 4-7 load (A images double-buffered, weight tiles in a 4-slot ring, both
// filled by LDS-DMA `buffer_load_dwordx4 ... lds` with the XOR swizzle on the source address, counted vmcnt);
// one raw barrier per (chunk, tap) step; epilogue shared with the implicit-GEMM kernel (conv_tile.h), run by
// the compute waves after the loaders have exited.
#include <stdlib.h>

#include "conv_tile.h"

#define HALO_RA_MAX 352  // image rows per tile (x 128 B per 64-channel chunk)
#define HALO_LA (HALO_RA_MAX / 32)

// D: weight tiles issued ahead of the one being multiplied (ring of D + 1 slots).  A [BN x 64] weight tile is
// only 8-16 KB, and what a CU's LDS-DMA stream delivers is (bytes in flight) / (latency): 3 tiles ahead
// (24 KB) starve a 3x3 conv (9 weight tiles per 45 KB image), 7 ahead (56 KB + the next image) do not.
// D <= taps is required by the vmcnt bookkeeping below.
// TAPS: 3 or 9 = the tap loop of the compute waves is unrolled and every A-fragment address (7 sub-tiles x
// taps) lives in a register -- the per-step address arithmetic (~90 vector instructions for 28 MFMAs) was what
// bound the first version of this kernel: a wave issues in order, and only 8 of an MFMA's 16 cycles are free
// for other vector instructions.  0 = any tap count, addresses computed per step.
template <int MRW, int NRW, bool BNB, int D, int TAPS>
__global__ __launch_bounds__(512) void conv_halo_kernel(ConvP p, HaloGeo q) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 32 * MRW, BN = 32 * NRW;
  constexpr int NSB = D + 1;
  constexpr int LA = HALO_LA, LB = BN / 32;  // LDS-DMA instructions per loader thread: image / weight tile
  constexpr int ABYTES = HALO_RA_MAX * 128, BBYTES = BN * 128;
  constexpr int STAGING = 2 * ABYTES + NSB * BBYTES;
  constexpr int EPI = BM * BN * 4 + 256 * 16 * 4;  // fp32 tile (+ the BN-backward row-lane sums)
  constexpr int MAIN = STAGING > EPI ? STAGING : EPI;
  constexpr int STAT = 2 * 8 * BN * 4;             // the epilogue's statistic rows [2][4 * 2][BN]
  constexpr bool STAT_IN_MAIN = MAIN >= EPI + STAT;  // behind the fp32 tile, in the (by then idle) staging ring
  static_assert((D - 2) * LB + LA <= 63 && D >= 3, "vmcnt range");

  const int tid = threadIdx.x;
  [[maybe_unused]] VsStamp vst;  // (diagnostic build: wave 0, a compute wave, stamps the block's phases; conv_tile.h)
#ifdef VS_STAMP
  for (int i_ = 0; i_ < 8; ++i_) vst.t[i_] = 0ull;
#endif
  VS_ST(vst, 0);
  int swz;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qq = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    swz = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (bid >> 3);
  }
  const int tn = swz % q.tilesN, tm = swz / q.tilesN;
  const int n0 = tn * BN;

  unsigned* aoff_tab = (unsigned*)(smem + MAIN);          // [HALO_RA_MAX] byte offset of image row j
  int* orow = (int*)(smem + MAIN + HALO_RA_MAX * 4);      // [BM] output position of tile row r, or -1
  int* abase = orow + BM;                                 // [BM] image row of tile row r at tap (0, 0)
  // [2][4 * 2][BN]; with 128-column tiles there is no room for it behind the tables (160 KiB of LDS)
  float* statbuf = STAT_IN_MAIN ? (float*)(smem + EPI) : (float*)(abase + BM);

  // ---- tables (every row decoded once per block) ----
  const int ipg = q.D1 * q.D2, opg = q.O1 * q.O2p;
  // (round 6, profiles/r06_launch_anatomy.txt: this decode was ~1.0-1.2 us of every block -- six run-time integer
  //  divisions per row; every operand is far below 2^24, so the float-reciprocal split of common.h is exact)
  const float rcp_ipg = 1.0f / (float)ipg, rcp_D2 = 1.0f / (float)q.D2, rcp_per = 1.0f / (float)q.per,
              rcp_opg = 1.0f / (float)opg, rcp_O2p = 1.0f / (float)q.O2p;
  for (int j = tid; j < HALO_RA_MAX; j += 512) {
    unsigned off = VS_OOB;
    if (j < q.RA) {
      int g, rem, i1, i2;
      fast_divmod(j, ipg, rcp_ipg, g, rem);
      fast_divmod(rem, q.D2, rcp_D2, i1, i2);
      const int gg = tm * q.G + g;
      if (gg < q.ngroups) {
        int hi, lo;
        fast_divmod(gg, q.per, rcp_per, hi, lo);
        long long pos = -1;
        if (q.kind == 0) {  // hi = clip, lo = spatial chunk
          const int t = i1 - q.p1, sp = lo * q.O2 + i2;
          if ((unsigned)t < (unsigned)q.T && sp < q.HW) pos = ((long long)hi * q.T + t) * q.HW + sp;
        } else {  // hi = frame, lo = line chunk
          const int h = lo * q.O1 + i1 - q.p1, w = i2 - q.p2;
          if ((unsigned)h < (unsigned)q.H && (unsigned)w < (unsigned)q.W) pos = ((long long)hi * q.H + h) * q.W + w;
        }
        if (pos >= 0) off = (unsigned)(pos * p.g_ld * 2);
      }
    }
    aoff_tab[j] = off;
  }
  for (int r = tid; r < BM; r += 512) {
    int m = -1, ab = 0;
    if (r < q.rows) {
      int g, rem, o1, o2;  // o2 >= O2: padding lanes of a 16-aligned line
      fast_divmod(r, opg, rcp_opg, g, rem);
      fast_divmod(rem, q.O2p, rcp_O2p, o1, o2);
      const int gg = tm * q.G + g;
      ab = (g * q.D1 + o1) * q.D2 + o2;
      if (gg < q.ngroups && o2 < q.O2) {
        int hi, lo;
        fast_divmod(gg, q.per, rcp_per, hi, lo);
        if (q.kind == 0) {
          const int sp = lo * q.O2 + o2;
          if (sp < q.HW) m = (hi * q.T + o1) * q.HW + sp;
        } else {
          const int h = lo * q.O1 + o1;
          if (h < q.H) m = (hi * q.H + h) * q.W + o2;
        }
      }
    }
    orow[r] = m;
    abase[r] = ab;
  }
  __syncthreads();
  VS_ST(vst, 1);

  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int taps = q.k1 * q.k2;
  const int chunks = p.Cg >> 6;
  const int nsteps = chunks * taps;

  // Roles (MI355X_MICROARCH.md, two waves per SIMD: split by wave number >= 4): waves 4-7 only move bytes --
  // every LDS-DMA, its address arithmetic and the counted vmcnt waits -- waves 0-3 only read fragments and issue
  // MFMAs.  One barrier per (chunk, tap) step joins them.  Tile s+1 is complete behind the barrier of step s,
  // so a compute wave reads the fragments of step s+1 at the end of step s, beside its last MFMAs.
  if (wv >= 4) {
    // ============================== loader waves ==============================
    const int lw = wv - 4;
    const int kc = lane & 7, r8 = lane >> 3;
    unsigned asrc[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int j = (i * 4 + lw) * 8 + r8;
      const unsigned o = aoff_tab[j];
      asrc[i] = o == VS_OOB ? VS_OOB : o + (unsigned)((kc ^ ((j >> 1) & 7)) << 4);
    }
    unsigned bsrc[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int n = (i * 4 + lw) * 8 + r8;
      bsrc[i] = (n0 + n < p.Ncols)
                    ? (unsigned)((long long)(n0 + n) * p.K * 2) + (unsigned)((kc ^ ((n >> 1) & 7)) << 4)
                    : VS_OOB;
    }
    typedef __attribute__((address_space(3))) char* lds_ptr_t;
    const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem + (unsigned)lw * 1024u;
    auto rsrc_words = [](const void* base, unsigned bytes) __attribute__((always_inline)) {
      const unsigned long a = (unsigned long)base;
      return (i32x4){(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
    };
    const i32x4 xdesc = rsrc_words(p.x, p.x_bytes), wdesc = rsrc_words(p.w, p.w_bytes);
    // issued from inline asm: a builtin LDS-DMA makes hipcc wait vmcnt(0) before every later LDS access
    auto dma16 = [](const i32x4& desc, unsigned lds_addr, unsigned voff) __attribute__((always_inline)) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                   :
                   : "s"(lds_addr), "v"(voff), "s"(desc)
                   : "memory");
    };
    auto dma_image = [&](int c) __attribute__((always_inline)) {
      const unsigned dst = lds0 + (unsigned)((c & 1) * ABYTES);
      const unsigned add = c < chunks ? (unsigned)(c * 128) : VS_OOB;  // past the last chunk: zeros, never read
#pragma unroll
      for (int i = 0; i < LA; ++i) dma16(xdesc, dst + i * 4096, (asrc[i] | add) >= VS_OOB ? VS_OOB : asrc[i] + add);
    };
    auto dma_weights = [&](int s) __attribute__((always_inline)) {
      const int c = s / taps, tw = s - c * taps;
      const unsigned dst = lds0 + (unsigned)(2 * ABYTES + (s % NSB) * BBYTES);
      const unsigned add = s < nsteps ? (unsigned)((tw * p.Cg + c * 64) * 2) : VS_OOB;
#pragma unroll
      for (int i = 0; i < LB; ++i) dma16(wdesc, dst + i * 4096, (bsrc[i] | add) >= VS_OOB ? VS_OOB : bsrc[i] + add);
    };
    // issue order: image of chunk 0, tiles 0 .. D-1; in step s: [image of the next chunk when the step opens a
    // chunk], then tile s + D.  vmcnt retires in issue order, so "tile s+1 landed" implies every image issued
    // before it -- in particular the image of the chunk that step s+1 belongs to (issued taps >= D steps before
    // that step, ahead of the tile of its own step).
    dma_image(0);
#pragma unroll
    for (int d = 0; d < D; ++d) dma_weights(d);
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"((D - 1) * LB) : "memory");  // image 0 and tile 0
    __builtin_amdgcn_s_barrier();
    int c = 0, tp = 0;
    for (int s = 0; s < nsteps; ++s) {
      // tile s+1 (issued in step s+1-D): younger are the D-2 tiles of steps s+2-D .. s-1 and, if one of those
      // steps opened a chunk (tp steps ago, 1 <= tp <= D-2), that step's image
      if (tp >= 1 && tp <= D - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((D - 2) * LB + LA) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"i"((D - 2) * LB) : "memory");
      __builtin_amdgcn_s_barrier();  // tile s+1 complete everywhere; tile s-1 and the older image are free
      if (tp == 0) dma_image(c + 1);
      dma_weights(s + D);
      if (++tp == taps) { tp = 0; ++c; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // run-ahead copies (never read) before LDS is reused
    __builtin_amdgcn_s_barrier();
    return;
  }

  // ============================== compute waves ==============================
  const int wm = wv >> 1, wn = wv & 1;
  const int lr = lane & 15, lq = lane >> 4;
  int arow[MRW];
#pragma unroll
  for (int a = 0; a < MRW; ++a) arow[a] = abase[(wm * MRW + a) * 16 + lr];
  unsigned baddr[NRW];
#pragma unroll
  for (int b = 0; b < NRW; ++b) {
    const int row = (wn * NRW + b) * 16 + lr;
    baddr[b] = (unsigned)(row * 128 + ((lq ^ ((row >> 1) & 7)) << 4));
  }
  f32x4 acc[MRW][NRW];
#pragma unroll
  for (int a = 0; a < MRW; ++a)
#pragma unroll
    for (int b = 0; b < NRW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // image-row offset of the current tap (d1, d2), walked in the weights' tap order; dgrad mirrors the taps
  int d1 = 0, d2 = 0;
  auto tap_offset = [&]() __attribute__((always_inline)) {
    return q.flip ? (q.k1 - 1 - d1) * q.D2 + (q.k2 - 1 - d2) : d1 * q.D2 + d2;
  };
  auto next_tap = [&]() __attribute__((always_inline)) {
    if (++d2 == q.k2) {
      d2 = 0;
      if (++d1 == q.k1) d1 = 0;
    }
  };
  // fragments of (chunk c, ring slot, tap offset) for one 32-deep half ks
  auto load_frags = [&](int c, int slot, int toff, int ks, bf16x8* af, bf16x8* bfr) __attribute__((always_inline)) {
    const char* A = smem + (c & 1) * ABYTES;
    const char* B = smem + 2 * ABYTES + slot * BBYTES;
#pragma unroll
    for (int a = 0; a < MRW; ++a) {
      const unsigned R = (unsigned)(arow[a] + toff);
      af[a] = *(const bf16x8*)(A + (((R << 7) + ((lq ^ ((R >> 1) & 7u)) << 4)) ^ (unsigned)(ks << 6)));
    }
#pragma unroll
    for (int b = 0; b < NRW; ++b) bfr[b] = *(const bf16x8*)(B + (baddr[b] ^ (unsigned)(ks << 6)));
  };
  auto mma = [&](const bf16x8* af, const bf16x8* bfr) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < MRW; ++a)
#pragma unroll
      for (int b = 0; b < NRW; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
  };

  bf16x8 fa0[MRW], fb0[NRW], fa1[MRW], fb1[NRW];
  if constexpr (TAPS > 0) {
    // every A-fragment byte offset (ks = 0) inside an image buffer, per tap -- two 16-bit offsets per register
    // (an image buffer is HALO_RA_MAX x 128 B = 44 KiB): 9 x 7 full words were 63 VGPRs of a 256-VGPR wave and the
    // 224-row tiles spilled 6-7 registers (a private segment = ~3 us more per launch, tools/probes/scratch_dispatch.hip)
    static_assert(HALO_RA_MAX * 128 <= 65536, "16-bit fragment offsets");
    constexpr int MP = (MRW + 1) / 2;
    unsigned aaddr[TAPS][MP];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) {
      const int toff = tap_offset();
      next_tap();
#pragma unroll
      for (int a2 = 0; a2 < MP; ++a2) {
        unsigned pk = 0u;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int a = a2 * 2 + h;
          if (a < MRW) {
            const unsigned R = (unsigned)(arow[a] + toff);
            pk |= ((R << 7) + ((lq ^ ((R >> 1) & 7u)) << 4)) << (16 * h);
          }
        }
        aaddr[tp][a2] = pk;
      }
    }
    auto loadA = [&](const char* A, const unsigned* aa, unsigned x, bf16x8* af) __attribute__((always_inline)) {
#pragma unroll
      for (int a = 0; a < MRW; ++a) {
        // (unpacked by an asm statement: written in C the compiler hoists all 63 unpacked offsets out of the tap loop
        //  again -- 69-72 spilled registers instead of 6-7)
        unsigned off;
        if (a & 1) asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(off) : "v"(aa[a >> 1]));
        else asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(off) : "v"(aa[a >> 1]));
        af[a] = *(const bf16x8*)(A + (off ^ x));
      }
    };
    auto loadB = [&](const char* B, unsigned x, bf16x8* bfr) __attribute__((always_inline)) {
#pragma unroll
      for (int b = 0; b < NRW; ++b) bfr[b] = *(const bf16x8*)(B + (baddr[b] ^ x));
    };
    const char* Bring = smem + 2 * ABYTES;
    __builtin_amdgcn_s_barrier();  // image 0 and tile 0 landed
    VS_ST(vst, 2);
    loadA(smem, aaddr[0], 0u, fa0);
    loadB(Bring, 0u, fb0);
    int slot = 0;
    for (int c = 0; c < chunks; c += 2) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        if (cb == 1 && c + 1 >= chunks) break;
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
          __builtin_amdgcn_s_barrier();  // the next step's weight tile (and its chunk's image) complete
          loadA(smem + cb * ABYTES, aaddr[tp], 64u, fa1);
          loadB(Bring + slot * BBYTES, 64u, fb1);
          __builtin_amdgcn_sched_barrier(0);
          mma(fa0, fb0);
          __builtin_amdgcn_sched_barrier(0);
          slot = (slot + 1 == NSB) ? 0 : slot + 1;
          // fragments of the next step (past the last step: bytes nobody uses)
          const int ntp = (tp + 1 == TAPS) ? 0 : tp + 1;
          const int ncb = (tp + 1 == TAPS) ? (cb ^ 1) : cb;
          loadA(smem + ncb * ABYTES, aaddr[ntp], 0u, fa0);
          loadB(Bring + slot * BBYTES, 0u, fb0);
          __builtin_amdgcn_sched_barrier(0);
          mma(fa1, fb1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  } else {
  __builtin_amdgcn_s_barrier();  // image 0 and tile 0 landed
  VS_ST(vst, 2);
  load_frags(0, 0, tap_offset(), 0, fa0, fb0);
  int c = 0, tp = 0, slot = 0;
  for (int s = 0; s < nsteps; ++s) {
    __builtin_amdgcn_s_barrier();  // tile s+1 (and its chunk's image) complete
    load_frags(c, slot, tap_offset(), 1, fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    mma(fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    next_tap();
    if (++tp == taps) { tp = 0; ++c; }
    slot = (slot + 1 == NSB) ? 0 : slot + 1;
    if (s + 1 < nsteps) load_frags(c, slot, tap_offset(), 0, fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    mma(fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
  }
  }
  VS_ST(vst, 3);
#ifdef VS_STAMP
  vst.t[7] = (unsigned long long)nsteps;
#endif
  // rows of the tile that map to no output position accumulated whatever their (clamped) image rows held:
  // zero them so that the batch-statistic partials of the epilogue see only real rows
#pragma unroll
  for (int a = 0; a < MRW; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (orow[(wm * MRW + a) * 16 + lq * 4 + r] < 0) {
#pragma unroll
        for (int b = 0; b < NRW; ++b) acc[a][b][r] = 0.f;
      }
  __builtin_amdgcn_s_barrier();  // the loaders' run-ahead copies have landed (they exit behind this barrier)

#ifdef VS_STAMP
  VsStamp* const stp_ = &vst;
#else
  VsStamp* const stp_ = nullptr;
#endif
  conv_tile_epilogue<BM, BN, 2, 2, BNB>(p, acc, smem, statbuf, tm, n0, [&](int row) { return orow[row]; }, stp_);
  VS_ST(vst, 6);
  VS_ST_FLUSH(p, blockIdx.x, vst);
}

// ------------------------------ host side ------------------------------------
static int halo_enabled() {
  static const int on = [] { const char* e = getenv("VS_CONV_HALO"); return e ? atoi(e) : 1; }();
  return on;
}

// Geometry of the halo kernel for this launch, or false when the implicit-GEMM kernel runs it.
// mode: kernel MODE of the launch (1 = unit-stride gather); dgrad: mirrored taps.
bool vs_halo_plan(const ConvP& p, int mode, int dgrad, int flags, HaloGeo* out) {
  if (!halo_enabled() || mode != 1 || (flags & VS_CONV_NOHALO)) return false;
  if (((flags >> 8) & 0xf) != 0 || (flags & (VS_CONV_NAIVE | (7 << 12) | (1 << 15)))) return false;  // forced tile / debug
  const int taps = p.kT * p.kH * p.kW;
  if (taps < 2 || taps > 25) return false;
  const bool temporal = p.kT > 1 && p.kH == 1 && p.kW == 1;
  const bool spatial = p.kT == 1 && (p.kH > 1 || p.kW > 1);
  if (!temporal && !spatial) return false;
  if ((p.kT & 1) == 0 || (p.kH & 1) == 0 || (p.kW & 1) == 0) return false;
  if (p.Cg % 64 != 0 || p.Ncols < 64 || p.Ncols % 8 != 0) return false;
  if (p.Rt != p.Gt || p.Rh != p.Gh || p.Rw != p.Gw) return false;  // "same" convolutions only
  // the launch parameters encode the padding as the offset of tap 0: -pad (forward), +pad (dgrad, tmul = -1)
  const int sgn = dgrad ? 1 : -1;
  if (p.offT != sgn * (p.kT / 2) || p.offH != sgn * (p.kH / 2) || p.offW != sgn * (p.kW / 2)) return false;
  if (p.M < 2048) return false;  // a handful of tiles: the other kernels' latency is the same
  const int T = p.Rt, H = p.Rh, W = p.Rw, HW = H * W;
  const long long nb = p.M / ((long long)T * HW);  // clips
  HaloGeo best;
  double best_score = 0.0;
  for (int bn = 64; bn <= 128; bn += 64) {
    if (bn == 128 && p.Ncols < 128) continue;
    const int tilesN = (p.Ncols + bn - 1) / bn;
    for (int mrw = 7; mrw >= 4; mrw -= 3) {
      const int bm = 32 * mrw;
      if (dgrad && mrw == 7 && bn == 128) continue;  // its BN-backward-sums epilogue does not fit 256 VGPRs
      for (int G = 1; G <= 8; G *= 2)
      for (int pad = 0; pad <= (temporal ? 0 : 1); ++pad) {
        HaloGeo g;
        g.G = G;
        g.flip = dgrad;
        g.T = T; g.H = H; g.W = W; g.HW = HW;
        g.mrw = mrw; g.nrw = bn / 32;
        if (temporal) {
          int S = bm / (G * T);
          if (S > HW) S = HW;
          if (S < 1) continue;
          g.kind = 0;
          g.k1 = p.kT; g.k2 = 1; g.p1 = p.kT / 2; g.p2 = 0;
          g.O1 = T; g.O2 = S; g.O2p = S; g.D1 = T + 2 * g.p1; g.D2 = S;
          g.per = (HW + S - 1) / S;
          g.ngroups = (int)(nb * g.per);
        } else {
          // whole lines, each padded to a multiple of 16 rows (`pad`): a 16-lane A fragment then reads 16
          // consecutive image rows and ds_read_b128 is conflict free; unpadded, the lanes behind a line end
          // land 16 rows after the first ones (2-way conflicts in 2 of the 4 lane groups, measured
          // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.45).  The largest LH (<= H) whose G groups fit.
          const int wp = pad ? (W + 15) / 16 * 16 : W;
          int LH = bm / (G * wp);
          if (LH > H) LH = H;
          if (LH < 1) continue;
          g.kind = 1;
          g.k1 = p.kH; g.k2 = p.kW; g.p1 = p.kH / 2; g.p2 = p.kW / 2;
          g.O1 = LH; g.O2 = W; g.O2p = wp; g.D1 = LH + 2 * g.p1; g.D2 = W + 2 * g.p2;
          g.per = (H + LH - 1) / LH;
          g.ngroups = (int)(nb * T * g.per);
        }
        g.RA = G * g.D1 * g.D2;
        g.rows = G * g.O1 * g.O2p;
        if (g.RA + (g.O2p - g.O2) > HALO_RA_MAX || g.rows > bm) continue;  // (padding lanes read past a line)
        g.tilesM = (g.ngroups + G - 1) / G;
        g.tilesN = tilesN;
        const double eff = (double)p.M / ((double)g.tilesM * bm);       // useful rows of the MFMA tiles
        const double blocks = (double)g.tilesM * tilesN;
        const double rounds = (double)(((long long)blocks + 255) / 256);
        const double fill = blocks / (256.0 * rounds);                    // one block per CU, whole rounds
        // fill bytes per tile-FLOP: image once per chunk + a weight tile per tap
        const double inten = (double)(G * g.O1 * g.O2) * bn * taps / ((double)g.RA + (double)taps * bn);  // FLOP / byte
        // lines that are not a multiple of 16 rows: the fragment reads conflict (measured SQ_LDS_BANK_CONFLICT /
        // SQ_LDS_IDX_ACTIVE 0.32 on s4.b, 0.49 on s5.b; simulated over row pitches 1024 + 0..128 B per 8 rows and
        // 6 swizzle families, no layout of 128-byte rows gets below 1.8x the conflict-free cycles: the hardware
        // serves a ds_read_b128 in lane groups that MIX two k-units, and an odd tap shift or a line wrap puts two of
        // a group's rows on one slot -- tools/probes/halo_lds_sim.py).  VS_HALO_CONFLICT_WEIGHT: A/B of the planner's
        // price for it (lower = prefer the padded lines).
        static const double cw = [] { const char* e = getenv("VS_HALO_CONFLICT_WEIGHT"); return e ? atof(e) : 0.85; }();
        const double conflicts = (!temporal && !pad && (W % 16) != 0) ? cw : 1.0;
        const double score = eff * fill * conflicts * (inten > 96.0 ? 1.0 : 0.6 + 0.4 * inten / 96.0) * (1.0 + 1e-4 * inten);
        if (score > best_score) { best_score = score; best = g; }
      }
    }
  }
  if (best_score <= 0.0) return false;
  // Where it pays (per-layer A/B at the bench shapes, isolated launches, profiles/r02_layer_times*.txt): the
  // few-row layers of slow s4 / s5 -- 3x3 in both directions (s4.b 26.5 -> 23.7 us, s5.b 34.5 -> 25.3 us),
  // [3,1,1] forward only and only while the image is shared by few column tiles (s4.a 34.7 -> 28.2 us, s5.a
  // 44.2 -> 38.4 us).  Layers with >= 50 k rows already fill the chip with 128 x 128 tiles and lose 5-40 % here
  // (s3.b, s2.b, the first blocks' conv a).  VS_CONV_HALO=2 / VS_CONV_FORCEHALO: every eligible launch.
  if (halo_enabled() != 2 && !(flags & VS_CONV_FORCEHALO)) {
    if (p.M > 16384) return false;
    if (temporal && (dgrad || (p.Ncols > 256 && p.M > 4096))) return false;
  }
  *out = best;
  return true;
}

static size_t halo_smem_bytes(int mrw, int nrw, int depth) {
  const size_t bm = 32 * mrw, bn = 32 * nrw;
  const size_t staging = 2 * (size_t)HALO_RA_MAX * 128 + (size_t)(depth + 1) * bn * 128;
  const size_t epi = bm * bn * 4 + 256 * 16 * 4;
  const size_t stat = 2 * 8 * bn * 4;  // statbuf [2][4 * WM][BN], WM = 2: inside the main region when the ring leaves room
  const size_t main_b = staging > epi ? staging : epi;
  return main_b + HALO_RA_MAX * 4 + 2 * bm * 4 + (main_b >= epi + stat ? 0 : stat);
}

template <int MRW, int NRW, bool BNB, int D, int TAPS>
static int halo_launch_depth(const ConvP& p, const HaloGeo& g, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_halo_kernel<MRW, NRW, BNB, D, TAPS>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_halo_kernel<MRW, NRW, BNB, D, TAPS>), dim3(g.tilesM * g.tilesN), dim3(512),
                     halo_smem_bytes(MRW, NRW, D), st, p, g);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// Kernel variant of a geometry: weight-ring depth D and the unrolled tap count (0 = generic tap loop).
// 9 taps: unrolled, deep weight ring where it fits LDS (64-column tiles); 3 taps: unrolled; else generic.
// The 224 x 128 tile has no registers left for the per-tap address table (it spills): generic path.
void vs_halo_variant(const HaloGeo& g, int* depth, int* taps_unrolled) {
  const int taps = g.k1 * g.k2;
  *depth = 3;
  *taps_unrolled = 0;
  if (g.mrw == 7 && g.nrw == 4) return;
  if (taps == 9) {
    *taps_unrolled = 9;
    if (g.nrw == 2) *depth = 7;
  } else if (taps == 3) {
    *taps_unrolled = 3;
  }
}

template <int MRW, int NRW, bool BNB>
static int halo_launch_one(const ConvP& p, const HaloGeo& g, hipStream_t st) {
  int depth, tu;
  vs_halo_variant(g, &depth, &tu);
  if constexpr (MRW == 7 && NRW == 4) return halo_launch_depth<MRW, NRW, BNB, 3, 0>(p, g, st);
  if (tu == 9) {
    if constexpr (NRW == 2) return halo_launch_depth<MRW, NRW, BNB, 7, 9>(p, g, st);
    else return halo_launch_depth<MRW, NRW, BNB, 3, 9>(p, g, st);
  }
  if (tu == 3) return halo_launch_depth<MRW, NRW, BNB, 3, 3>(p, g, st);
  return halo_launch_depth<MRW, NRW, BNB, 3, 0>(p, g, st);
}

int vs_halo_launch(const ConvP& p, const HaloGeo& g, hipStream_t st) {
  const bool bnb = (p.flags & VS_CONV_BNBWD) != 0;
#define HALO_CASE(M_, N_)                                                       \
  if (g.mrw == M_ && g.nrw == N_)                                               \
    return bnb ? halo_launch_one<M_, N_, true>(p, g, st) : halo_launch_one<M_, N_, false>(p, g, st);
  HALO_CASE(7, 2)
  HALO_CASE(7, 4)
  HALO_CASE(4, 2)
  HALO_CASE(4, 4)
#undef HALO_CASE
  vs_set_error("conv_halo: no kernel for the %d x %d wave tile", g.mrw, g.nrw);
  return VS_ERR_UNSUPPORTED;
}
