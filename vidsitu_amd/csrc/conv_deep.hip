// Deep-pipeline implicit-GEMM convolution kernel (gfx950): Conv3d forward / unit-stride data gradient of the
// MFMA-side layers of the SlowFast trunk (vidsitu_code/mdl_sf_base.py:22-33) -- >= 256 output channels, deep
// reductions -- on a 256 x 256 x 64 block tile.
//
// Why another tile kernel.  The 128 x 128 ring tile of conv_igemm.hip stalls once per 64-deep k-step on
// `s_waitcnt vmcnt(0)` + `s_barrier` with one tile of look-ahead, and at 64 FLOP per staged byte it cannot take in
// operands fast enough anyway: a CU's LDS-DMA path delivers 88 GB/s with 4 waves issuing, 120 GB/s with 8 (L2 hits;
// tools/probes/lds_dma_rate.hip), a 128 x 128 tile at the MFMA rate needs 32 KB per 512 cycles = 150 GB/s.  Measured
// in steady state (32 clips) it runs the s4 / s5 layers at 650-870 TFLOP/s.  Here:
//   * 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 (acc 8 x 4 fragments of v_mfma_f32_16x16x32_bf16):
//     128 FLOP per staged byte (68 GB/s at the MFMA rate), 0.375 fragment reads per MFMA.
//   * LDS = 2 stages x 64 KiB, each cut into four 16-KiB sub-buffers: At / Ab = the top / bottom 64 rows of every wave
//     row's 128, Bl / Br = the left / right 32 columns of every wave column's 64.  A k-tile is four phases of 16 MFMAs
//     per wave -- At x Bl, At x Br, Ab x Br, Ab x Bl -- and every phase: counted `s_waitcnt vmcnt`, ONE raw barrier,
//     the copies of ONE sub-buffer 7 phases (1.75 k-tiles) ahead of its first read, the fragment reads of the operand
//     the NEXT phase changes, then 16 MFMAs on fragments that are already in registers.  7 of the 8 sub-buffers
//     (112 KiB) are in flight or waiting at any time; each is read exactly once and refilled in the phase after.
//   * The waves of one half (wv >> 2; a SIMD hosts one wave of each half) issue a whole sub-buffer -- half 0 the
//     sub-buffers At and Br, half 1 Bl and Ab -- so every phase has one loading and one purely multiplying wave per
//     SIMD: the LDS-DMA issue (~100 cycles per 1-KiB instruction while the CU's address path is busy) of one wave runs
//     under the partner's MFMAs.  Of the four issue orders measured (all waves / halves x burst / spread between the
//     MFMAs: tools/probes/gemm_deep.hip, profiles/r04_gemm_deep_probe.txt) this one is the fastest: 1 336 TFLOP/s on a
//     4096^3 bf16 GEMM with random operands, the rate of the CDNA guide's 8-phase template.
//   * Gather: as in conv_igemm.hip (FAST): per row a base byte offset + a bitmask of valid taps, per k-tile one
//     (tap, delta) entry from an LDS table (fetched one k-tile ahead), every predicate an out-of-range buffer offset.
//   * Epilogue: two passes of 128 rows through an fp32 LDS tile (the 256 x 256 fp32 tile does not fit 160 KiB): BN
//     batch-statistic partials, affine, (masked) residual, ReLU, bf16 16-byte stores and -- BNB -- the BN-backward sums
//     of the unit the gradient belongs to; same arithmetic per element as conv_tile.h.
#include <stdlib.h>

#include <type_traits>

#include "conv_tile.h"

#define DEEP_EP 260  // fp32 epilogue row pitch (floats): lane quarters 4 rows apart land 16 banks apart

template <int MODE, bool BNB>
__global__ __launch_bounds__(512) void conv_deep_kernel(ConvP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int AF = 4;  // A fragments per half of a wave's rows
  constexpr unsigned SUB_AT = 0u, SUB_AB = 16384u, SUB_BL = 32768u, SUB_BR = 49152u, STAGE = 65536u;
  constexpr int MAIN = 128 * DEEP_EP * 4;  // 133 120 B: the epilogue's fp32 tile (>= the 2 x 64 KiB ring)
  static_assert(MAIN >= 2 * (int)STAGE, "ring fits the main region");

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  int swz;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tn = 0, tm = swz;
  if (p.tilesN == 2) {
    tn = swz & 1; tm = swz >> 1;
  } else if (p.tilesN == 4) {
    tn = swz & 3; tm = swz >> 2;
  } else if (p.tilesN == 8) {
    tn = swz & 7; tm = swz >> 3;
  } else if (p.tilesN != 1) {
    tn = swz % p.tilesN; tm = swz / p.tilesN;
  }
  const int m0 = tm * 256, n0 = tn * 256;
  const int K8 = p.K >> 3;
  const int nk = (p.K + 63) >> 6;

  int2* ktab = (int2*)(smem + MAIN);      // [nk * 8] (tap | weight unit << 5, byte delta); MODE 0: no table
  float* statbuf = (float*)(smem + MAIN);  // [2][8][256], after the main loop (the table is dead by then)

  if (MODE != 0) {
    const int C8 = p.Cg >> 3;
    const float rcpC8 = 1.0f / (float)C8, rcpkW = 1.0f / (float)p.kW, rcpkH = 1.0f / (float)p.kH;
    for (int k8 = tid; k8 < nk * 8; k8 += 512) {
      if (k8 >= K8) {  // K tail: tap 31 is never valid
        ktab[k8] = make_int2(31, 0);
        continue;
      }
      int tap, c8, dw, dh, dt, t2;
      if (p.korder) {  // chunk-major: k-tile kt = chunk kt / taps of tap kt % taps (ConvP::korder)
        const int ntaps = p.kT * p.kH * p.kW;
        int chunk;
        fast_divmod(k8 >> 3, ntaps, 1.0f / (float)ntaps, chunk, tap);
        c8 = chunk * 8 + (k8 & 7);
      } else {
        fast_divmod(k8, C8, rcpC8, tap, c8);
      }
      fast_divmod(tap, p.kW, rcpkW, t2, dw);
      fast_divmod(t2, p.kH, rcpkH, dt, dh);
      const long long dpos = (((long long)dt * p.Gh + dh) * p.Gw + dw) * p.tmul;
      ktab[k8] = make_int2(tap | ((tap * C8 + c8) << 5), (int)((dpos * p.g_ld + c8 * 8) * 2));
    }
  }

  // ---- copy side: the waves of half hf issue the 16 instructions of a sub-buffer, 4 each; instruction i of wave
  // wq covers rows sr = (i * 4 + wq) * 8 + (lane >> 3) of the sub-buffer, lane -> 16-byte unit lane & 7
  const int hf = wv >> 2, wq = wv & 3;
  const int kc = lane & 7, r8 = lane >> 3;
  const int kce = kc ^ (((wq & 1) << 2) | (r8 >> 1));  // logical unit fetched into physical unit kc ((sr >> 1) & 7 swizzle)
  unsigned roff[AF], vmask[AF], boff[4];
  {
    const bool small_rows = p.M < (1 << 24);
    const float rcpW = 1.0f / (float)p.Rw, rcpH = 1.0f / (float)p.Rh, rcpT = 1.0f / (float)p.Rt;
#pragma unroll
    for (int i = 0; i < AF; ++i) {
      const int sr = (i * 4 + wq) * 8 + r8;
      const int m = m0 + (sr >> 6) * 128 + hf * 64 + (sr & 63);  // half 0 fills At, half 1 Ab
      roff[i] = VS_OOB;
      vmask[i] = 0u;
      if (m < p.M) {
        int rw, t1, rh, t2, rt, n;
        if (MODE == 0 && p.dense) {
          rw = m; rh = 0; rt = 0; n = 0;
        } else if (small_rows) {
          fast_divmod(m, p.Rw, rcpW, t1, rw);
          fast_divmod(t1, p.Rh, rcpH, t2, rh);
          fast_divmod(t2, p.Rt, rcpT, n, rt);
        } else {
          rw = m % p.Rw; t1 = m / p.Rw;
          rh = t1 % p.Rh; t2 = t1 / p.Rh;
          rt = t2 % p.Rt; n = t2 / p.Rt;
        }
        if (MODE == 0) {
          const long long pos = p.dense ? (long long)m :
              ((long long)(n * p.Gt + rt * p.mulT) * p.Gh + rh * p.mulH) * p.Gw + rw * p.mulW;
          roff[i] = (unsigned)(pos * p.g_ld * 2);
          vmask[i] = 1u;
        } else {
          const int ct = rt * p.mulT + p.offT, ch = rh * p.mulH + p.offH, cw = rw * p.mulW + p.offW;
          const long long pos0 = (long long)n * p.Gt * p.Gh * p.Gw + ((long long)ct * p.Gh + ch) * p.Gw + cw;
          roff[i] = (unsigned)(pos0 * p.g_ld * 2);  // exact modulo 2^32 whenever the tap is valid
          auto axis_mask = [&](int c, int kk, int G) {
            unsigned mm = 0u;
            for (int dd = 0; dd < kk; ++dd) mm |= ((unsigned)(c + p.tmul * dd) < (unsigned)G ? 1u : 0u) << dd;
            return mm;
          };
          const unsigned mt = axis_mask(ct, p.kT, p.Gt), mh = axis_mask(ch, p.kH, p.Gh), mw = axis_mask(cw, p.kW, p.Gw);
          unsigned mk = 0u;
          int tap = 0;
          for (int dt = 0; dt < p.kT; ++dt)
            for (int dh = 0; dh < p.kH; ++dh) {
              const unsigned th = (mt >> dt) & (mh >> dh) & 1u;
              mk |= (th ? mw : 0u) << tap;
              tap += p.kW;
            }
          vmask[i] = mk;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sr = (i * 4 + wq) * 8 + r8;
      const int n = n0 + (sr >> 5) * 64 + (hf == 0 ? 32 : 0) + (sr & 31);  // half 0 fills Br, half 1 Bl
      boff[i] = n < p.Ncols ? (unsigned)((long long)n * p.K * 2) : VS_OOB;
    }
  }
  typedef __attribute__((address_space(3))) char* lds_ptr_t;
  const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem + (unsigned)wq * 1024u;
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
  // table entry of this thread's unit in k-tile `tile` (.x & 31 = tap, 31 = never valid; MODE 0: .y = byte offset of
  // the unit inside a row, also the weight row's)
  auto fetch_e = [&](int tile) __attribute__((always_inline)) {
    const int k8 = tile * 8 + kce;
    if (MODE == 0) return make_int2((tile < nk && k8 < K8) ? 0 : 31, k8 * 16);
    return tile < nk ? ktab[k8] : make_int2(31, 0);
  };
  // sub-buffer kinds: 0 At, 1 Bl, 2 Br, 3 Ab; kinds 0 / 2 are half 0's, 1 / 3 half 1's
  auto issue = [&](int kind, unsigned stage, const int2 e) __attribute__((always_inline)) {
    if ((kind & 1) != hf) return;
    const unsigned st = stage * STAGE;
    if (kind == 0 || kind == 3) {
#pragma unroll
      for (int i = 0; i < AF; ++i) {
        const unsigned ok = (vmask[i] >> (e.x & 31)) & 1u;
        dma16(xdesc, lds0 + st + (kind == 0 ? SUB_AT : SUB_AB) + (unsigned)i * 4096u, ok ? roff[i] + (unsigned)e.y : VS_OOB);
      }
    } else {
      const unsigned wk16 = MODE == 0 ? (unsigned)e.y : (((unsigned)e.x >> 5) << 4);
      const bool kv = (e.x & 31) != 31;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        dma16(wdesc, lds0 + st + (kind == 1 ? SUB_BL : SUB_BR) + (unsigned)i * 4096u,
              (kv && boff[i] != VS_OOB) ? boff[i] + wk16 : VS_OOB);
    }
  };
  // before the read of a sub-buffer of kind kr (issued 7 phases ago by half kr & 1): that half's 3 younger sub-buffers
  // (12 instructions per wave) may stay in flight
  auto wait_for = [&](int kr) __attribute__((always_inline)) {
    if (hf == (kr & 1)) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  };

  // ---- compute side
  const int wm = wv >> 2, wn = wv & 3;
  const int lr = lane & 15, lq = lane >> 4;
  const unsigned lpart = (unsigned)(lr * 128 + ((lq ^ (lr >> 1)) << 4));  // ks = 0; ks = 1: ^ 64
  const char* aBase[2] = {smem + wm * 8192 + lpart, smem + wm * 8192 + (lpart ^ 64u)};  // + stage + sub + a * 2048
  const char* bBase[2] = {smem + wn * 4096 + lpart, smem + wn * 4096 + (lpart ^ 64u)};  // + stage + sub + b * 2048

  f32x4 acc[2 * AF][4];
#pragma unroll
  for (int a = 0; a < 2 * AF; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 A0[AF][2], A1[AF][2], B0[2][2], B1[2][2];

  auto readA = [&](bf16x8 (&dst)[AF][2], unsigned off) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < AF; ++a)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) dst[a][ks] = *(const bf16x8*)(aBase[ks] + off + a * 2048);
  };
  auto readB = [&](bf16x8 (&dst)[2][2], unsigned off) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) dst[b][ks] = *(const bf16x8*)(bBase[ks] + off + b * 2048);
  };
  auto mma = [&](const bf16x8 (&Af)[AF][2], const bf16x8 (&Bf)[2][2], const int a0, const int b0)
      __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int a = 0; a < AF; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a0 + a][b0 + b] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[a][ks], Bf[b][ks], acc[a0 + a][b0 + b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto phase_open = [&](int kr) __attribute__((always_inline)) {
    wait_for(kr);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto phase_close = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  __syncthreads();  // table visible
  // Read order r = 0, 1, 2, ...: At(0), Bl(0), Br(0), Ab(0), At(1), ...: sub-buffer r is read in phase r - 2 and issued
  // in phase r - 9.  Prologue: r = 0 .. 6, then two phases without MFMAs.
  int2 e2;  // table entry of k-tile t + 2 while tile t is multiplied
  {
    const int2 e0 = fetch_e(0), e1 = fetch_e(1);
    e2 = fetch_e(2);
    issue(0, 0, e0); issue(1, 0, e0); issue(2, 0, e0); issue(3, 0, e0);
    issue(0, 1, e1); issue(1, 1, e1); issue(2, 1, e1);
    phase_open(0);  // phase -2: issue r = 7 (Ab 1), read At(0)
    issue(3, 1, e1);
    readA(A0, 0 + SUB_AT);
    phase_close();
    phase_open(1);  // phase -1: issue r = 8 (At 2), read Bl(0)
    issue(0, 0, e2);
    readB(B0, 0 + SUB_BL);
    phase_close();
  }

  // one k-tile; PAR = tile parity (stage, and which B register set holds Bl / Br)
  auto tile_body = [&](auto par, int t) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par)::value;
    constexpr unsigned ST = PAR * STAGE, STN = (PAR ^ 1) * STAGE;
    bf16x8(&BL)[2][2] = PAR ? B1 : B0;
    bf16x8(&BR)[2][2] = PAR ? B0 : B1;
    // P1: issue Bl(t+2), read Br(t), At x Bl
    phase_open(2);
    issue(1, PAR, e2);
    readB(BR, ST + SUB_BR);
    mma(A0, BL, 0, 0);
    phase_close();
    // P2: issue Br(t+2), read Ab(t), At x Br
    phase_open(3);
    issue(2, PAR, e2);
    readA(A1, ST + SUB_AB);
    mma(A0, BR, 0, 2);
    phase_close();
    // P3: issue Ab(t+2), read At(t+1) and the table entry of tile t+3, Ab x Br
    phase_open(0);
    issue(3, PAR, e2);
    readA(A0, STN + SUB_AT);
    const int2 e3 = fetch_e(t + 3);
    mma(A1, BR, AF, 2);
    phase_close();
    // P4: issue At(t+3), read Bl(t+1) into the set Br(t) leaves, Ab x Bl
    phase_open(1);
    issue(0, PAR ^ 1, e3);
    readB(BR, STN + SUB_BL);
    mma(A1, BL, AF, 0);
    phase_close();
    e2 = e3;
  };
  int t = 0;
  for (; t + 1 < nk; t += 2) {
    tile_body(std::integral_constant<int, 0>{}, t);
    tile_body(std::integral_constant<int, 1>{}, t + 1);
  }
  if (t < nk) tile_body(std::integral_constant<int, 0>{}, t);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // run-ahead copies (zeros nobody reads) before LDS is reused
  __syncthreads();

  // ------------------------------------ epilogue ------------------------------------
  // acc[a][b][r]: tile row wm * 128 + (a >> 2) * 64 + (a & 3) * 16 + lq * 4 + r, column wn * 64 + (b >> 1) * 32 +
  // (b & 1) * 16 + lr
  const int flags = p.flags;
  if (flags & VS_CONV_STATS) {  // (rows past M were zero-filled: they add nothing)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int a = 0; a < 2 * AF; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[a][b][r];
          s += v;
          q += v * v;
        }
      const int col = wn * 64 + (b >> 1) * 32 + (b & 1) * 16 + lr;
      statbuf[(wm * 4 + lq) * 256 + col] = s;
      statbuf[(8 + wm * 4 + lq) * 256 + col] = q;
    }
  }
  float sc[4], sh[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int n = n0 + wn * 64 + (b >> 1) * 32 + (b & 1) * 16 + lr;
    sc[b] = ((flags & VS_CONV_AFFINE) && n < p.Ncols) ? p.scale[n] : 1.f;
    sh[b] = ((flags & VS_CONV_AFFINE) && n < p.Ncols) ? p.shift[n] : 0.f;
  }
  const bool has_res = (flags & VS_CONV_RESIDUAL) != 0;
  const bool relu = (flags & VS_CONV_RELU) != 0;
  // copy-out: thread = one 8-channel column chunk x 16 row lanes
  const int c8 = tid & 31, rl = tid >> 5;
  const int n = n0 + c8 * 8;
  const bool nok = n < p.Ncols;
  const int nn = nok ? n : 0;
  const int bpr = p.Ncols >> 3;
  float mu[8], is[8], ga[8], be[8], sg[8], sx[8];
  if (BNB) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mu[e] = p.bn_mean[nn + e];
      is[e] = p.bn_invstd[nn + e];
      ga[e] = has_res ? 0.f : p.bn_gamma[nn + e];
      be[e] = has_res ? 0.f : p.bn_beta[nn + e];
      sg[e] = 0.f;
      sx[e] = 0.f;
    }
  }
  float* E = (float*)smem;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = wn * 64 + (b >> 1) * 32 + (b & 1) * 16 + lr;
#pragma unroll
      for (int aa = 0; aa < AF; ++aa)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          E[(wm * 64 + aa * 16 + lq * 4 + r) * DEEP_EP + col] = acc[h * AF + aa][b][r] * sc[b] + sh[b];
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      // the group's global operands are all requested before the first use
      int mm[4];
      uint4 rv[4], yv4[4];
      unsigned rb[4], bb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int prow = rl + 16 * (g * 4 + j);
        const int m = m0 + (prow >> 6) * 128 + h * 64 + (prow & 63);
        mm[j] = (m < p.M && nok) ? m : -1;
        const long long mc = mm[j] >= 0 ? mm[j] : 0;
        rv[j] = make_uint4(0u, 0u, 0u, 0u);
        rb[j] = 0xffu;
        if (has_res) {
          rv[j] = *(const uint4*)(p.res + mc * p.res_ld + nn);
          if (p.res_bits) rb[j] = p.res_bits[mc * bpr + (nn >> 3)];
        }
        if (BNB) {
          yv4[j] = *(const uint4*)(p.bny + mc * p.bny_ld + nn);
          bb[j] = has_res ? (unsigned)p.bn_bits[mc * bpr + (nn >> 3)] : 0u;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int prow = rl + 16 * (g * 4 + j);
        const float4 v0 = *(const float4*)(E + prow * DEEP_EP + c8 * 8);
        const float4 v1 = *(const float4*)(E + prow * DEEP_EP + c8 * 8 + 4);
        if (mm[j] >= 0) {
          float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          if (has_res) {
            float rf[8];
            unpack8_bf16(rv[j], rf);
            mask8(rf, rb[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rf[e];
          }
          if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          const uint4 o = pack8_bf16(v);
          *(uint4*)(p.y + (long long)mm[j] * p.y_ld + n) = o;
          if (BNB) {  // the sums see the gradient as stored
            float gq[8], yv[8];
            unpack8_bf16(o, gq);
            unpack8_bf16(yv4[j], yv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const bool on = has_res ? (((bb[j] >> e) & 1u) != 0u)
                                      : (((yv[e] - mu[e]) * is[e] * ga[e] + be[e]) > 0.f);
              const float gm = on ? gq[e] : 0.f;
              sg[e] += gm;
              sx[e] += gm * (yv[e] - mu[e]) * is[e];
            }
          }
        }
      }
    }
    __syncthreads();  // every thread is done with this pass's fp32 rows
  }
  if (BNB) {
    float* red = (float*)smem;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[tid * 16 + e] = sg[e];
      red[tid * 16 + 8 + e] = sx[e];
    }
    __syncthreads();
    if (tid < 256 && n0 + tid < p.Ncols) {  // fixed order over the 16 row lanes
      const int cc = tid >> 3, e = tid & 7;
      float ts = 0.f, tq = 0.f;
      for (int r = 0; r < 16; ++r) {
        ts += red[(r * 32 + cc) * 16 + e];
        tq += red[(r * 32 + cc) * 16 + 8 + e];
      }
      float* dst = p.stats + (long long)tm * 2 * p.Ncols;
      dst[n0 + tid] = ts;
      dst[p.Ncols + n0 + tid] = tq;
    }
  } else if ((flags & VS_CONV_STATS) && tid < 256 && n0 + tid < p.Ncols) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) {  // fixed order: wave rows, lane quarters
      s += statbuf[w * 256 + tid];
      q += statbuf[(8 + w) * 256 + tid];
    }
    float* dst = p.stats + (long long)tm * 2 * p.Ncols;
    dst[n0 + tid] = s;
    dst[p.Ncols + n0 + tid] = q;
  }
}

// ------------------------------ host side ------------------------------------
static int deep_mode() {
  static const int on = [] { const char* e = getenv("VS_CONV_DEEP"); return e ? atoi(e) : 1; }();
  return on;
}

static size_t deep_smem_bytes(int K) {
  const size_t tab = (size_t)((K + 63) >> 6) * 64;  // int2 per 16-byte unit of a k-tile
  const size_t stat = 2 * 8 * 256 * 4;
  return (size_t)128 * DEEP_EP * 4 + (tab > stat ? tab : stat);
}

// Does this launch run on the deep-pipeline kernel?  mode: kernel MODE (0 pointwise, 1 gather; the transposed gather
// of strided data gradients stays on the tile kernel).
bool vs_deep_plan(const ConvP& p, int mode, int flags, DeepGeo* out) {
  if (!deep_mode() || mode < 0 || mode > 1 || (flags & VS_CONV_NODEEP)) return false;
  if (((flags >> 8) & 0xf) != 0 || (flags & (VS_CONV_NAIVE | (7 << 12) | VS_CONV_SPLITK | (7 << 16)))) return false;  // forced plan / debug
  if (p.in_scale || p.bny2 || (flags & VS_CONV_BNB2)) return false;  // apply on load, two-unit sums: tile kernel only
  const int taps = p.kT * p.kH * p.kW;
  if (taps > 31 || p.Cg % 8 != 0 || p.Ncols % 8 != 0) return false;
  if (p.K > 8 * 2048) return false;  // table size
  if (p.Ncols < 192) return false;
  const int tilesM = (p.M + 255) / 256, tilesN = (p.Ncols + 255) / 256;
  const int nk = (p.K + 63) / 64;
  if (deep_mode() != 2 && !(flags & VS_CONV_FORCEDEEP)) {
    // Where it pays (per-layer A/B at 8 and 32 clips, isolated launches: profiles/r04_deep_layers.txt).  A block costs
    // ~13 us of prologue + two-pass epilogue and ~1.15 us per k-tile; with one block per CU (150 KiB of LDS) whole
    // rounds of 256 tiles count, and columns beyond Ncols are wasted work.  Useful fraction of the launch:
    //   fill x column efficiency x nk * 1.15 / (13 + nk * 1.15);
    // measured: >= 0.49 wins 1.14-1.30x (s4.a / s4.b / s5.b0.a at 32 clips, s4.b0.a at 8), 0.39 wins 1.15x or ties
    // (s4.a dgrad at 8 clips, s4.b0.a dgrad at 32), <= 0.36 ties or loses (s4 shortcut 0.95-1.01x, s4.b0.a dgrad at 8
    // clips 0.85x); fewer than 160 tiles (s5 at 8 clips, s4.a / s4.b forward at 8 clips) lose 0.4-0.7x.
    const long long tiles = (long long)tilesM * tilesN;
    const double col_eff = (double)p.Ncols / (256.0 * tilesN);
    const long long rounds = (tiles + 255) / 256;
    const double fill = (double)tiles / (256.0 * rounds);
    const double loop = nk * 1.15 / (13.0 + nk * 1.15);
    if (tiles < 160 || fill * col_eff * loop < 0.38) return false;
  }
  out->tilesM = tilesM;
  out->tilesN = tilesN;
  out->smem = (int)deep_smem_bytes(p.K);
  return true;
}

template <int MODE, bool BNB>
static int deep_launch_one(const ConvP& p, const DeepGeo& g, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_deep_kernel<MODE, BNB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_deep_kernel<MODE, BNB>), dim3(g.tilesM * g.tilesN), dim3(512), (size_t)g.smem, st, p);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

int vs_deep_launch(const ConvP& p, int mode, const DeepGeo& g, hipStream_t st) {
  const bool bnb = (p.flags & VS_CONV_BNBWD) != 0;
  if (mode == 0) return bnb ? deep_launch_one<0, true>(p, g, st) : deep_launch_one<0, false>(p, g, st);
  return bnb ? deep_launch_one<1, true>(p, g, st) : deep_launch_one<1, false>(p, g, st);
}
