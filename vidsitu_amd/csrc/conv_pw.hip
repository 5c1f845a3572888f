// Pointwise (1x1x1) convolutions with a shallow reduction (K <= 512) on bf16 MFMA (gfx950): the c / shortcut /
// a units of the bottleneck blocks reached from vidsitu_code/mdl_sf_base.py:22-33 (forward), and the data
// gradients of the unit-stride pointwise units (the same GEMM with the transposed weight image).
//
// Why a second kernel next to conv_igemm.hip: these GEMMs have 1..8 k-steps, so a block of the tiled kernel
// is all prologue and epilogue -- row decode, a first load that waits out one memory latency, 64 KB of
// weights fetched again for every 64 x 128 outputs -- and the layers sat at a fixed ~13 us plus 4.4 TB/s
// (profiles/r02_pw_ab.txt: every tile shape within 10 % of every other).  Here
//   * a block is persistent: it owns one BN-column slice of the weight, fetched ONCE into LDS
//     ([k-chunk][BN rows][128 B], the tile kernel's swizzled row image), and walks a strided list of
//     64-row tiles;
//   * the activation rows stream through a ring of 64 x 64 chunks filled by LDS-DMA
//     (`buffer_load_dwordx4 ... lds`), NSLOT - 1 chunks in flight, and the ring does not drain at a tile
//     boundary: the chunks of the next tiles are already on their way while the epilogue of this one runs;
//   * blocks that share a tile list (one per weight slice) sit on one XCD, so the activation chunk one of
//     them misses is an L2 hit for the others.
// The epilogue is conv_tile_epilogue (conv_tile.h): batch-statistic partials, folded-BN affine, masked
// residual, ReLU, and -- BNB -- the consumer unit's BN-backward sums, bitwise the tile kernel's results.
#include "common.h"

#include "conv_tile.h"

#define PW_BM 64
#define PW_SLOT (PW_BM * 128)

__device__ __forceinline__ void pw_wait_vm(int n) {
  switch (n) {  // wave-uniform
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
  }
}

// AOL: the activation operand is the producer's raw convolution output, BN + ReLU applied to every A fragment after
// its LDS read (ConvP::in_scale / in_shift; the constants of the <= 128 input channels sit behind the statistic rows).
template <int BN, bool BNB, int EDBG = 0, bool AOL = false>
__global__ __launch_bounds__(256) void conv_pw_kernel(ConvP p, PwGeo g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = PW_BM;
  constexpr int WM = (BN == 128) ? 1 : 2, WN = 4 / WM;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MR = TM / 16, NR = TN / 16;
  constexpr int BJ = BN / 32;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 15, lq = lane >> 4;
  const int kc = tid & 7, lrow = tid >> 3;
  [[maybe_unused]] VsStamp vst;  // (diagnostic build, conv_tile.h; a persistent block: 2 = its first chunk landed, 3 / 4 / 5 = the FIRST
                //  tile's loop / staging / stores, 7 = tiles of the block)
#ifdef VS_STAMP
  for (int i_ = 0; i_ < 8; ++i_) vst.t[i_] = 0ull;
  unsigned long long vs_first = ~0ull, vs_loop = ~0ull;
#endif
  VS_ST(vst, 0);

  // block -> (XCD, weight slice, tile list): the nsl blocks of one tile list are consecutive on one XCD
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int slice = idx % g.nsl, gid = (idx / g.nsl) * 8 + xcd;
  if (gid >= g.tilesM) return;
  const int ntl = (g.tilesM - gid + g.ngroups - 1) / g.ngroups;  // tiles of this block
  const int n0 = slice * BN;
  const int nk = g.nk, K8 = p.K >> 3;
  const int D = g.nslot - 1;

  const int b_bytes = nk * BN * 128;
  char* Bl = smem;
  char* ring = smem + b_bytes;
  char* epi = ring + g.nslot * PW_SLOT;
  float* statbuf = (float*)(epi + g.epi_bytes);
  float* aol_tab = (float*)((char*)statbuf + 4096);  // [2][128]: scale, shift of input channel k (zero beyond K)
  if (AOL) {
    if (tid < 128) {
      aol_tab[tid] = tid < p.K ? p.in_scale[tid] : 0.f;
      aol_tab[128 + tid] = tid < p.K ? p.in_shift[tid] : 0.f;
    }  // (published by the first chunk's barrier)
  }

  typedef __attribute__((address_space(3))) char* lds_ptr_t;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem + (unsigned)wv * 1024u;
  auto rsrc_words = [](const void* base, unsigned bytes) __attribute__((always_inline)) {
    const unsigned long a = (unsigned long)base;
    return (i32x4){(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
  };
  const i32x4 xdesc = rsrc_words(p.x, p.x_bytes), wdesc = rsrc_words(p.w, p.w_bytes);
  // issued from inline asm (see conv_igemm.hip: a builtin LDS-DMA makes hipcc wait vmcnt(0) before every
  // later ds_read); M0 is written and read inside one statement
  auto dma16 = [](const i32x4& desc, unsigned lds_addr, unsigned voff) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(desc)
                 : "memory");
  };
  // the thread that fills physical 16-byte unit kc of row r fetches logical unit kc ^ ((r >> 1) & 7)
  const int kce = kc ^ ((lrow >> 1) & 7);

  // ---- the weight slice, once ----
  for (int s = 0; s < nk; ++s) {
    const int k8 = s * 8 + kce;
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int n = n0 + lrow + 32 * j;
      const unsigned off = (k8 < K8 && n < p.Ncols) ? (unsigned)(((long long)n * p.K + k8 * 8) * 2) : VS_OOB;
      dma16(wdesc, lds0 + (unsigned)(s * BN * 128 + j * 4096), off);
    }
  }

  // ---- loader state: the chunk stream runs D chunks ahead of the compute ----
  int lt = 0, lk = 0;  // tile (index into this block's list) and k-chunk the next copy fetches
  unsigned roffL[2];
  auto tile_rows = [&](int ti) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      roffL[i] = VS_OOB;
      const int m = (gid + ti * g.ngroups) * BM + lrow + 32 * i;
      if (ti < ntl && m < p.M) {
        if (g.dense) {
          roffL[i] = (unsigned)((long long)m * p.g_ld * 2);
        } else {
          const int rw = m % p.Rw, t1 = m / p.Rw;
          const int rh = t1 % p.Rh, t2 = t1 / p.Rh;
          const int rt = t2 % p.Rt, n = t2 / p.Rt;
          const long long pos = ((long long)(n * p.Gt + rt * p.mulT) * p.Gh + rh * p.mulH) * p.Gw + rw * p.mulW;
          roffL[i] = (unsigned)(pos * p.g_ld * 2);
        }
      }
    }
  };
  tile_rows(0);
  auto issue = [&](int slot) __attribute__((always_inline)) {
    const int k8 = lk * 8 + kce;
    const unsigned dst = lds0 + (unsigned)(b_bytes + slot * PW_SLOT);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned off = (k8 < K8 && roffL[i] != VS_OOB) ? roffL[i] + (unsigned)k8 * 16u : VS_OOB;
      dma16(xdesc, dst + i * 4096, off);
    }
    if (++lk == nk) {
      lk = 0;
      ++lt;
      tile_rows(lt);
    }
  };
  for (int d = 0; d < D; ++d) issue(d);
  VS_ST(vst, 1);
#ifdef VS_STAMP
  vst.t[7] = (unsigned long long)ntl;
#endif

  f32x4 acc[MR][NR];
  int m0c = 0;  // first row of the tile `compute` works on (AOL)
  auto compute = [&](int slot, int kch) __attribute__((always_inline)) {
    const char* A = ring + slot * PW_SLOT;
    const char* B = Bl + kch * BN * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[MR], bfr[NR];
      const int ch = ks * 4 + lq;
#pragma unroll
      for (int a = 0; a < MR; ++a) {
        const int row = wm * TM + a * 16 + lr;
        af[a] = *(const bf16x8*)(A + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
        if (AOL) {  // rows past M were zero-filled and have to stay zero (the statistic partials sum every tile row)
          const bf16x8 t = aol_frag_k(af[a], aol_tab + kch * 64 + ch * 8, aol_tab + 128 + kch * 64 + ch * 8);
          af[a] = (m0c + row < p.M) ? t : __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
        }
      }
#pragma unroll
      for (int b = 0; b < NR; ++b) {
        const int row = wn * TN + b * 16 + lr;
        bfr[b] = *(const bf16x8*)(B + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int a = 0; a < MR; ++a)
#pragma unroll
        for (int b = 0; b < NR; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    }
  };

  const int nwait = (D - 1) * 2;
  int st_c = 0, st_l = D;  // slot computed / slot refilled this step
  for (int ti = 0; ti < ntl; ++ti) {
    const int tm = gid + ti * g.ngroups;
    const int m0 = tm * BM;
    m0c = m0;
#pragma unroll
    for (int a = 0; a < MR; ++a)
#pragma unroll
      for (int b = 0; b < NR; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kch = 0; kch < nk; ++kch) {
      if (!(g.dbg & 16)) pw_wait_vm(nwait);  // my part of this chunk (and, the first time, of the weight slice) landed
      if (!(g.dbg & 32)) __builtin_amdgcn_s_barrier();  // everyone's did; slot st_l (the previous chunk) is no longer read
      __builtin_amdgcn_sched_barrier(0);
#ifdef VS_STAMP
      { const unsigned long long now_ = vs_now(); vs_first = now_ < vs_first ? now_ : vs_first; }
#endif
      if (!(g.dbg & 8)) issue(st_l);
      __builtin_amdgcn_sched_barrier(0);
      if (!(g.dbg & 4)) compute(st_c, kch);
      __builtin_amdgcn_sched_barrier(0);
      st_c = (st_c + 1 == g.nslot) ? 0 : st_c + 1;
      st_l = (st_l + 1 == g.nslot) ? 0 : st_l + 1;
    }
    // The copies in flight belong to later tiles and land in ring slots; the epilogue works in its own
    // region.  Its own loads / stores join the same in-order counter behind them, so every counted wait
    // above stays conservative.
    if (g.dbg & 2) {
      if (acc[0][0][0] == 123.456f) epi[tid] = 1;  // keep the accumulators alive
      continue;
    }
#ifdef VS_STAMP
    { const unsigned long long now_ = vs_now(); vs_loop = now_ < vs_loop ? now_ : vs_loop; }
    VsStamp* const stp_ = ti == 0 ? &vst : nullptr;
#else
    VsStamp* const stp_ = nullptr;
#endif
    conv_tile_epilogue<BM, BN, WM, WN, BNB, true, false, EDBG>(p, acc, epi, statbuf, tm, n0, [&](int row) {
      return (m0 + row < p.M && !(g.dbg & 1)) ? m0 + row : -1;
    }, stp_);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // run-ahead copies (never read) before the block retires
#ifdef VS_STAMP
  vst.t[2] = vs_first;
  vst.t[3] = vs_loop;
#endif
  VS_ST(vst, 6);
  VS_ST_FLUSH(p, blockIdx.x, vst);
}

// ------------------------------ host side ------------------------------------
static int pw_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

bool vs_pw_plan(const ConvP& p, int mode, int flags, PwGeo* out) {
  static const int on = pw_env("VS_CONV_PW", 1), f_bn = pw_env("VS_PW_BN", 0), f_ns = pw_env("VS_PW_NSLOT", 0),
                   f_occ = pw_env("VS_PW_OCC", 0);
  if (!on || mode != 0 || (flags & VS_CONV_NOPW) || (flags & VS_CONV_NAIVE)) return false;
  if (((flags >> 8) & 0xf) || ((flags >> 16) & 7) || (flags & (VS_CONV_SPLITK | (7 << 12)))) return false;  // forced plans / debug
  if (p.kT * p.kH * p.kW != 1 || p.K > 512 || p.K % 8 || p.Ncols < 64 || p.Ncols % 8 || p.M < 2048) return false;
  PwGeo g;
  g.nk = (p.K + 63) / 64;
  g.bn = (p.Ncols >= 128 && g.nk <= 2) ? 128 : 64;
  if (f_bn == 64 || (f_bn == 128 && p.Ncols >= 128 && g.nk <= 4)) g.bn = f_bn;
  const bool has_res = (p.flags & VS_CONV_RESIDUAL) != 0, bnb = (p.flags & VS_CONV_BNBWD) != 0;
  // (bf16 staging tile: row pitch bn + 8 elements, see conv_tile_epilogue)
  g.epi_bytes = has_res ? PW_BM * g.bn * 4 : (bnb ? PW_BM * (g.bn + 8) * 2 + 16384 : PW_BM * (g.bn + 8) * 2);
  // + statbuf [2][4 * WM][BN] = 4 KB for both variants (WM * BN = 128) + 1 KB: apply-on-load constants [2][128]
  const int fixed = g.nk * g.bn * 128 + g.epi_bytes + 4096 + 1024;
  // Measured (profiles/r02_pw_ab.txt): a block's epilogue is ~1 us of serial VALU / LDS work per tile and only a
  // second resident block hides it -- one block per CU loses to the tile kernel at every shape, a deeper ring
  // buys nothing (3 slots = 5 = 8 at equal occupancy).  After the tile kernel's prologue lost its integer
  // divisions the launch wins only at K <= 128 with unit stride and >= 128 columns (s2 / s3 c units and the
  // first shortcut, their a units' data gradients: 33.7 vs 38.2, 39.3 vs 41.2 us); everywhere else the tile
  // kernel is as fast or faster.  VS_CONV_PW=2 / VS_CONV_FORCEPW / VS_PW_*: every eligible shape (A/B, tests).
  int nslot = 3;
  if (f_ns >= 3 && f_ns <= 8) nslot = f_ns;
  if (f_occ == 1 && f_ns == 0) nslot = 8;
  while (nslot > 3 && fixed + nslot * PW_SLOT > 160 * 1024) --nslot;
  if (fixed + nslot * PW_SLOT > 160 * 1024) return false;
  const int occ = (fixed + nslot * PW_SLOT) * 2 <= 160 * 1024 ? 2 : 1;
  const bool forced = on == 2 || (flags & VS_CONV_FORCEPW) || f_ns || f_occ || f_bn;
  // (not the dgrad + BN-backward-sums form: its epilogue reads two more tensors per tile with nothing to hide them behind --
  //  123 us per launch in the step's timeline against ~40 on the tile kernel)
  if (!forced && (occ < 2 || g.nk > 2 || !p.dense || p.Ncols < 128 || bnb)) return false;
  g.nslot = nslot;
  g.smem = fixed + nslot * PW_SLOT;
  g.nsl = (p.Ncols + g.bn - 1) / g.bn;
  g.tilesM = (p.M + PW_BM - 1) / PW_BM;
  int gx = (32 * occ) / g.nsl;
  if (gx < 1) gx = 1;
  if (8 * gx > g.tilesM) gx = (g.tilesM + 7) / 8;
  g.gx = gx;
  g.ngroups = 8 * gx;
  g.dbg = pw_env("VS_PW_DBG", 0);
  g.dense = (p.mulT == 1 && p.mulH == 1 && p.mulW == 1 && p.Gt == p.Rt && p.Gh == p.Rh && p.Gw == p.Rw) ? 1 : 0;
  *out = g;
  return true;
}

template <int BN, bool BNB>
static int pw_launch_one(const ConvP& p, const PwGeo& g, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_pw_kernel<BN, BNB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_pw_kernel<BN, BNB>), dim3(8 * g.nsl * g.gx), dim3(256), g.smem, st, p, g);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

template <int EDBG>
static int pw_launch_edbg(const ConvP& p, const PwGeo& g, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)conv_pw_kernel<128, false, EDBG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
  hipLaunchKernelGGL((conv_pw_kernel<128, false, EDBG>), dim3(8 * g.nsl * g.gx), dim3(256), g.smem, st, p, g);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

static int pw_launch_aol(const ConvP& p, const PwGeo& g, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_pw_kernel<128, false, 0, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_pw_kernel<128, false, 0, true>), dim3(8 * g.nsl * g.gx), dim3(256), g.smem, st, p, g);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// apply on load is built for the 128-column variant without the BN-backward-sums epilogue (the c units' forward)
bool vs_pw_aol_ok(const PwGeo& g, const ConvP& p) { return g.bn == 128 && !(p.flags & VS_CONV_BNBWD) && p.K <= 128; }

int vs_pw_launch(const ConvP& p, const PwGeo& g, hipStream_t st) {
  const bool bnb = (p.flags & VS_CONV_BNBWD) != 0;
  if (p.in_scale) {
    if (!vs_pw_aol_ok(g, p)) {
      vs_set_error("conv_pw: apply on load is not built for this variant");
      return VS_ERR_UNSUPPORTED;
    }
    return pw_launch_aol(p, g, st);
  }
  static const int edbg = pw_env("VS_PW_EDBG", 0);  // epilogue ablations (tools only): 1 no statistics, 2 no staging writes
  if (edbg && g.bn == 128 && !bnb) {
    if (edbg == 1) return pw_launch_edbg<1>(p, g, st);
    if (edbg == 2) return pw_launch_edbg<2>(p, g, st);
    return pw_launch_edbg<3>(p, g, st);
  }
  if (g.bn == 128) return bnb ? pw_launch_one<128, true>(p, g, st) : pw_launch_one<128, false>(p, g, st);
  return bnb ? pw_launch_one<64, true>(p, g, st) : pw_launch_one<64, false>(p, g, st);
}
