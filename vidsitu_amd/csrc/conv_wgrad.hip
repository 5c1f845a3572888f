// Conv3d weight gradient on bf16 MFMA (gfx950).
//
// Replaces autograd's cudnn_convolution_backward_weight for the trunk convs
// (vidsitu_code/mdl_sf_base.py:22-33).  GEMM view:
//   dW[co][k'] = sum_p dY[p][co] * Xg[p][k'],   k' = (dt,dh,dw,ci),
//   M = Cout, N = taps*Cin, K = output positions p (split over blocks).
// Both operands are stored position-major (channels contiguous), i.e. the
// reduction index is the strided one, so tiles are staged [64 positions][128 ch]
// in LDS exactly as loaded (coalesced 16-byte channel vectors) and MFMA
// fragments are fetched with the gfx950 transposing read ds_read_b64_tr_b16
// (4 positions x 16 channels per 16-lane group).  256-byte LDS rows with an XOR
// on the 8-byte unit index keep those reads conflict free:
//   unit' = unit ^ ((row & 3) << 2) ^ (((row >> 3) & 1) << 4).
// Split-K partials go to fp32 slabs [S][Cout][K'] and are summed in a fixed
// order by a second kernel (bitwise reproducible; no atomics).
#include <stdlib.h>

#include "common.h"

#include <mutex>
#include <type_traits>
#include "glue_bodies.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct WgradP {
  const uint16_t* dy;
  const uint16_t* x;
  float* out;  // slabs [S][Cout][Kp]
  int P;       // total output positions
  int Cout, Kp, Cin, x_ld, dy_ld;
  int To, Ho, Wo, Ti, Hi, Wi;
  int kT, kH, kW, sT, sH, sW, pT, pH, pW;
  int tilesM, tilesN, S, rows_per_split;
  unsigned x_bytes, dy_bytes;
  int xcd_order;  // 1: XCD-aware block order (default); 0: round-robin (VS_WGRAD_XCD=0, A/B)
  int dbg;        // VS_WGRAD_DBG ablations of the ring kernel (wrong results; tools only): 1 no MFMA, 2 no copies in
                  // the loop, 4 no vmcnt wait, 8 no barrier, 16 no fragment reads, 32 no table rebuild, 64 no epilogue
  // Apply on load (pointwise ring launches): x is the producer unit's RAW convolution output and the operand is
  // relu(x * in_scale[c] + in_shift[c]) rounded to bf16 (vs_bn_apply's bits), formed on the fragments.  NULL: off.
  const float* in_scale;
  const float* in_shift;
};

#define WG_OOB 0x80000000u

#define WG_ROWTAB 1024  // positions decoded per refill (16 steps of 64)

__device__ __forceinline__ int wg_swz_chunk(int row) {
  // 16-byte chunk XOR equivalent of the 8-byte unit swizzle above
  return ((row & 3) << 1) | (((row >> 3) & 1) << 3);
}

template <int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MR = TM / 16, NR = TN / 16;
  constexpr int CM = BM / 8, CN = BN / 8;            // 16-byte chunks per row
  constexpr int RPM = 256 / CM, RPN = 256 / CN;      // rows per load pass
  constexpr int IM = (64 + RPM - 1) / RPM, IN = (64 + RPN - 1) / RPN;
  constexpr int IMG = 64 * 256;                      // one image: 64 rows x 256 B
  constexpr int STAGE = 2 * IMG;
  static_assert(WM * WN == 4, "4 waves");

  const int tid = threadIdx.x;
  // XCD-aware order: workgroups b, b + 8, ... share an XCD; each XCD gets a contiguous run of the logical
  // order [split][tile], so the tiles of one position split -- which read the same x / dY rows -- sit
  // behind one L2 instead of fetching those rows once per XCD (speed only: every (split, tile) pair is
  // still computed by exactly one block).
  int bid = blockIdx.x;
  if (p.xcd_order) {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int ntile = p.tilesM * p.tilesN;
  const int s = bid / ntile;
  bid -= s * ntile;
  const int tn = bid % p.tilesN, tm = bid / p.tilesN;
  const int m0 = tm * BM, n0 = tn * BN;
  const int pbeg = s * p.rows_per_split;
  const int pend = min(p.P, pbeg + p.rows_per_split);

  int4* rowtab = (int4*)(smem + 2 * STAGE);

  // fixed per-thread column chunks
  const int ccm = tid % CM, rm = tid / CM;
  const int ccn = tid % CN, rn = tid / CN;
  const bool mcol_ok = (m0 + ccm * 8) < p.Cout;
  const int ncol = n0 + ccn * 8;
  const bool ncol_ok = ncol < p.Kp;
  int dt = 0, dh = 0, dw = 0, c0 = 0;
  if (MODE == 1 && ncol_ok) {
    const int tap = ncol / p.Cin;
    c0 = ncol - tap * p.Cin;
    dw = tap % p.kW;
    const int t2 = tap / p.kW;
    dh = t2 % p.kH;
    dt = t2 / p.kH;
  }

  u32x4 ra[IM], rb[IN];
  const __amdgpu_buffer_rsrc_t xsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dysrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dy_bytes, 0x00020000);
  // constant per thread: byte offset of its channel chunk and of its tap inside x
  const unsigned dycol = (unsigned)((m0 + ccm * 8) * 2);
  const unsigned xtap = (MODE == 1)
      ? (unsigned)(((((long long)dt * p.Hi + dh) * p.Wi + dw) * p.x_ld + c0) * 2)
      : (unsigned)(ncol * 2);

  // branch-free loads: every predicate becomes an out-of-range offset (buffer_load -> zeros)
  auto gload = [&](int pstep, int chunk0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < IM; ++i) {
      const int r = rm + RPM * i;
      const int pp = pstep + r;
      const unsigned ok = (unsigned)(r < 64) & (unsigned)mcol_ok & (unsigned)(pp < pend);
      const unsigned off = ok ? (unsigned)pp * (unsigned)(p.dy_ld * 2) + dycol : WG_OOB;
      ra[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(dysrc, off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < IN; ++i) {
      const int r = rn + RPN * i;
      const int pp = pstep + r;
      unsigned ok = (unsigned)(r < 64) & (unsigned)ncol_ok & (unsigned)(pp < pend);
      unsigned off;
      if (MODE == 0) {
        off = (unsigned)pp * (unsigned)(p.x_ld * 2) + xtap;
      } else {
        const int4 e = rowtab[(pp - chunk0) & (WG_ROWTAB - 1)];
        const int ti = e.y + dt, hi = e.z + dh, wi = e.w + dw;
        ok &= (unsigned)((unsigned)ti < (unsigned)p.Ti) & (unsigned)((unsigned)hi < (unsigned)p.Hi) &
              (unsigned)((unsigned)wi < (unsigned)p.Wi);
        off = (unsigned)e.x + xtap;
      }
      rb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, ok ? off : WG_OOB, 0, 0));
    }
  };

  auto sstore = [&](int buf) __attribute__((always_inline)) {
    char* A = smem + buf * STAGE;
    char* B = A + IMG;
#pragma unroll
    for (int i = 0; i < IM; ++i) {
      const int r = rm + RPM * i;
      if (r < 64) *(u32x4*)(A + r * 256 + ((ccm ^ wg_swz_chunk(r)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < IN; ++i) {
      const int r = rn + RPN * i;
      if (r < 64) *(u32x4*)(B + r * 256 + ((ccn ^ wg_swz_chunk(r)) << 4)) = rb[i];
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp4 = li & 3;

  f32x4 acc[MR][NR];
#pragma unroll
  for (int a = 0; a < MR; ++a)
#pragma unroll
    for (int b = 0; b < NR; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) __attribute__((always_inline)) {
    const char* A = smem + buf * STAGE;
    const char* B = A + IMG;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int row = ks * 32 + 8 * g + q;  // this lane supplies row `row` (and row+4)
      const int swz = ((row & 3) << 2) | (((row >> 3) & 1) << 4);
      bf16x8 af[MR], bfr[NR];
#pragma unroll
      for (int a = 0; a < MR; ++a) {
        const int unit = ((wm * TM + a * 16) >> 2) + pp4;
        const lds_s16x4* ptr = (const lds_s16x4*)(A + row * 256 + ((unit ^ swz) << 3));
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)ptr);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ptr + 128));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        af[a] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int b = 0; b < NR; ++b) {
        const int unit = ((wn * TN + b * 16) >> 2) + pp4;
        const lds_s16x4* ptr = (const lds_s16x4*)(B + row * 256 + ((unit ^ swz) << 3));
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)ptr);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ptr + 128));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        bfr[b] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int a = 0; a < MR; ++a)
#pragma unroll
        for (int b = 0; b < NR; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    }
  };

  // positions are processed in chunks of WG_ROWTAB rows; the pipeline restarts
  // at each chunk so the row table can be rebuilt behind a barrier.
  for (int chunk0 = pbeg; chunk0 < pend; chunk0 += WG_ROWTAB) {
    const int cend = min(pend, chunk0 + WG_ROWTAB);
    if (MODE == 1) {
      __syncthreads();  // previous chunk's readers are done
      for (int i = tid; i < WG_ROWTAB; i += 256) {
        const int pp = chunk0 + i;
        int4 e = make_int4(0, -(1 << 20), -(1 << 20), -(1 << 20));
        if (pp < cend) {
          int wo, t1, ho, t2, to, n;
          if (p.P < (1 << 24)) {
            fast_divmod(pp, p.Wo, 1.0f / (float)p.Wo, t1, wo);
            fast_divmod(t1, p.Ho, 1.0f / (float)p.Ho, t2, ho);
            fast_divmod(t2, p.To, 1.0f / (float)p.To, n, to);
          } else {
            wo = pp % p.Wo; t1 = pp / p.Wo;
            ho = t1 % p.Ho; t2 = t1 / p.Ho;
            to = t2 % p.To; n = t2 / p.To;
          }
          e.y = to * p.sT - p.pT;
          e.z = ho * p.sH - p.pH;
          e.w = wo * p.sW - p.pW;
          const long long pos0 = (((long long)n * p.Ti + e.y) * p.Hi + e.z) * p.Wi + e.w;
          e.x = (int)(unsigned)(pos0 * p.x_ld * 2);  // exact modulo 2^32 whenever the tap is valid
        }
        rowtab[i] = e;
      }
      __syncthreads();
    }
    const int nsteps = (cend - chunk0 + 63) >> 6;
    gload(chunk0, chunk0);
    __syncthreads();  // all waves finished computing on both buffers (previous chunk)
    sstore(0);
    __syncthreads();
    for (int st = 0; st < nsteps - 1; ++st) {  // straight-line body, last step peeled
      const int cur = st & 1;
      gload(chunk0 + (st + 1) * 64, chunk0);
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs
      compute(cur);
      __builtin_amdgcn_sched_barrier(0);
      sstore(cur ^ 1);
      __syncthreads();
    }
    compute((nsteps - 1) & 1);
    __syncthreads();
  }

  // D[m][n]: row = g*4 + reg (cout), col = li (k' column)
  float* dst = p.out + (long long)s * p.Cout * p.Kp;
#pragma unroll
  for (int a = 0; a < MR; ++a)
#pragma unroll
    for (int b = 0; b < NR; ++b) {
      const int col = n0 + wn * TN + b * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * TM + a * 16 + g * 4 + r;
        if (row < p.Cout && col < p.Kp) dst[(long long)row * p.Kp + col] = acc[a][b][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA ring variant (taps <= 31).  Same images, same MFMA order and therefore the same bits as
// conv_wgrad_kernel, but: (1) both operand tiles go global -> LDS by `buffer_load_dwordx4 ... lds`
// (one wave-instruction = 64 x 16 B = four 256-byte tile rows, swizzle applied to the SOURCE
// channel chunk), NS stages, NS-1 tiles in flight behind a counted vmcnt and ONE raw barrier per
// 64-position step; (2) the position table holds (byte offset, bitmask of valid taps) per row, is
// double buffered and rebuilt one chunk ahead, so the pipeline never drains inside a block
// (the register-staged kernel restarts it every 1024 positions).
// ---------------------------------------------------------------------------------------------
// ROWS (round 4): positions per ring stage.  64 = the round-1..3 ring; 32 = half stages -- the SAME 64 KiB hold four of
// them instead of two, i.e. three 32-position steps in flight behind `vmcnt(8)` instead of one 64-position step behind
// `vmcnt(0)`, at 2 blocks per CU as before (whole 256-byte rows either way: halving the K depth of THIS kernel's stage
// halves the row count, not the row length).  One barrier per 16 MFMAs per wave instead of per 32.
template <int BM, int BN, int WM, int WN, int MODE, int NS, bool AOL = false, int ROWS = 64>
__device__ __forceinline__ void conv_wgrad_ring_body(const WgradP& p, const int blk, const int nblk) {
  static_assert(!AOL || MODE == 0, "apply on load: pointwise launches");
  static_assert(ROWS == 64, "stage depth (the half-depth variant of round 4 was measured slower and removed)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MR = TM / 16, NR = TN / 16;
  constexpr int CM = BM / 8, CN = BN / 8;  // 16-byte chunks that carry data (of 16 per row)
  constexpr int IMG = ROWS * 256;
  constexpr int STAGE = 2 * IMG;
  constexpr int D = NS - 1;
  constexpr int RG = ROWS / 16;  // row groups (of 16 positions: 4 rows per wave instruction x 4 waves) per image
  constexpr int L = 2 * RG;      // DMA instructions per thread per step: RG row groups x 2 images
  constexpr int SPC = WG_ROWTAB / ROWS;  // steps per table chunk
  static_assert(WM * WN == 4, "4 waves");
  static_assert((D - 1) * L <= 63, "vmcnt range");

  const int tid = threadIdx.x;
  // XCD-aware order: workgroups b, b + 8, ... share an XCD; each XCD gets a contiguous run of the logical
  // order [split][tile], so the tiles of one position split -- which read the same x / dY rows -- sit
  // behind one L2 instead of fetching those rows once per XCD (speed only: every (split, tile) pair is
  // still computed by exactly one block).
  int bid = blk;
  if (p.xcd_order) {
    const int nwg = nblk, q = nwg >> 3, r = nwg & 7, xcd = blk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blk >> 3);
  }
  const int ntile = p.tilesM * p.tilesN;
  const int s = bid / ntile;
  bid -= s * ntile;
  const int tn = bid % p.tilesN, tm = bid / p.tilesN;
  const int m0 = tm * BM, n0 = tn * BN;
  const int pbeg = s * p.rows_per_split;
  const int pend = min(p.P, pbeg + p.rows_per_split);
  const int nsteps = (pend - pbeg + ROWS - 1) / ROWS;
  int2* rowtab = (int2*)(smem + NS * STAGE);  // [2][WG_ROWTAB] (byte offset, tap mask)

  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int rg = lane >> 4, cc = lane & 15;
  // logical 16-byte chunk this thread fetches into physical chunk cc of rows 16 i + 4 wv + rg
  const int lc = cc ^ ((rg << 1) | ((wv >> 1) << 3));
  const bool mcol_ok = lc < CM && (m0 + lc * 8) < p.Cout;
  const int ncol = n0 + lc * 8;
  const bool ncol_ok = lc < CN && ncol < p.Kp;
  int tap = 0, c0 = 0, dt = 0, dh = 0, dw = 0;
  if (MODE == 1 && ncol_ok) {
    tap = ncol / p.Cin;
    c0 = ncol - tap * p.Cin;
    dw = tap % p.kW;
    const int t2 = tap / p.kW;
    dh = t2 % p.kH;
    dt = t2 / p.kH;
  }
  const unsigned dycol = (unsigned)((m0 + lc * 8) * 2);
  const unsigned xtap = (MODE == 1)
      ? (unsigned)(((((long long)dt * p.Hi + dh) * p.Wi + dw) * p.x_ld + c0) * 2)
      : (unsigned)(ncol * 2);
  const unsigned dy_pitch = (unsigned)(p.dy_ld * 2), x_pitch = (unsigned)(p.x_ld * 2);

  typedef __attribute__((address_space(3))) char* lds_ptr_t;
  const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem + (unsigned)wv * 1024u;
  auto rsrc_words = [](const void* base, unsigned bytes) __attribute__((always_inline)) {
    const unsigned long a = (unsigned long)base;
    return (i32x4){(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
  };
  const i32x4 xdesc = rsrc_words(p.x, p.x_bytes), dydesc = rsrc_words(p.dy, p.dy_bytes);
  // issued from inline asm: see conv_igemm.hip (a builtin LDS-DMA serialises every later ds_read)
  auto dma16 = [](const i32x4& desc, unsigned lds_addr, unsigned voff) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(desc)
                 : "memory");
  };

  const bool small_p = p.P < (1 << 24);
  const float rcpWo = 1.0f / (float)p.Wo, rcpHo = 1.0f / (float)p.Ho, rcpTo = 1.0f / (float)p.To;
  auto build_tab = [&](int chunk) __attribute__((always_inline)) {
    int2* tab = rowtab + (chunk & 1) * WG_ROWTAB;
    const int base = pbeg + chunk * WG_ROWTAB;
    for (int i = tid; i < WG_ROWTAB; i += 256) {
      const int pp = base + i;
      int2 e = make_int2(0, 0);
      if (pp < pend) {
        int wo, t1, ho, t2, to, n;
        if (small_p) {  // float-reciprocal split (exact below 2^24 positions) instead of three integer divisions
          fast_divmod(pp, p.Wo, rcpWo, t1, wo);
          fast_divmod(t1, p.Ho, rcpHo, t2, ho);
          fast_divmod(t2, p.To, rcpTo, n, to);
        } else {
          wo = pp % p.Wo; t1 = pp / p.Wo;
          ho = t1 % p.Ho; t2 = t1 / p.Ho;
          to = t2 % p.To; n = t2 / p.To;
        }
        const int ti0 = to * p.sT - p.pT, hi0 = ho * p.sH - p.pH, wi0 = wo * p.sW - p.pW;
        const long long pos0 = (((long long)n * p.Ti + ti0) * p.Hi + hi0) * p.Wi + wi0;
        e.x = (int)(unsigned)(pos0 * p.x_ld * 2);  // exact modulo 2^32 whenever the tap is valid
        unsigned mt = 0u, mh = 0u, mw = 0u;
        for (int a = 0; a < p.kT; ++a) mt |= ((unsigned)(ti0 + a) < (unsigned)p.Ti ? 1u : 0u) << a;
        for (int a = 0; a < p.kH; ++a) mh |= ((unsigned)(hi0 + a) < (unsigned)p.Hi ? 1u : 0u) << a;
        for (int a = 0; a < p.kW; ++a) mw |= ((unsigned)(wi0 + a) < (unsigned)p.Wi ? 1u : 0u) << a;
        unsigned mk = 0u;
        int tp = 0;
        for (int a = 0; a < p.kT; ++a)
          for (int b = 0; b < p.kH; ++b) {
            const unsigned th = (mt >> a) & (mh >> b) & 1u;
            mk |= (th ? mw : 0u) << tp;
            tp += p.kW;
          }
        e.y = (int)mk;
      }
      tab[i] = e;
    }
  };

  auto dma = [&](int st, int stage) __attribute__((always_inline)) {
    const unsigned A = lds0 + (unsigned)(stage * STAGE);
    const unsigned B = A + IMG;
    const int rel0 = st * ROWS + 4 * wv + rg;  // position of row group 0, relative to pbeg
    // the row-table entries of all RG row groups are read BEFORE the first copy is issued: the copies are asm
    // statements with a memory clobber, so a table read placed between two of them stays there -- four dependent
    // ds_read -> wait -> copy round trips per k-step (round 4, read off the loop's ISA)
    int2 ent[RG];
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < RG; ++i) {
        const int rel = rel0 + 16 * i;
        ent[i] = rowtab[((rel >> 10) & 1) * WG_ROWTAB + (rel & (WG_ROWTAB - 1))];
      }
    }
#pragma unroll
    for (int i = 0; i < RG; ++i) {
      const int rel = rel0 + 16 * i;
      const int pp = pbeg + rel;
      const unsigned ok = (unsigned)mcol_ok & (unsigned)(pp < pend);
      dma16(dydesc, A + i * 4096, ok ? (unsigned)pp * dy_pitch + dycol : WG_OOB);
    }
#pragma unroll
    for (int i = 0; i < RG; ++i) {
      const int rel = rel0 + 16 * i;
      const int pp = pbeg + rel;
      unsigned ok = (unsigned)ncol_ok & (unsigned)(pp < pend);
      unsigned off;
      if (MODE == 0) {
        off = (unsigned)pp * x_pitch + xtap;
      } else {
        ok &= ((unsigned)ent[i].y >> tap) & 1u;
        off = (unsigned)ent[i].x + xtap;
      }
      dma16(xdesc, B + i * 4096, ok ? off : WG_OOB);
    }
  };

  const int wm = wv / WN, wn = wv % WN;
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp4 = li & 3;
  f32x4 acc[MR][NR];
#pragma unroll
  for (int a = 0; a < MR; ++a)
#pragma unroll
    for (int b = 0; b < NR; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // AOL: a lane's x fragment of n-tile b holds 8 positions of ONE input channel (the column it also stores)
  float asc[NR], ash[NR];
#pragma unroll
  for (int b = 0; b < NR; ++b) {
    const int col = n0 + wn * TN + b * 16 + li;
    asc[b] = (AOL && col < p.Kp) ? p.in_scale[col] : 0.f;
    ash[b] = (AOL && col < p.Kp) ? p.in_shift[col] : 0.f;
  }
  auto compute = [&](int stage) __attribute__((always_inline)) {
    const char* A = smem + stage * STAGE;
    const char* B = A + IMG;
#pragma unroll
    for (int ks = 0; ks < ROWS / 32; ++ks) {
      const int row = ks * 32 + 8 * g + q;  // this lane supplies row `row` (and row+4)
      const int swz = ((row & 3) << 2) | (((row >> 3) & 1) << 4);
      bf16x8 af[MR], bfr[NR];
      typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
      for (int a = 0; a < MR; ++a) {
        const int unit = ((wm * TM + a * 16) >> 2) + pp4;
        const lds_s16x4* ptr = (const lds_s16x4*)(A + row * 256 + ((unit ^ swz) << 3));
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)ptr);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ptr + 128));
        af[a] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int b = 0; b < NR; ++b) {
        const int unit = ((wn * TN + b * 16) >> 2) + pp4;
        const lds_s16x4* ptr = (const lds_s16x4*)(B + row * 256 + ((unit ^ swz) << 3));
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)ptr);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ptr + 128));
        bfr[b] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        if (AOL) bfr[b] = aol_frag_n(bfr[b], asc[b], ash[b]);
      }
#pragma unroll
      for (int a = 0; a < MR; ++a)
#pragma unroll
        for (int b = 0; b < NR; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    }
  };

  if (MODE == 1) {
    build_tab(0);
    __syncthreads();
  }
#pragma unroll
  for (int d = 0; d < D; ++d) dma(d, d);
  int st_c = 0, st_l = D;
  for (int st = 0; st < nsteps; ++st) {
    if (!(p.dbg & 4)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"((D - 1) * L) : "memory");
    if (!(p.dbg & 8)) __builtin_amdgcn_s_barrier();  // tile st landed everywhere; stage st_l and the old table are free
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 1 && (st % SPC) == 0 && pbeg + (st / SPC + 1) * WG_ROWTAB < pend && !(p.dbg & 32))
      build_tab(st / SPC + 1);  // next chunk's table, first read >= 13 steps from now
    if (!(p.dbg & 2)) dma(st + D, st_l);
    __builtin_amdgcn_sched_barrier(0);
    if (!(p.dbg & 1)) compute(st_c);
    __builtin_amdgcn_sched_barrier(0);
    st_c = (st_c + 1 == NS) ? 0 : st_c + 1;
    st_l = (st_l + 1 == NS) ? 0 : st_l + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((p.dbg & 64) && acc[0][0][0] != 123.456f) return;

  float* dst = p.out + (long long)s * p.Cout * p.Kp;
#pragma unroll
  for (int a = 0; a < MR; ++a)
#pragma unroll
    for (int b = 0; b < NR; ++b) {
      const int col = n0 + wn * TN + b * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * TM + a * 16 + g * 4 + r;
        if (row < p.Cout && col < p.Kp) dst[(long long)row * p.Kp + col] = acc[a][b][r];
      }
    }
}

template <int BM, int BN, int WM, int WN, int MODE, int NS, bool AOL = false, int ROWS = 64>
__global__ __launch_bounds__(256) void conv_wgrad_ring_kernel(WgradP p) {
  conv_wgrad_ring_body<BM, BN, WM, WN, MODE, NS, AOL, ROWS>(p, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// Deep-pipeline variant (round 4).  The ring kernel above keeps ONE 64-position step in flight behind
// `vmcnt(0)` + barrier: its copy stream and its MFMAs do not overlap inside a block (ablation,
// profiles/r02_wgrad_ablation.txt: copies alone 17 us, MFMAs alone 13.5 us, both 32.6 us on s4.b).  Here:
//   * 512 threads = 8 waves as 2 (Cout) x 4 (taps * Cin), output tile 128 x 256, wave tile 64 x 64
//     (16 v_mfma_f32_16x16x32_bf16 per 32 positions; 96 FLOP per staged byte instead of 64);
//   * the reduction is cut into UNITS of 32 positions = [dY 32 x 128 ch | x 32 x 128 ch | x 32 x 128 ch]
//     (24 KiB, whole 256-byte rows), six unit slots in LDS; one phase per unit: counted vmcnt, ONE raw
//     barrier, the copies of the unit six ahead (into the slot whose fragments were read two phases ago),
//     the transposing fragment reads (ds_read_b64_tr_b16) of the NEXT unit, 16 MFMAs on fragments already in
//     registers -- five units (120 KiB) in flight;
//   * the waves of one half (wv >> 2; one wave of each half per SIMD) issue a whole unit's 24 copies, the halves
//     alternating by unit: every phase has a loading and a purely multiplying wave per SIMD (conv_deep.hip);
//   * the position table (byte offset + tap mask per position) covers 512 positions, double buffered, rebuilt
//     every 16 phases one chunk ahead; the fp32 tile leaves through LDS as 16-byte stores.
// Same slab layout / reduce as the other kernels; a different (fixed) summation order over the positions.
// ---------------------------------------------------------------------------------------------
#define WGD_UNIT (3 * 8192)
#define WGD_NU 6
#define WGD_TAB 512
#define WGD_EP 260

// STAG: every wave copies 3 of a unit's 24 pieces (rows wv * 4 .. + 3 of the three images) instead of the waves of one half
// copying 6 each: waves 0-3 issue theirs when the phase opens (copies, then MFMAs), waves 4-7 after their MFMAs (MFMAs,
// then copies) -- the two waves of a SIMD run half a phase apart and the CU's address path sees one half at a time.
// (blk / nblk: this block's index and the block count of THIS problem's grid -- blockIdx.x / gridDim.x of the stand-alone
//  launch, a sub-range of the grid in the grouped launch below)
template <int MODE, bool STAG>
__device__ __forceinline__ void conv_wgrad_deep_body(const WgradP& p, const int blk, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int hf = wv >> 2, wq = wv & 3;
  int bid = blk;
  if (p.xcd_order) {
    const int nwg = nblk, q = nwg >> 3, r = nwg & 7, xcd = blk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blk >> 3);
  }
  const int ntile = p.tilesM * p.tilesN;
  const int s = bid / ntile;
  bid -= s * ntile;
  const int tn = bid % p.tilesN, tm = bid / p.tilesN;
  const int m0 = tm * 128, n0 = tn * 256;
  const int pbeg = s * p.rows_per_split;
  const int pend = min(p.P, pbeg + p.rows_per_split);
  const int nu = (pend - pbeg + 31) >> 5;
  int2* rowtab = (int2*)(smem + WGD_NU * WGD_UNIT);  // [2][WGD_TAB] (byte offset, tap mask)

  // ---- copy side: instruction j (0, 1) of wave wq covers rows (j * 4 + wq) * 4 + rg of an image, lane -> 16-byte chunk cc
  const int rg = lane >> 4, cc = lane & 15;
  // logical chunk fetched into physical chunk cc (row swizzle: (row & 3) << 1 | ((row >> 3) & 1) << 3)
  const int lc = cc ^ ((rg << 1) | (((STAG ? wv : wq) >> 1 & 1) << 3));
  const bool mcol_ok = (m0 + lc * 8) < p.Cout;
  const unsigned dycol = (unsigned)((m0 + lc * 8) * 2);
  bool ncol_ok[2];
  int tap[2];
  unsigned xtap[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ncol = n0 + i * 128 + lc * 8;
    ncol_ok[i] = ncol < p.Kp;
    tap[i] = 0;
    xtap[i] = (unsigned)(ncol * 2);
    if (MODE == 1 && ncol_ok[i]) {
      tap[i] = ncol / p.Cin;
      const int c0 = ncol - tap[i] * p.Cin;
      const int dw = tap[i] % p.kW, t2 = tap[i] / p.kW;
      const int dh = t2 % p.kH, dt = t2 / p.kH;
      xtap[i] = (unsigned)(((((long long)dt * p.Hi + dh) * p.Wi + dw) * p.x_ld + c0) * 2);
    }
  }
  const unsigned dy_pitch = (unsigned)(p.dy_ld * 2), x_pitch = (unsigned)(p.x_ld * 2);
  typedef __attribute__((address_space(3))) char* lds_ptr_t;
  const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem + (unsigned)(STAG ? wv : wq) * 1024u;
  auto rsrc_words = [](const void* base, unsigned bytes) __attribute__((always_inline)) {
    const unsigned long a = (unsigned long)base;
    return (i32x4){(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
  };
  const i32x4 xdesc = rsrc_words(p.x, p.x_bytes), dydesc = rsrc_words(p.dy, p.dy_bytes);
  auto dma16 = [](const i32x4& desc, unsigned lds_addr, unsigned voff) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(desc)
                 : "memory");
  };
  const bool small_p = p.P < (1 << 24);
  const float rcpWo = 1.0f / (float)p.Wo, rcpHo = 1.0f / (float)p.Ho, rcpTo = 1.0f / (float)p.To;
  auto build_tab = [&](int chunk) __attribute__((always_inline)) {  // 512 positions, one per thread
    int2* tab = rowtab + (chunk & 1) * WGD_TAB;
    const int pp = pbeg + chunk * WGD_TAB + tid;
    int2 e = make_int2(0, 0);
    if (pp < pend) {
      int wo, t1, ho, t2, to, n;
      if (small_p) {
        fast_divmod(pp, p.Wo, rcpWo, t1, wo);
        fast_divmod(t1, p.Ho, rcpHo, t2, ho);
        fast_divmod(t2, p.To, rcpTo, n, to);
      } else {
        wo = pp % p.Wo; t1 = pp / p.Wo;
        ho = t1 % p.Ho; t2 = t1 / p.Ho;
        to = t2 % p.To; n = t2 / p.To;
      }
      const int ti0 = to * p.sT - p.pT, hi0 = ho * p.sH - p.pH, wi0 = wo * p.sW - p.pW;
      const long long pos0 = (((long long)n * p.Ti + ti0) * p.Hi + hi0) * p.Wi + wi0;
      e.x = (int)(unsigned)(pos0 * p.x_ld * 2);  // exact modulo 2^32 whenever the tap is valid
      unsigned mt = 0u, mh = 0u, mw = 0u;
      for (int a = 0; a < p.kT; ++a) mt |= ((unsigned)(ti0 + a) < (unsigned)p.Ti ? 1u : 0u) << a;
      for (int a = 0; a < p.kH; ++a) mh |= ((unsigned)(hi0 + a) < (unsigned)p.Hi ? 1u : 0u) << a;
      for (int a = 0; a < p.kW; ++a) mw |= ((unsigned)(wi0 + a) < (unsigned)p.Wi ? 1u : 0u) << a;
      unsigned mk = 0u;
      int tp = 0;
      for (int a = 0; a < p.kT; ++a)
        for (int b = 0; b < p.kH; ++b) {
          const unsigned th = (mt >> a) & (mh >> b) & 1u;
          mk |= (th ? mw : 0u) << tp;
          tp += p.kW;
        }
      e.y = (int)mk;
    }
    tab[tid] = e;
  };
  // the 6 copies of this wave's share of unit u (only the half u & 1 issues): dY, x image 0, x image 1
  auto dma_unit = [&](int u, int slot) __attribute__((always_inline)) {
    if (!STAG && (u & 1) != hf) return;
    const unsigned U = lds0 + (unsigned)(slot * WGD_UNIT);
#pragma unroll
    for (int j = 0; j < (STAG ? 1 : 2); ++j) {
      const int rel = u * 32 + (STAG ? wv * 4 : j * 16 + wq * 4) + rg;  // position relative to pbeg
      const int pp = pbeg + rel;
      const bool pok = pp < pend;
      int2 e = make_int2(0, 0);  // (read before the first copy: the copies are asm statements with a memory clobber)
      if (MODE == 1) e = rowtab[((rel >> 9) & 1) * WGD_TAB + (rel & (WGD_TAB - 1))];
      dma16(dydesc, U + j * 4096, (mcol_ok && pok) ? (unsigned)pp * dy_pitch + dycol : WG_OOB);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        unsigned ok = (unsigned)ncol_ok[i] & (unsigned)pok;
        unsigned off;
        if (MODE == 0) {
          off = (unsigned)pp * x_pitch + xtap[i];
        } else {
          ok &= ((unsigned)e.y >> tap[i]) & 1u;
          off = (unsigned)e.x + xtap[i];
        }
        dma16(xdesc, U + (1 + i) * 8192 + j * 4096, ok ? off : WG_OOB);
      }
    }
  };

  // ---- compute side
  const int wm = wv >> 2, wn = wv & 3;
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp4 = li & 3;
  const int frow = 8 * g + q;  // this lane supplies unit rows frow and frow + 4
  const int fswz = ((frow & 3) << 2) | (((frow >> 3) & 1) << 4);
  unsigned aoff[4], boff[4];  // byte offsets of the lane's fragment sources inside a unit
#pragma unroll
  for (int a = 0; a < 4; ++a) aoff[a] = (unsigned)(frow * 256 + (((((wm * 64 + a * 16) >> 2) + pp4) ^ fswz) << 3));
#pragma unroll
  for (int b = 0; b < 4; ++b)
    boff[b] = (unsigned)((1 + (wn >> 1)) * 8192 + frow * 256 + ((((((wn & 1) * 64 + b * 16) >> 2) + pp4) ^ fswz) << 3));
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 FA0[4], FB0[4], FA1[4], FB1[4];
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto read_frags = [&](int slot, bf16x8 (&fa)[4], bf16x8 (&fb)[4]) __attribute__((always_inline)) {
    const char* U = smem + slot * WGD_UNIT;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const lds_s16x4* ptr = (const lds_s16x4*)(U + aoff[a]);
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)ptr);
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ptr + 128));
      fa[a] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const lds_s16x4* ptr = (const lds_s16x4*)(U + boff[b]);
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)ptr);
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ptr + 128));
      fb[b] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
  };
  auto mma = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4]) __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  // before the fragment reads of unit u (issued 5 phases ago by half u & 1): that half's two younger units stay in flight
  auto open_phase = [&](int u) __attribute__((always_inline)) {
    if (STAG || hf == (u & 1)) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // (STAG: 4 younger units x 3 pieces)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto close_phase = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  if (MODE == 1) {
    build_tab(0);
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 5; ++u) dma_unit(u, u);
  open_phase(0);  // phase -1: unit 0 landed; issue unit 5; fragments of unit 0
  dma_unit(5, 5);
  read_frags(0, FA0, FB0);
  close_phase();
  int slot = 0;  // slot of unit ph
  // phase ph: multiply unit ph (fragments in registers), read unit ph + 1, issue unit ph + 6 into unit ph's slot
  auto phase = [&](auto par, int ph) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par)::value;
    open_phase(ph + 1);
    if (MODE == 1 && (ph & 15) == 0 && pbeg + ((ph >> 4) + 1) * WGD_TAB < pend) build_tab((ph >> 4) + 1);
    if (!STAG || hf == 0) dma_unit(ph + 6, slot);
    const int nslot = slot + 1 == WGD_NU ? 0 : slot + 1;
    if (PAR == 0) {
      read_frags(nslot, FA1, FB1);
      mma(FA0, FB0);
    } else {
      read_frags(nslot, FA0, FB0);
      mma(FA1, FB1);
    }
    if (STAG && hf == 1) {
      __builtin_amdgcn_sched_barrier(0);
      dma_unit(ph + 6, slot);
    }
    close_phase();
    slot = nslot;
  };
  int ph = 0;
  for (; ph + 1 < nu; ph += 2) {
    phase(std::integral_constant<int, 0>{}, ph);
    phase(std::integral_constant<int, 1>{}, ph + 1);
  }
  if (ph < nu) phase(std::integral_constant<int, 0>{}, ph);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // run-ahead copies (zeros nobody reads) before LDS is reused
  __syncthreads();

  // fp32 tile through LDS: D[m][n]: row = g * 4 + reg (cout), col = li (k' column)
  float* E = (float*)smem;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = wn * 64 + b * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) E[(wm * 64 + a * 16 + g * 4 + r) * WGD_EP + col] = acc[a][b][r];
    }
  __syncthreads();
  float* dst = p.out + (long long)s * p.Cout * p.Kp;
  for (int idx = tid; idx < 128 * 64; idx += 512) {
    const int row = idx >> 6, c4 = idx & 63;
    if (m0 + row < p.Cout && n0 + c4 * 4 < p.Kp)
      *(float4*)(dst + (long long)(m0 + row) * p.Kp + n0 + c4 * 4) = *(const float4*)(E + row * WGD_EP + c4 * 4);
  }
}

template <int MODE, bool STAG>
__global__ __launch_bounds__(512) void conv_wgrad_deep_kernel(WgradP p) {
  conv_wgrad_deep_body<MODE, STAG>(p, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// Grouped launch (round 5): the weight gradients of several convolutions -- a ResBlock's a, b, c and
// shortcut -- as ONE grid of deep-pipeline blocks.  The problems' block ranges lie one after the other (first[j] ..
// first[j + 1]); a block finds its problem by a scan of at most WGG_MAX entries and runs the deep body on that problem's
// WgradP with its index inside the range.  Together the problems fill the chip with far fewer position splits than
// each alone (a slow s4 block: 50 output tiles -> 5 splits instead of 10 / 14 / 24; slow s5: 200 tiles, no split at
// all), so fewer fp32 slabs are written and re-read, and one launch (+ one grouped slab reduce) replaces three or four
// (+ their reduces).  Same body, same summation order per problem for a given split count.
// ---------------------------------------------------------------------------------------------
#define WGG_MAX 20
// Jobs (round 6, opt-in: VS_WGG_JOBS=1): a job = the output tiles of ONE position split of ONE problem -- blocks that
// stream the same dY / x rows.  Block b runs on XCD b % 8 (round-robin dispatch); the sequence of all jobs' tiles is
// cut into chunks of 32 (one residency round of one XCD) and chunk c is round c / 8 of XCD c % 8, so the blocks that
// re-read a slab sit on ONE L2 at the same time instead of on eight.  With the default mapping (problem ranges back
// to back, a split's tiles on consecutive blocks = on eight different XCDs) a grouped launch fetches 2x the bytes
// from beyond L2 -- and takes the same time (see the switch in vs_conv_wgrad_group).
#define WGG_JOBS 16  // table entries per XCD
struct WgGroupP {
  int n;
  int first[WGG_MAX + 1];  // block range of problem j starts at first[j] (a multiple of 8 with the XCD order) ...
  int count[WGG_MAX];      // ... and holds count[j] blocks (the rest of the range up to first[j + 1] exits at once)
  int mode[WGG_MAX];       // 0 pointwise, 1 gather
  int jobs;                // 1: the job table maps the blocks; 0: the problem ranges above do
  unsigned short jobend[8];        // slots (blocks) of each XCD
  unsigned jobtab[8][WGG_JOBS];    // first slot (9 bits) | problem (5) | split (6) | first tile (12); unused: ~0
  WgradP p[WGG_MAX];
};
static_assert(sizeof(WgGroupP) <= 4096, "kernel arguments");

__global__ __launch_bounds__(512) void conv_wgrad_deep_group_kernel(WgGroupP g) {
  const int b = blockIdx.x;
  if (g.jobs) {
    const int xcd = b & 7, slot = b >> 3;
    // (static indices into the kernel arguments only: a run-time index into the by-value struct read 8-byte entries
    //  4 bytes off on hipcc 7.2 -- every table word is selected by a uniform compare instead)
    unsigned e = ~0u;
    int jend = 0;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const bool mine = x == xcd;
      jend = mine ? (int)g.jobend[x] : jend;
#pragma unroll
      for (int i = 0; i < WGG_JOBS; ++i) {
        const unsigned t = g.jobtab[x][i];
        if (mine && (int)(t >> 23) <= slot) e = t;  // (unused entries: first slot 511; the last match wins)
      }
    }
    if (slot >= jend) return;
    const int j = __builtin_amdgcn_readfirstlane((int)((e >> 18) & 31));
    const int sp = (int)((e >> 12) & 63), tile0 = (int)(e & 4095);
    const int tile = tile0 + slot - (int)(e >> 23);
    const int blk = __builtin_amdgcn_readfirstlane(sp * (g.p[j].tilesM * g.p[j].tilesN) + tile);
#ifdef VS_WGG_DEBUG
    if (j >= g.n || sp >= g.p[j].S || tile >= g.p[j].tilesM * g.p[j].tilesN) {
      if (threadIdx.x == 0) printf("bad job: b %d xcd %d slot %d e %x j %d sp %d tile %d (n %d S %d nt %d)\n", b, xcd, slot, e, j, sp, tile, g.n, g.p[j].S, g.p[j].tilesM * g.p[j].tilesN);
      return;
    }
#endif
    if (g.mode[j] == 0) conv_wgrad_deep_body<0, true>(g.p[j], blk, g.count[j]);
    else conv_wgrad_deep_body<1, true>(g.p[j], blk, g.count[j]);
    return;
  }
  int j = 0;
#pragma unroll
  for (int i = 1; i < WGG_MAX; ++i)
    if (i < g.n && b >= g.first[i]) j = i;
  j = __builtin_amdgcn_readfirstlane(j);
  const int lo = g.first[j], cnt = g.count[j];
  if (b - lo >= cnt) return;
  if (g.mode[j] == 0) conv_wgrad_deep_body<0, true>(g.p[j], b - lo, cnt);
  else conv_wgrad_deep_body<1, true>(g.p[j], b - lo, cnt);
}

// the slab reduces of a grouped launch as one grid: entry e owns the virtual 256-thread blocks [first[e], first[e + 1])
struct WgReduceGroupP {
  int n;
  long long first[WGG_MAX + 1];
  const float* slabs[WGG_MAX];
  float* dw[WGG_MAX];
  long long len[WGG_MAX];
  int S[WGG_MAX];
};

__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(WgReduceGroupP g) {
  __shared__ float4 part[16][17];
  const long long b = blockIdx.x;
  int e = 0;
#pragma unroll
  for (int i = 1; i < WGG_MAX; ++i)
    if (i < g.n && b >= g.first[i]) e = i;
  wgrad_reduce_body(g.slabs[e], g.dw[e], g.len[e], g.S[e], part, b - g.first[e], threadIdx.x);
}

// dw[i] = sum_s slab[s][i], bitwise reproducible: block = 16 float4 columns x 16 slab slices,
// each slice summed in order, the 16 slice sums combined in order through LDS.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* slabs, float* dw,
                                                          long long n, int S) {
  __shared__ float4 part[16][17];
  wgrad_reduce_body(slabs, dw, n, S, part, blockIdx.x, threadIdx.x);
}

// ---- deferred slab reduce (round 3) ----------------------------------------------------------------------------
// Between vs_wgrad_reduce_defer(1) and (0) a slab reduce issued behind a weight gradient is not launched: it waits, one
// slot per host thread, for the next vs_bn_bwd_finalize on the same stream, which launches ONE grid holding both
// (bn_pool.hip) -- the two ~5 us kernels are neighbours on the stream and independent of each other.  Whatever would
// read the slabs' workspace or dw earlier flushes it: a second reduce, a weight gradient handed the same workspace,
// vs_wgrad_reduce_flush() (the trunk calls it at the end of a backward segment), switching the mode off.
namespace {
struct PendingState {
  int mode = 0;  // 0 off, 1 defer, 2 suspended (no stash, no flush: launches on a side lane)
  bool have = false;
  VsPendingReduce r{};
};
thread_local PendingState g_pending;

static void pending_launch(const VsPendingReduce& r) {
  const long long grid = wgrad_reduce_vblocks(r.n, r.S);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)grid), dim3(256), 0, r.st, r.slabs, r.dw, r.n, r.S);
}
}  // namespace

static void pending_flush() {
  if (g_pending.have) {
    g_pending.have = false;
    pending_launch(g_pending.r);
  }
}

// true: the reduce was stashed (the caller must not launch it)
static bool pending_stash(const float* slabs, float* dw, long long n, int S, hipStream_t st) {
  if (g_pending.mode != 1) return false;
  pending_flush();
  g_pending.r = VsPendingReduce{slabs, dw, n, S, st};
  g_pending.have = true;
  return true;
}

bool vs_pending_reduce_take(hipStream_t st, VsPendingReduce* out) {
  if (!g_pending.have || g_pending.r.st != st) return false;
  *out = g_pending.r;
  g_pending.have = false;
  return true;
}

extern "C" int vs_wgrad_reduce_defer(int mode) {
  VS_CHECK_ARG(mode >= 0 && mode <= 2, "mode 0 (off), 1 (defer), 2 (suspend)");
  if (mode == 0) pending_flush();
  g_pending.mode = mode;
  VS_CHECK_LAUNCH();
  return VS_OK;
}

extern "C" int vs_wgrad_reduce_flush(void) {
  pending_flush();
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// The same reduce for MANY layers in one launch (the slabs of every weight gradient of a backward segment are
// kept -- 288 GB of HBM: 1.5 GB of slabs per step is nothing -- and summed by one launch at the end of the
// segment instead of one 6 us launch behind every wgrad).  table[i] = {slab base, dw base (addresses), n
// elements, S, first block}; a block finds its entry by one binary search; then exactly wgrad_reduce_kernel's
// order (bitwise the per-layer reduce).
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const long long* __restrict__ table, int nent) {
  __shared__ float4 part[16][17];
  const long long b = blockIdx.x;
  int lo = 0, hi = nent - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid * 5 + 4] <= b) lo = mid; else hi = mid - 1;
  }
  const float* slabs = (const float*)table[lo * 5 + 0];
  float* dw = (float*)table[lo * 5 + 1];
  const long long n = table[lo * 5 + 2];
  const int S = (int)table[lo * 5 + 3];
  const long long blk = b - table[lo * 5 + 4];
  const long long n4 = n >> 2;
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long long i = blk * 16 + col;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const int per = (S + 15) / 16;
    const int s0 = sl * per, s1 = min(S, s0 + per);
    int s = s0;
    for (; s + 8 <= s1; s += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(slabs + (long long)(s + u) * n + i * 4);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc.x += v[u].x;
        acc.y += v[u].y;
        acc.z += v[u].z;
        acc.w += v[u].w;
      }
    }
    for (; s < s1; ++s) {
      const float4 v = *(const float4*)(slabs + (long long)s * n + i * 4);
      acc.x += v.x;
      acc.y += v.y;
      acc.z += v.z;
      acc.w += v.w;
    }
  }
  part[sl][col] = acc;
  __syncthreads();
  if (sl == 0 && i < n4) {
    float4 t = part[0][col];
    for (int k = 1; k < 16; ++k) {
      const float4 v = part[k][col];
      t.x += v.x;
      t.y += v.y;
      t.z += v.z;
      t.w += v.w;
    }
    *(float4*)(dw + i * 4) = t;
  }
}

extern "C" int vs_wgrad_reduce_batched(const int64_t* table, int n_entries, int64_t total_blocks, void* stream) {
  VS_CHECK_ARG(table && n_entries > 0 && total_blocks > 0, "bad args");
  hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)table, n_entries);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

extern "C" int64_t vs_wgrad_reduce_blocks(int64_t n) { return (n / 4 + 15) / 16; }

extern "C" int vs_wgrad_reduce(const float* slabs, float* dw, int64_t n, int splits, void* stream) {
  VS_CHECK_ARG(slabs && dw && n > 0 && splits > 1, "bad args");
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)wgrad_reduce_vblocks(n, splits)), dim3(256), 0, (hipStream_t)stream,
                     slabs, dw, (long long)n, splits);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// ------------------------------ host side ------------------------------------
struct WgCfg {
  int bm, bn, S, rows_per_split, tilesM, tilesN;
  int deep;  // conv_wgrad_deep_kernel (128 x 256 tile, 512 threads)
};

// (VS_WGRAD_NODEEP / VS_WGRAD_FORCEDEEP: include/vidsitu_hip.h)

// Deep-pipeline plan: 128 x 256 output tiles, one block per CU (152 KiB of LDS), the positions split so that the grid
// is one residency round.  Where it pays (profiles/r04_wgrad_deep.txt): >= 128 output channels and >= 192 columns
// mostly filling their tiles, enough tiles x splits for >= half the chip, >= 16 units of 32 positions per block.
static bool wg_deep_plan(const vs_conv_desc* d, WgCfg* c) {
  static const int mode = [] { const char* e = getenv("VS_WGRAD_DEEP"); return e ? atoi(e) : 1; }();
  const bool force = mode == 2 || (d->flags & VS_WGRAD_FORCEDEEP);
  if (mode == 0 || (d->flags & VS_WGRAD_NODEEP)) return false;
  if (((d->flags >> 8) & 0xf) || ((d->flags >> 16) & 7) || ((d->flags >> 24) & 0xff)) return false;  // forced tile / ring / slots
  const int taps = d->kT * d->kH * d->kW;
  const int Kp = taps * d->Cin;
  const long long P = (long long)d->N * d->To * d->Ho * d->Wo;
  if (taps > 31 || Kp % 8 != 0 || d->Cout % 8 != 0) return false;
  const int tilesM = (d->Cout + 127) / 128, tilesN = (Kp + 255) / 256;
  const long long tiles = (long long)tilesM * tilesN;
  const double row_eff = (double)d->Cout / (128.0 * tilesM), col_eff = (double)Kp / (256.0 * tilesN);
  long long S = 256 / tiles;
  const long long maxS = P / 512;  // >= 16 units per block
  if (S > maxS) S = maxS;
  const long long slab_cap = (64ll << 20) / ((long long)d->Cout * Kp * 4);
  if (S > slab_cap) S = slab_cap;
  if (S < 1) S = 1;
  long long rps = (P + S - 1) / S;
  rps = (rps + 31) / 32 * 32;
  S = (P + rps - 1) / rps;
  if (!force && (row_eff < 0.9 || col_eff < 0.75 || tiles * S < 128 || tiles > 256 || rps < 512)) return false;
  // Position floor.  Mid-round 4 the slow-pathway s4 / s5 layers at 8 clips per GPU (12 544 / 3 136 positions) LOST with
  // this kernel in the step (-0.4 %, floor 40 000: profiles/r04_wgrad_deep.txt) although it was 1.03-1.38x the ring
  // kernel alone -- one 152-KiB block per CU shares no CU with the data gradient beside it.  After the round's later
  // fixes (row-table entries read before the copy statements, the column-form slab reduce, the ring loop of the paired
  // data gradient without accumulator moves) the same A/B reads +0.5-0.7 % with them on the deep kernel (floor 10 000:
  // 668.3-669.9 vs 662.1-666.5 clips/s; floor 3 000: another +0.2 %; 0 = 3 000; VS_WGRAD_DEEP=2 everywhere: -2.7 %;
  // profiles/r04_wgrad_deep_floor.txt).  From 32 clips on the step gains 2.5 %.
  static const long long min_p = [] { const char* e = getenv("VS_WGRAD_DEEP_MINP"); return e ? atoll(e) : 3000ll; }();
  if (!force && P < min_p) return false;
  if (tiles * S > 65535) return false;
  c->bm = 128;
  c->bn = 256;
  c->tilesM = tilesM;
  c->tilesN = tilesN;
  c->S = (int)S;
  c->rows_per_split = (int)rps;
  c->deep = 1;
  return true;
}

static const int kWgTiles[8][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}, {32, 128}, {32, 64}, {16, 128}, {16, 64}};

static WgCfg wg_pick(const vs_conv_desc* d) {
  WgCfg c;
  c.deep = 0;
  if (wg_deep_plan(d, &c)) return c;
  const int Kp = d->kT * d->kH * d->kW * d->Cin;
  const long long P = (long long)d->N * d->To * d->Ho * d->Wo;
  c.bm = d->Cout >= 128 ? 128 : (d->Cout >= 64 ? 64 : (d->Cout >= 32 ? 32 : 16));
  c.bn = Kp >= 128 ? 128 : 64;
  // <= 16 output channels (fast pathway s2 / s3): a 128-column tile as soon as Kp > 64 -- 72 or 96 columns as 64 + a
  // nearly empty second tile doubled the blocks that re-read dY; 32 channels: 64 columns (tools/wgrad_small_sweep.py,
  // profiles/r02_wgrad_small_sweep.txt: s2.p1.b 41.6 -> 27.1 us, s2.p1.a 45.7 -> 29.4, s4.p1.a 15.3 -> 13.7)
  if (d->Cout <= 16) c.bn = Kp > 64 ? 128 : 64;
  else if (d->Cout <= 32) c.bn = 64;
  // (Tried: for the few-position layers of slow s5 -- 3 136 positions, 2-12 MB of dW -- output tiles small enough to
  // fill the chip with no position split at all (64 x 128 / 64 x 64, no slabs, no reduce): slower, 52 vs 43-48 us
  // on s5.b, 29.6 vs 23 us on s5.c -- tools/wgrad_sweep.py, profiles/r02_wgrad_sweep.txt.)
  const int forced = (d->flags >> 8) & 0xf;  // experiments / tests: tile id + 1 in bits 8..11
  if (forced >= 1 && forced <= 8) {
    c.bm = kWgTiles[forced - 1][0];
    c.bn = kWgTiles[forced - 1][1];
  }
  c.tilesM = (d->Cout + c.bm - 1) / c.bm;
  c.tilesN = (Kp + c.bn - 1) / c.bn;
  const long long tiles = (long long)c.tilesM * c.tilesN;
  // ~384 blocks (1.5 per CU); every block keeps >= 8 steps (512 positions) so the slab traffic
  // (S x |dW| fp32, written and re-read) stays small next to the operand reads.  Alone on the GPU a
  // 3x3 / temporal weight gradient is fastest with two blocks per CU (512 slots; pointwise ones with 384),
  // but in the training step it runs on a side lane beside the same unit's dgrad, where a smaller grid
  // is faster for the whole step (A/B on one box: 512 slots 14.18-14.27 ms, 384: 14.10-14.12,
  // 256: 14.07-14.14, 192: 14.13-14.22, 128: 14.46-14.54) and writes fewer slabs; 384 keeps most of the
  // stand-alone speed (256 costs the kernel alone 13 %).
  // Skinny outputs (fast pathway: a handful of tiles, 10^5..10^6 positions) get one residency round too: 512
  // blocks (2048, several rounds of short blocks, was 25-50 % slower; 256 / 384 / 768 slower as well).
  static const long long slots = [] {
    const char* e = getenv("VS_WGRAD_SLOTS");  // experiment knob: resident block slots to fill
    return e ? atoll(e) : 384ll;
  }();
  static const long long small_slots = [] {
    const char* e = getenv("VS_WGRAD_SLOTS_SMALL");
    return e ? atoll(e) : 512ll;
  }();
  long long target = (d->Cout <= 32) ? small_slots : slots;
  const int forced_slots = (d->flags >> 24) & 0xff;  // experiments: resident block slots / 8 in bits 24..31
  if (forced_slots) target = 8ll * forced_slots;
  // Round the split DOWN so that the grid fits one residency round (two blocks are resident per CU,
  // 80 KB of LDS each) -- 528 blocks on 512 slots run as two rounds (s4.a: +45 %)
  long long S = target / tiles;
  // Layers whose output alone has >= 128 tiles (slow s5: conv a 192, conv b 144) run UNSPLIT: the split of 2 they used
  // to get wrote and re-read two fp32 slabs of a 9-13 MB dW and put a reduce launch behind the kernel; with their
  // dgrad's tiles in the same grid (conv_pair.hip) the chip is full either way.  Train step 12.311-12.314 ms against
  // 12.321-12.361 (alternating, one box: profiles/r03_wgrad_unsplit.txt); a threshold of 96 tiles is slower (12.43).
  // VS_WGRAD_S1_TILES overrides (0 = never).
  static const long long s1_tiles = [] {
    const char* e = getenv("VS_WGRAD_S1_TILES");
    return e ? atoll(e) : 128ll;
  }();
  if (s1_tiles > 0 && tiles >= s1_tiles && !forced_slots) S = 1;
  const long long maxS = (P + 511) / 512;
  if (S > maxS) S = maxS;
  const long long slab_cap = (64ll << 20) / ((long long)d->Cout * Kp * 4);  // <= 64 MB of slabs
  if (S > slab_cap) S = slab_cap;
  if (S > 1024) S = 1024;
  if (S < 1) S = 1;
  // (Rounds 3-4 measured two variants that are gone from the tree: splits rounded to whole multiples of 8, one per XCD
  //  -- VS_WGRAD_ALIGN8, 4 % slower over the layers: the ring kernel is not bound by L2 misses,
  //  profiles/r04_wgrad_xcd_align.txt -- and half-depth ring stages, 4 x 32 positions -- VS_WGRAD_HALF, -3.4 %,
  //  profiles/r04_wgrad_half_stages.txt.)
  long long rps = (P + S - 1) / S;
  rps = (rps + 63) / 64 * 64;
  S = (P + rps - 1) / rps;
  c.S = (int)S;
  c.rows_per_split = (int)rps;
  return c;
}

extern "C" size_t vs_conv_wgrad_workspace_bytes(const vs_conv_desc* d) {
  const WgCfg c = wg_pick(d);
  if (c.S <= 1) return 0;
  return (size_t)c.S * d->Cout * d->kT * d->kH * d->kW * d->Cin * sizeof(float);
}

// conv_pair.hip: between vs_conv_pair_begin / _end a 128 x 128 two-stage ring launch (and the slab reduce behind it) is
// recorded instead of issued
static bool pair_take_wgrad(const WgradP& p, int grid, size_t smem, int mode, hipStream_t st);
static bool pair_defer_reduce(const float* slabs, float* dw, long long n, int splits);
#ifndef VS_CONV_PAIR_TU  // compiled on its own (not through conv_pair.hip): nothing is ever recorded
static bool pair_take_wgrad(const WgradP&, int, size_t, int, hipStream_t) { return false; }
static bool pair_defer_reduce(const float*, float*, long long, int) { return false; }
#endif

template <int BM, int BN, int WM, int WN>
static int wg_launch(const WgradP& p, int mode, int ring, hipStream_t st) {
  const int grid = p.tilesM * p.tilesN * p.S;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<BM, BN, WM, WN, 0>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<BM, BN, WM, WN, 1>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_ring_kernel<BM, BN, WM, WN, 0, 2>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_ring_kernel<BM, BN, WM, WN, 1, 2>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_ring_kernel<BM, BN, WM, WN, 0, 3>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_ring_kernel<BM, BN, WM, WN, 1, 3>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  if (ring >= 2) {
    const size_t tab = mode ? 2 * WG_ROWTAB * sizeof(int2) : 0;
    const size_t smem = (size_t)ring * 2 * 64 * 256 + tab;
    if constexpr (BM == 128 && BN == 128) {
      if (ring == 2 && pair_take_wgrad(p, grid, smem, mode, st)) return VS_OK;
    }
    if (p.in_scale) {  // apply on load: wgrad_aol_ok admitted only (128-row tile, 2-stage ring, pointwise)
      if constexpr (BM == 128) {
        static std::once_flag aattr;
        std::call_once(aattr, [] {
          (void)hipFuncSetAttribute((const void*)conv_wgrad_ring_kernel<BM, BN, WM, WN, 0, 2, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
        hipLaunchKernelGGL((conv_wgrad_ring_kernel<BM, BN, WM, WN, 0, 2, true>), dim3(grid), dim3(256), smem, st, p);
        VS_CHECK_LAUNCH();
        return VS_OK;
      }
    }
    if (ring == 2 && mode == 0)
      hipLaunchKernelGGL((conv_wgrad_ring_kernel<BM, BN, WM, WN, 0, 2>), dim3(grid), dim3(256), smem, st, p);
    else if (ring == 2)
      hipLaunchKernelGGL((conv_wgrad_ring_kernel<BM, BN, WM, WN, 1, 2>), dim3(grid), dim3(256), smem, st, p);
    else if (mode == 0)
      hipLaunchKernelGGL((conv_wgrad_ring_kernel<BM, BN, WM, WN, 0, 3>), dim3(grid), dim3(256), smem, st, p);
    else
      hipLaunchKernelGGL((conv_wgrad_ring_kernel<BM, BN, WM, WN, 1, 3>), dim3(grid), dim3(256), smem, st, p);
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  const size_t smem = 2 * 2 * 64 * 256 + (mode ? WG_ROWTAB * sizeof(int4) : 0);
  if (mode == 0)
    hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WM, WN, 0>), dim3(grid), dim3(256), smem, st, p);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WM, WN, 1>), dim3(grid), dim3(256), smem, st, p);
  VS_CHECK_LAUNCH();
  return VS_OK;
}

// the launch apply on load is built for: pointwise, unit stride, the 128-row tiles on the two-stage ring
static bool wgrad_aol_plan_ok(const vs_conv_desc* d) {
  const bool dense = (d->kT * d->kH * d->kW == 1) && d->sT == 1 && d->sH == 1 && d->sW == 1 &&
                     d->pT == 0 && d->pH == 0 && d->pW == 0;
  if (!dense || ((d->flags >> 16) & 7) || ((d->flags >> 8) & 0xf)) return false;
  const WgCfg c = wg_pick(d);
  return c.bm == 128 && (c.bn == 128 || c.bn == 64);
}

// WgradP of one weight gradient from its descriptor and plan (tensor pointers, extents, tile counts, split)
static int wg_fill_params(WgradP& p, const void* dy, const void* x, float* out, const vs_conv_desc* d, const WgCfg& c) {
  p.dy = (const uint16_t*)dy;
  p.x = (const uint16_t*)x;
  p.in_scale = nullptr;
  p.in_shift = nullptr;
  p.out = out;
  p.P = d->N * d->To * d->Ho * d->Wo;
  p.Cout = d->Cout;
  p.Cin = d->Cin;
  p.Kp = d->kT * d->kH * d->kW * d->Cin;
  p.x_ld = d->x_ld;
  p.dy_ld = d->y_ld;
  p.To = d->To; p.Ho = d->Ho; p.Wo = d->Wo;
  p.Ti = d->Ti; p.Hi = d->Hi; p.Wi = d->Wi;
  p.kT = d->kT; p.kH = d->kH; p.kW = d->kW;
  p.sT = d->sT; p.sH = d->sH; p.sW = d->sW;
  p.pT = d->pT; p.pH = d->pH; p.pW = d->pW;
  {
    const long long xb = (long long)d->N * d->Ti * d->Hi * d->Wi * d->x_ld * 2;
    const long long db = (long long)p.P * d->y_ld * 2;
    VS_CHECK_ARG(xb < (1ll << 31) && db < (1ll << 31), "tensor larger than 2 GiB");
    p.x_bytes = (unsigned)xb;
    p.dy_bytes = (unsigned)db;
  }
  p.tilesM = c.tilesM;
  p.tilesN = c.tilesN;
  p.S = c.S;
  p.rows_per_split = c.rows_per_split;
  p.xcd_order = 0;
  p.dbg = 0;
  return VS_OK;
}

// ---- grouped launch: plan, workspace, entry point (kernel: conv_wgrad_deep_group_kernel) ----
static bool wgg_item_ok(const vs_conv_desc* d) {
  const int taps = d->kT * d->kH * d->kW;
  const long long Kp = (long long)taps * d->Cin;
  if (taps > 31 || Kp % 8 != 0 || d->Cout % 8 != 0 || d->x_ld % 8 != 0 || d->y_ld % 8 != 0) return false;
  if (((d->flags >> 8) & 0xf) || ((d->flags >> 16) & 7) || ((d->flags >> 24) & 0xff)) return false;  // forced plans
  const long long P = (long long)d->N * d->To * d->Ho * d->Wo;
  // the same 2-GiB tensor limit wg_fill_params enforces: a group declared OK must not fail inside the launch
  const long long xb = (long long)d->N * d->Ti * d->Hi * d->Wi * d->x_ld * 2;
  const long long db = P * d->y_ld * 2;
  return P >= 512 && xb < (1ll << 31) && db < (1ll << 31);
}

// Splits of a group.  Every candidate "units per block" u gives each problem S_j = round(units_j / u) splits; the grid is
// laid out longest blocks first (the block ranges of the problems in descending block length) and the candidate's cost is
// the makespan of that list on the chip's block slots (greedy: a block goes to the slot that frees first) plus what its
// slabs cost the reduce behind it.  Block cost = WGG_C0 units of prologue + epilogue + its 32-position units (one unit =
// ~0.64 us on the deep body); slabs at ~3 TB/s.  Measured (tools/wgrad_group_time.py): one residency round with a 4x
// longer problem in it (slow s5's first block: conv a runs on 4x the positions) loses to two rounds of balanced blocks,
// and 256 nominal blocks that round to 270 lose 12 % to 224 -- hence a simulated makespan instead of a block-count target.
// VS_WGG_SLOTS: block slots (default 256 = one 152-KiB block per CU).
#define WGG_C0 14
static bool wgg_plan(const vs_wgrad_item* it, int n, WgCfg* cfg, size_t* slab_off, size_t* slab_total, int* order) {
  if (n < 1 || n > WGG_MAX) return false;
  static const int slots = [] { const char* e = getenv("VS_WGG_SLOTS"); const int v = e ? atoi(e) : 256; return v < 1 ? 1 : (v > 1024 ? 1024 : v); }();
  long long units[WGG_MAX], P[WGG_MAX], maxS[WGG_MAX], dwn[WGG_MAX];
  int tiles[WGG_MAX];
  long long W = 0, umax = 0;
  for (int j = 0; j < n; ++j) {
    const vs_conv_desc* d = &it[j].d;
    if (!wgg_item_ok(d)) return false;
    const int Kp = d->kT * d->kH * d->kW * d->Cin;
    cfg[j].bm = 128;
    cfg[j].bn = 256;
    cfg[j].deep = 1;
    cfg[j].tilesM = (d->Cout + 127) / 128;
    cfg[j].tilesN = (Kp + 255) / 256;
    tiles[j] = cfg[j].tilesM * cfg[j].tilesN;
    P[j] = (long long)d->N * d->To * d->Ho * d->Wo;
    units[j] = (P[j] + 31) / 32;
    dwn[j] = (long long)d->Cout * Kp;
    maxS[j] = P[j] / 512;
    const long long slab_cap = (64ll << 20) / (dwn[j] * 4);
    if (maxS[j] > slab_cap) maxS[j] = slab_cap;
    if (maxS[j] < 1) maxS[j] = 1;
    W += (long long)tiles[j] * units[j];
    if (units[j] > umax) umax = units[j];
  }
  auto splits = [&](long long u, long long* S, long long* rps) {
    for (int j = 0; j < n; ++j) {
      long long s = (units[j] + u / 2) / u;
      if (s > maxS[j]) s = maxS[j];
      if (s < 1) s = 1;
      long long r = (P[j] + s - 1) / s;
      r = (r + 31) / 32 * 32;
      S[j] = (P[j] + r - 1) / r;
      rps[j] = r;
    }
  };
  auto cost = [&](const long long* S, const long long* rps) {
    // longest blocks first
    int ord[WGG_MAX];
    for (int j = 0; j < n; ++j) ord[j] = j;
    for (int a = 0; a < n; ++a)
      for (int b = a + 1; b < n; ++b)
        if (rps[ord[b]] > rps[ord[a]]) { const int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    long long slot[1024];
    for (int i = 0; i < slots; ++i) slot[i] = 0;
    long long span = 0;
    int nxt = 0;  // blocks of equal length arrive in runs: the slots are used round-robin from the earliest-free one
    for (int a = 0; a < n; ++a) {
      const int j = ord[a];
      const long long len = WGG_C0 + (rps[j] + 31) / 32;
      const long long nb = (long long)tiles[j] * S[j];
      for (long long b = 0; b < nb; ++b) {
        // earliest-free slot: with runs of equal blocks the slots fill in index order, so a linear probe from `nxt`
        // finds it in O(1) amortised; a full scan every `slots` blocks keeps it exact enough for a cost model
        int best = nxt;
        if ((b % slots) == 0) {
          for (int i = 0; i < slots; ++i)
            if (slot[i] < slot[best]) best = i;
        }
        slot[best] += len;
        if (slot[best] > span) span = slot[best];
        nxt = (best + 1) % slots;
      }
    }
    double slab = 0.0;
    for (int j = 0; j < n; ++j)
      if (S[j] > 1) slab += 2.0 * (double)S[j] * (double)dwn[j] * 4.0;  // written, then read by the reduce
    return (double)span + slab / 3.0e12 / 0.64e-6;
  };
  long long bestS[WGG_MAX], bestR[WGG_MAX], S[WGG_MAX], R[WGG_MAX];
  double best = -1.0;
  long long u = W / (3ll * slots);
  if (u < 16) u = 16;
  // many equally long tiles (or few slots) put the first candidate past the longest problem: the loop below must still
  // evaluate at least one plan -- start no higher than the no-split candidate u = umax (every S_j = 1)
  if (u > umax) u = umax;
  for (; u <= umax + 1; u += (u + 15) / 16) {
    splits(u, S, R);
    const double c = cost(S, R);
    if (best < 0.0 || c < best) {
      best = c;
      for (int j = 0; j < n; ++j) { bestS[j] = S[j]; bestR[j] = R[j]; }
    }
  }
  if (best < 0.0) return false;  // no candidate evaluated (cannot happen with the clamp above; never plan from garbage)
  size_t off = 0;
  long long blocks = 0;
  for (int j = 0; j < n; ++j) {
    cfg[j].S = (int)bestS[j];
    cfg[j].rows_per_split = (int)bestR[j];
    slab_off[j] = off;
    if (bestS[j] > 1) off += (size_t)bestS[j] * dwn[j] * sizeof(float);
    blocks += (long long)tiles[j] * bestS[j];
    order[j] = j;
  }
  for (int a = 0; a < n; ++a)
    for (int b = a + 1; b < n; ++b)
      if (bestR[order[b]] > bestR[order[a]]) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
  *slab_total = off;
  return blocks <= 65535;
}

extern "C" int vs_conv_wgrad_group_ok(const vs_wgrad_item* items, int n) {
  if (items == nullptr) return 0;
  WgCfg cfg[WGG_MAX];
  size_t off[WGG_MAX], tot;
  int order[WGG_MAX];
  return wgg_plan(items, n, cfg, off, &tot, order) ? 1 : 0;
}

extern "C" size_t vs_conv_wgrad_group_workspace_bytes(const vs_wgrad_item* items, int n) {
  if (items == nullptr) return 0;
  WgCfg cfg[WGG_MAX];
  size_t off[WGG_MAX], tot = 0;
  int order[WGG_MAX];
  if (!wgg_plan(items, n, cfg, off, &tot, order)) return 0;
  return tot;
}

extern "C" int vs_conv_wgrad_group(const vs_wgrad_item* items, int n, void* workspace, size_t ws_bytes, void* stream) {
  VS_CHECK_ARG(items != nullptr && n >= 1 && n <= WGG_MAX, "1 .. 20 items");
  WgCfg cfg[WGG_MAX];
  size_t off[WGG_MAX], tot = 0;
  int order[WGG_MAX];
  if (!wgg_plan(items, n, cfg, off, &tot, order)) {
    vs_set_error("vs_conv_wgrad_group: an item is outside the deep-pipeline kernel's envelope (ask vs_conv_wgrad_group_ok)");
    return VS_ERR_UNSUPPORTED;
  }
  if (tot > 0 && (workspace == nullptr || ws_bytes < tot)) {
    vs_set_error("vs_conv_wgrad_group: workspace too small (%zu < %zu)", ws_bytes, tot);
    return VS_ERR_WORKSPACE;
  }
  if (g_pending.have) pending_flush();  // (a deferred reduce of an earlier launch may read this workspace / write these dw)
  WgGroupP g;
  WgReduceGroupP r;
  g.n = n;
  r.n = 0;
  r.first[0] = 0;
  int first = 0;
  for (int a = 0; a < n; ++a) {  // grid position a = problem order[a]: longest blocks first
    const int j = order[a];
    const vs_wgrad_item& t = items[j];
    VS_CHECK_ARG(t.dy && t.x && t.dw, "null tensor");
    const vs_conv_desc* d = &t.d;
    float* out = cfg[j].S > 1 ? (float*)((char*)workspace + off[j]) : t.dw;
    const int rc = wg_fill_params(g.p[a], t.dy, t.x, out, d, cfg[j]);
    if (rc) return rc;
    const bool dense = (d->kT * d->kH * d->kW == 1) && d->sT == 1 && d->sH == 1 && d->sW == 1 && d->pT == 0 &&
                       d->pH == 0 && d->pW == 0;
    g.mode[a] = dense ? 0 : 1;
    // (An XCD-contiguous block order inside each problem -- its range padded to a multiple of 8, the tiles of one split on
    //  one XCD's L2 -- was measured: slow s4 / s5 blocks 4-9 % faster alone on the chip, the step 0.1-0.3 % slower with
    //  three-block groups; not kept.)
    g.first[a] = first;
    g.count[a] = cfg[j].tilesM * cfg[j].tilesN * cfg[j].S;
    first += g.count[a];
    if (cfg[j].S > 1) {
      const int e = r.n++;
      r.slabs[e] = out;
      r.dw[e] = t.dw;
      r.len[e] = (long long)d->Cout * g.p[a].Kp;
      r.S[e] = cfg[j].S;
      r.first[e + 1] = r.first[e] + wgrad_reduce_vblocks(r.len[e], r.S[e]);
    }
  }
  g.first[n] = first;
  for (int j = n; j < WGG_MAX; ++j) g.first[j + 1] = first;
  // ---- jobs -> XCDs (VS_WGG_JOBS=0: the round-5 block ranges, A/B) ----
  // All jobs' tiles in one sequence (the grid order: longest blocks first, a problem's splits one after the other, a
  // split's tiles consecutive), cut into chunks of 32 = one residency round of one XCD (32 CUs, one 152-KiB block
  // each); chunk c runs on XCD c % 8 as its round c / 8.  Every round of every XCD is full (the chip fills exactly as
  // with the round-5 order) and a job is cut at most once per chunk boundary: its slabs are fetched into one L2, or
  // two, instead of eight.
  g.jobs = 0;
  int grid = first;
  {
    // Default OFF: measured (profiles/r06_wgrad_jobs.txt) the mapping halves the launch's fabric-side bytes (443 -> 220 MB,
    // L2 hit rate 0.30 -> 0.63) and buys no time -- neutral alone, 0.8 % SLOWER in the step: the deep body is bound by
    // its LDS-DMA issue, not by where its operands come from.  VS_WGG_JOBS=1: on.
    static const int jobs_on = [] { const char* e = getenv("VS_WGG_JOBS"); return e ? atoi(e) : 0; }();
    bool fits = jobs_on != 0 && n <= 32;
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int x = 0; x < 8; ++x) {
      g.jobend[x] = 0;
      for (int i = 0; i < WGG_JOBS; ++i) g.jobtab[x][i] = ~0u;
    }
    long long pos = 0;  // position in the sequence of all blocks
    for (int a = 0; a < n && fits; ++a) {
      const int nt = g.p[a].tilesM * g.p[a].tilesN;
      if (nt > 4095 || g.p[a].S > 63) { fits = false; break; }
      for (int sp = 0; sp < g.p[a].S && fits; ++sp) {
        int t0 = 0;
        while (t0 < nt) {  // the piece of this job inside the current chunk
          const long long chunk = pos >> 5;
          const int room = 32 - (int)(pos & 31), c = nt - t0 < room ? nt - t0 : room;
          const int x = (int)(chunk & 7), slot0 = (int)(chunk >> 3) * 32 + (int)(pos & 31);
          if (cnt[x] == WGG_JOBS || slot0 + c > 510) { fits = false; break; }
          g.jobtab[x][cnt[x]++] = ((unsigned)slot0 << 23) | ((unsigned)a << 18) | ((unsigned)sp << 12) | (unsigned)t0;
          t0 += c;
          pos += c;
        }
      }
    }
    if (fits) {
      const long long chunks = (pos + 31) >> 5;
      int mx = 0;
      for (int x = 0; x < 8; ++x) {
        // slots of XCD x: its full chunks, and the sequence's tail if the last chunk is its own
        long long sl = 0;
        for (long long c = x; c < chunks; c += 8) sl = (c >> 3) * 32 + ((c == chunks - 1 && (pos & 31)) ? (pos & 31) : 32);
        g.jobend[x] = (unsigned short)sl;
        if ((int)sl > mx) mx = (int)sl;
      }
#ifdef VS_WGG_DEBUG
      for (int x = 0; x < 8; ++x) {
        fprintf(stderr, "xcd %d (%d slots):", x, (int)g.jobend[x]);
        for (int i = 0; i < cnt[x]; ++i) fprintf(stderr, " [slot %u p %u s %u t %u]", g.jobtab[x][i] >> 23, (g.jobtab[x][i] >> 18) & 31, (g.jobtab[x][i] >> 12) & 63, g.jobtab[x][i] & 4095);
        fprintf(stderr, "\n");
      }
      for (int a = 0; a < n; ++a) fprintf(stderr, "problem %d: tiles %d x %d S %d rps %d count %d\n", a, g.p[a].tilesM, g.p[a].tilesN, g.p[a].S, g.p[a].rows_per_split, g.count[a]);
#endif
      g.jobs = 1;
      grid = 8 * mx;
    }
  }
  static std::once_flag gattr;
  std::call_once(gattr, [] {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_deep_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const size_t smem = (size_t)WGD_NU * WGD_UNIT + 2 * WGD_TAB * sizeof(int2);
  hipLaunchKernelGGL(conv_wgrad_deep_group_kernel, dim3(grid), dim3(512), smem, (hipStream_t)stream, g);
  VS_CHECK_LAUNCH();
  if (r.n > 0) {
    hipLaunchKernelGGL(wgrad_reduce_group_kernel, dim3((unsigned)r.first[r.n]), dim3(256), 0, (hipStream_t)stream, r);
    VS_CHECK_LAUNCH();
  }
  return VS_OK;
}

static int wgrad_impl(const void* dy, const void* x, float* dw, const vs_conv_desc* d, void* workspace,
                      size_t ws_bytes, void* stream, bool reduce_now, int* splits_out,
                      const float* in_scale = nullptr, const float* in_shift = nullptr) {
  VS_CHECK_ARG(d && dy && x && dw, "null argument");
  VS_CHECK_ARG(d->Cin % 8 == 0 && d->Cout % 8 == 0, "Cin and Cout must be multiples of 8");
  VS_CHECK_ARG(d->x_ld % 8 == 0 && d->y_ld % 8 == 0, "row pitches must be multiples of 8");
  VS_CHECK_ARG((long long)d->N * d->Ti * d->Hi * d->Wi < (1ll << 31), "too many positions");
  const WgCfg c = wg_pick(d);
  const size_t need = vs_conv_wgrad_workspace_bytes(d);
  if (need > 0 && (workspace == nullptr || ws_bytes < need)) {
    vs_set_error("vs_conv_wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
    return VS_ERR_WORKSPACE;
  }
  if (g_pending.have && ((const void*)g_pending.r.slabs == workspace || g_pending.r.dw == dw)) pending_flush();
  if (in_scale && !wgrad_aol_plan_ok(d)) {
    vs_set_error("vs_conv_wgrad_aol: apply on load is not built for this weight gradient (ask vs_conv_wgrad_aol_ok)");
    return VS_ERR_UNSUPPORTED;
  }
  WgradP p;
  {
    const int rc = wg_fill_params(p, dy, x, c.S > 1 ? (float*)workspace : dw, d, c);
    if (rc) return rc;
  }
  p.in_scale = in_scale;
  p.in_shift = in_shift;
  {
    static const int xo = [] { const char* e = getenv("VS_WGRAD_XCD"); return e ? atoi(e) : 1; }();
    // measured per layer (batch 8): a win up to ~16 output tiles (s3.b 43.5 -> 35.4 us, s4.c 22.1 -> 19.7,
    // s3.c 24.7 -> 21.2, s2.c 29.7 -> 26.4), a loss from 36 tiles on (s4.b 36.7 -> 38.9, s5.b 48.9 -> 52.9), where
    // one split's tiles no longer fit an XCD's share of the grid anyway
    p.xcd_order = xo && (c.tilesM * c.tilesN <= 32 || xo == 2 || (c.S % 8 == 0 && c.tilesM * c.tilesN <= 64));
    static const int dbg = [] { const char* e = getenv("VS_WGRAD_DBG"); return e ? atoi(e) : 0; }();
    p.dbg = dbg;
  }
  p.rows_per_split = c.rows_per_split;
  const bool dense = (d->kT * d->kH * d->kW == 1) && d->sT == 1 && d->sH == 1 && d->sW == 1 &&
                     d->pT == 0 && d->pH == 0 && d->pW == 0;
  const int mode = dense ? 0 : 1;
  hipStream_t st = (hipStream_t)stream;
  // staging: VS_CONV_RING(1) register pipeline, (2|3) LDS-DMA ring stages, 0 = heuristic
  const int fring = (d->flags >> 16) & 7;
  // heuristic: the 2-stage ring wins 3-8 % on the 128-row tiles (profiles/r01_wgrad_ring.txt) and
  // loses a little on the skinny ones
  int ring = fring == 1 ? 0 : (fring >= 2 ? (fring > 3 ? 3 : fring) : (c.bm == 128 ? 2 : 0));
  if (d->kT * d->kH * d->kW > 31) ring = 0;  // the tap bitmask of the ring's position table
  int rc;
  if (c.deep) {
    static std::once_flag dattr;
    std::call_once(dattr, [] {
      (void)hipFuncSetAttribute((const void*)conv_wgrad_deep_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_wgrad_deep_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_wgrad_deep_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_wgrad_deep_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    static const int stag = [] { const char* e = getenv("VS_WGRAD_DEEP_STAG"); return e ? atoi(e) : 1; }();  // +2.7 % at 32 clips, neutral at 8
    if (in_scale) {
      vs_set_error("vs_conv_wgrad_aol: not built for the deep-pipeline plan");
      return VS_ERR_UNSUPPORTED;
    }
    const size_t smem = (size_t)WGD_NU * WGD_UNIT + 2 * WGD_TAB * sizeof(int2);
    const int grid = p.tilesM * p.tilesN * p.S;
    if (stag && mode == 0) hipLaunchKernelGGL((conv_wgrad_deep_kernel<0, true>), dim3(grid), dim3(512), smem, st, p);
    else if (stag) hipLaunchKernelGGL((conv_wgrad_deep_kernel<1, true>), dim3(grid), dim3(512), smem, st, p);
    else if (mode == 0) hipLaunchKernelGGL((conv_wgrad_deep_kernel<0, false>), dim3(grid), dim3(512), smem, st, p);
    else hipLaunchKernelGGL((conv_wgrad_deep_kernel<1, false>), dim3(grid), dim3(512), smem, st, p);
    VS_CHECK_LAUNCH();
    rc = VS_OK;
  } else if (c.bm == 128 && c.bn == 128) rc = wg_launch<128, 128, 2, 2>(p, mode, ring, st);
  else if (c.bm == 128) rc = wg_launch<128, 64, 2, 2>(p, mode, ring, st);
  else if (c.bm == 64 && c.bn == 128) rc = wg_launch<64, 128, 2, 2>(p, mode, ring, st);
  else if (c.bm == 64) rc = wg_launch<64, 64, 2, 2>(p, mode, ring, st);
  else if (c.bm == 32 && c.bn == 128) rc = wg_launch<32, 128, 1, 4>(p, mode, ring, st);
  else if (c.bm == 32) rc = wg_launch<32, 64, 1, 4>(p, mode, ring, st);
  else if (c.bn == 128) rc = wg_launch<16, 128, 1, 4>(p, mode, ring, st);
  else rc = wg_launch<16, 64, 1, 4>(p, mode, ring, st);
  if (rc) return rc;
  if (splits_out) *splits_out = c.S;
  if (c.S > 1 && reduce_now) {
    const long long n = (long long)d->Cout * p.Kp;
    const long long grid = wgrad_reduce_vblocks(n, c.S);
    if (!pair_defer_reduce((const float*)workspace, dw, n, c.S) &&
        !pending_stash((const float*)workspace, dw, n, c.S, st)) {
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)grid), dim3(256), 0, st,
                         (const float*)workspace, dw, n, c.S);
      VS_CHECK_LAUNCH();
    }
  }
  return VS_OK;
}

extern "C" int vs_conv_wgrad(const void* dy, const void* x, float* dw, const vs_conv_desc* d,
                             void* workspace,
                             size_t ws_bytes, void* stream) {
  return wgrad_impl(dy, x, dw, d, workspace, ws_bytes, stream, true, nullptr);
}

// dW of a 1x1x1 convolution whose input is relu(x * in_scale[c] + in_shift[c]) with x the producer unit's raw output
// (see vs_conv_fwd_aol): the operand vs_bn_apply would have stored, formed on the fragments.  _ok: 1 where built.
extern "C" int vs_conv_wgrad_aol_ok(const vs_conv_desc* d) { return (d && wgrad_aol_plan_ok(d)) ? 1 : 0; }

extern "C" int vs_conv_wgrad_aol(const void* dy, const void* x, float* dw, const vs_conv_desc* d, const float* in_scale,
                                 const float* in_shift, void* workspace, size_t ws_bytes, void* stream) {
  VS_CHECK_ARG(in_scale && in_shift, "null constants");
  return wgrad_impl(dy, x, dw, d, workspace, ws_bytes, stream, true, nullptr, in_scale, in_shift);
}

extern "C" int vs_conv_wgrad_partial(const void* dy, const void* x, float* dw, const vs_conv_desc* d, void* slabs,
                                     size_t slab_bytes, int* splits, void* stream) {
  VS_CHECK_ARG(splits != nullptr, "null splits");
  return wgrad_impl(dy, x, dw, d, slabs, slab_bytes, stream, false, splits);
}
