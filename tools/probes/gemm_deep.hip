// Probe: the main loop planned for the MFMA-side convolutions, as a plain bf16 GEMM  C[M][N] = A[M][K] * B[N][K]^T.
//
// 512 threads = 8 waves as 2 (M) x 4 (N), block tile 256 x 256 x 64, wave tile 128 x 64 (acc 8 x 4 fragments),
// v_mfma_f32_16x16x32_bf16.  LDS = 2 stages x 64 KiB, each stage cut into four 16-KiB sub-buffers
//   At / Ab : the top / bottom 64 rows of every wave row's 128       (128 rows x 128 B)
//   Bl / Br : the left / right 32 columns of every wave column's 64  (128 rows x 128 B)
// A k-tile is four phases of 16 MFMAs per wave: At x Bl, At x Br, Ab x Br, Ab x Bl.  Every phase
//   s_waitcnt vmcnt(12); s_barrier; issue ONE sub-buffer (2 LDS-DMA per wave) 7 phases ahead of its first read;
//   read the operand fragments the NEXT phase changes (4 or 8 ds_read_b128) ; 16 MFMAs on fragments already in
//   registers; s_waitcnt lgkmcnt(0)
// so 7 of the 8 sub-buffers (112 KiB) are in flight or waiting, each read exactly once, and a sub-buffer is refilled
// in the phase after its only read.  Read order r = 0, 1, 2, ...: At(0), Bl(0), Br(0), Ab(0), At(1), ...; sub-buffer
// r is read in phase r - 2 and issued in phase r - 9.
// build: hipcc -O3 --offload-arch=gfx950 gemm_deep.hip -o gemm_deep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#include <type_traits>
#include <cstring>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define OOB 0x80000000u

struct P {
  const uint16_t* A;
  const uint16_t* B;
  uint16_t* C;
  int M, N, K;
  unsigned a_bytes, b_bytes;
  int tilesN;
  int prio;
  unsigned long long* dbg;
};

__device__ __forceinline__ uint16_t f2bf(float f) {
  unsigned u = __builtin_bit_cast(unsigned, f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// AF = A fragments per half of a wave's rows: 4 -> 256-row block tile (wave tile 128 x 64), 2 -> 128-row block tile
// (wave tile 64 x 64).  The waves of half hf = wv >> 2 (a SIMD hosts one wave of each half) issue the copies of
// sub-buffers At / Br (half 0) and Bl / Ab (half 1), in a burst when the phase opens: every phase has one loading and
// one purely multiplying wave per SIMD ("halves burst", the fastest of the four issue orders tried -- v2 of this probe).
// STAG (AF = 4 only): every wave copies 2 of a sub-buffer's 16 pieces; waves 0-3 issue theirs when the phase opens, waves 4-7
// after their MFMAs (the two waves of a SIMD run half a phase apart).
// STAMP (diagnostic build; +3 s_memtime per phase): per wave, summed over the main loop, the cycles of a phase's three
// segments -- [vmcnt wait + barrier] [copy issue] [fragment reads + 16 MFMAs + lgkmcnt(0)] -- separately for the phases in
// which the wave is the loading one and those in which it only multiplies; written to p.dbg[block][wave][8].
template <int AF, bool STAG = false, bool STAMP = false>
__global__ __launch_bounds__(512) void gemm_deep(P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 64 * AF;
  constexpr unsigned ASUB = AF * 4096u;               // bytes of an A sub-buffer (2 wave rows x AF x 16 rows x 128 B)
  constexpr unsigned SUB_AT = 0, SUB_AB = ASUB, SUB_BL = 2 * ASUB, SUB_BR = 2 * ASUB + 16384u;
  constexpr unsigned STAGE = 2 * ASUB + 32768u;
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  int swz;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tn = swz % p.tilesN, tm = swz / p.tilesN;
  const int m0 = tm * BM, n0 = tn * 256;
  const int nk = (p.K + 63) >> 6;

  const int hf = wv >> 2, wq = wv & 3;
  // copies: instruction i of this wave's share of a sub-buffer covers rows sr = (i * 4 + wq) * 8 + (lane >> 3)
  unsigned srcA[AF], srcB[4];  // byte offsets at k = 0 of this half's A sub-buffer (At or Ab) and B sub-buffer (Br or Bl)
  {
    const int kc = lane & 7, r8 = lane >> 3;
#pragma unroll
    for (int i = 0; i < AF; ++i) {
      const int sr = (i * 4 + wq) * 8 + r8;  // 0 .. 32 AF - 1: wave row sr / (16 AF), row sr % (16 AF) of the half
      const unsigned unit = (unsigned)((kc ^ ((sr >> 1) & 7)) << 4);
      const int arow = m0 + (sr / (16 * AF)) * (32 * AF) + (hf == 0 ? 0 : 16 * AF) + (sr % (16 * AF));
      srcA[i] = arow < p.M ? (unsigned)((long long)arow * p.K * 2) + unit : OOB;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sr = (i * 4 + wq) * 8 + r8;
      const unsigned unit = (unsigned)((kc ^ ((sr >> 1) & 7)) << 4);
      const int brow = n0 + (sr >> 5) * 64 + (hf == 0 ? 32 : 0) + (sr & 31);  // half 0: Br, half 1: Bl
      srcB[i] = brow < p.N ? (unsigned)((long long)brow * p.K * 2) + unit : OOB;
    }
  }
  unsigned sA[2][2], sB[2][2];  // STAG: [half of the tile][i], rows sr = (i * 8 + wv) * 8 + (lane >> 3)
  if (STAG) {
    const int kc = lane & 7, r8 = lane >> 3;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int sr = (i * 8 + wv) * 8 + r8;
        const unsigned unit = (unsigned)((kc ^ ((sr >> 1) & 7)) << 4);
        const int arow = m0 + (sr >> 6) * 128 + h * 64 + (sr & 63);
        sA[h][i] = arow < p.M ? (unsigned)((long long)arow * p.K * 2) + unit : OOB;
        const int brow = n0 + (sr >> 5) * 64 + h * 32 + (sr & 31);
        sB[h][i] = brow < p.N ? (unsigned)((long long)brow * p.K * 2) + unit : OOB;
      }
  }
  typedef __attribute__((address_space(3))) char* lds_ptr_t;
  const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)smem + (unsigned)(STAG ? wv : wq) * 1024u;
  auto rsrc_words = [](const void* base, unsigned bytes) __attribute__((always_inline)) {
    const unsigned long a = (unsigned long)base;
    return (i32x4){(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
  };
  const i32x4 adesc = rsrc_words(p.A, p.a_bytes), bdesc = rsrc_words(p.B, p.b_bytes);
  auto dma16 = [](const i32x4& desc, unsigned lds_addr, unsigned voff) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(desc)
                 : "memory");
  };
  // sub-buffer of read index r: kind = r & 3 (0 At, 1 Bl, 2 Br, 3 Ab), tile = r >> 2, stage = tile & 1; kinds 0 / 2 are
  // half 0's, kinds 1 / 3 half 1's
  auto issue = [&](int kind, int tile) __attribute__((always_inline)) {
    if (STAG) {
      const unsigned st = (unsigned)(tile & 1) * STAGE;
      const unsigned kadd = tile < nk ? (unsigned)tile * 128u : OOB;
      const bool isA = kind == 0 || kind == 3;
      const int h = (kind == 0 || kind == 1) ? 0 : 1;
      const unsigned sub = kind == 0 ? SUB_AT : kind == 1 ? SUB_BL : kind == 2 ? SUB_BR : SUB_AB;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned s0 = isA ? sA[h][i] : sB[h][i];
        dma16(isA ? adesc : bdesc, lds0 + st + sub + (unsigned)i * 8192u, (s0 | kadd) >= OOB ? OOB : s0 + kadd);
      }
      return;
    }
    if ((kind & 1) != hf) return;
    const unsigned st = (unsigned)(tile & 1) * STAGE;
    const unsigned kadd = tile < nk ? (unsigned)tile * 128u : OOB;  // past the last tile: zeros nobody reads
    if (kind == 0 || kind == 3) {
#pragma unroll
      for (int i = 0; i < AF; ++i)
        dma16(adesc, lds0 + st + (kind == 0 ? SUB_AT : SUB_AB) + (unsigned)i * 4096u,
              (srcA[i] | kadd) >= OOB ? OOB : srcA[i] + kadd);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        dma16(bdesc, lds0 + st + (kind == 1 ? SUB_BL : SUB_BR) + (unsigned)i * 4096u,
              (srcB[i] | kadd) >= OOB ? OOB : srcB[i] + kadd);
    }
  };
  // before the read of sub-buffer kind kr (issued 7 phases ago by half kr & 1): that half's copies issued since --
  // kinds kr+1 .. kr+6 -- may stay in flight.  Per wave an A sub-buffer is AF instructions, a B sub-buffer 4.
  auto wait_for = [&](auto kr_) __attribute__((always_inline)) {
    constexpr int kr = decltype(kr_)::value;
    constexpr int nA = (kr == 2) ? 2 : (kr == 3) ? 1 : (kr == 0) ? 1 : 2;  // A sub-buffers among the 3 younger ones of this half
    constexpr int N = nA * AF + (3 - nA) * 4;
    if (STAG) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // six younger sub-buffers x 2 pieces per wave
    else if (hf == (kr & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
  };

  // ---- compute side
  const int wm = wv >> 2, wn = wv & 3;
  const int lr = lane & 15, lq = lane >> 4;
  const unsigned lpart = (unsigned)(lr * 128 + ((lq ^ (lr >> 1)) << 4));  // ks = 0; ks = 1: ^ 64
  const char* aBase[2] = {smem + wm * (AF * 2048) + lpart, smem + wm * (AF * 2048) + (lpart ^ 64u)};  // + stage + sub + a * 2048
  const char* bBase[2] = {smem + wn * 4096 + lpart, smem + wn * 4096 + (lpart ^ 64u)};                // + stage + sub + b * 2048

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
      for (int ks = 0; ks < 2; ++ks)
        dst[a][ks] = *(const bf16x8*)(aBase[ks] + off + a * 2048);
  };
  auto readB = [&](bf16x8 (&dst)[2][2], unsigned off) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        dst[b][ks] = *(const bf16x8*)(bBase[ks] + off + b * 2048);
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
  unsigned long long tsum[2][3] = {{0, 0, 0}, {0, 0, 0}}, tcnt[2] = {0, 0}, tprev = 0, tbar = 0;
  int trole = 0;
  auto phase_open = [&](auto kr_) __attribute__((always_inline)) {
    if (STAMP) tprev = __builtin_readcyclecounter();
    wait_for(kr_);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (STAMP) {
      tbar = __builtin_readcyclecounter();
      // the sub-buffer issued in this phase is of kind kr + 3 (mod 4): its half is the loading one
      trole = ((decltype(kr_)::value + 3) & 1) == hf ? 1 : 0;
      tsum[trole][0] += tbar - tprev;
    }
  };
  auto issued = [&]() __attribute__((always_inline)) {  // call right after the phase's issue()
    if (STAMP) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_readcyclecounter();
      tsum[trole][1] += t - tbar;
      tbar = t;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto phase_close = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (STAMP) {
      tsum[trole][2] += __builtin_readcyclecounter() - tbar;
      tcnt[trole] += 1;
    }
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;

  // prologue: r = 0 .. 6
  issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0); issue(0, 1); issue(1, 1); issue(2, 1);
  // phase -2: issue r = 7 (Ab 1), read At(0)
  phase_open(K0{});
  issue(3, 1);
  readA(A0, 0 + SUB_AT);
  phase_close();
  // phase -1: issue r = 8 (At 2), read Bl(0)
  phase_open(K1{});
  issue(0, 2);
  readB(B0, 0 + SUB_BL);
  phase_close();

  // one k-tile; PAR = tile parity (stage and B register-set roles)
  auto tile_body = [&](auto par, int t) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par)::value;
    constexpr unsigned ST = PAR * STAGE, STN = (PAR ^ 1) * STAGE;
    bf16x8(&BL)[2][2] = PAR ? B1 : B0;
    bf16x8(&BR)[2][2] = PAR ? B0 : B1;
    // P1: issue Bl(t+2), read Br(t), At x Bl
    phase_open(K2{});
    if (!STAG || hf == 0) issue(1, t + 2);
    issued();
    readB(BR, ST + SUB_BR);
    mma(A0, BL, 0, 0);
    if (STAG && hf == 1) { __builtin_amdgcn_sched_barrier(0); issue(1, t + 2); }
    phase_close();
    // P2: issue Br(t+2), read Ab(t), At x Br
    phase_open(K3{});
    if (!STAG || hf == 0) issue(2, t + 2);
    issued();
    readA(A1, ST + SUB_AB);
    mma(A0, BR, 0, 2);
    if (STAG && hf == 1) { __builtin_amdgcn_sched_barrier(0); issue(2, t + 2); }
    phase_close();
    // P3: issue Ab(t+2), read At(t+1), Ab x Br
    phase_open(K0{});
    if (!STAG || hf == 0) issue(3, t + 2);
    issued();
    readA(A0, STN + SUB_AT);
    mma(A1, BR, AF, 2);
    if (STAG && hf == 1) { __builtin_amdgcn_sched_barrier(0); issue(3, t + 2); }
    phase_close();
    // P4: issue At(t+3), read Bl(t+1) into the set Br(t) leaves, Ab x Bl
    phase_open(K1{});
    if (!STAG || hf == 0) issue(0, t + 3);
    issued();
    readB(BR, STN + SUB_BL);
    mma(A1, BL, AF, 0);
    if (STAG && hf == 1) { __builtin_amdgcn_sched_barrier(0); issue(0, t + 3); }
    phase_close();
  };
  int t = 0;
  for (; t + 1 < nk; t += 2) {
    tile_body(std::integral_constant<int, 0>{}, t);
    tile_body(std::integral_constant<int, 1>{}, t + 1);
  }
  if (t < nk) tile_body(std::integral_constant<int, 0>{}, t);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (STAMP && lane == 0) {
    unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wv) * 8;
    d[0] = tsum[0][0]; d[1] = tsum[0][1]; d[2] = tsum[0][2]; d[3] = tcnt[0];
    d[4] = tsum[1][0]; d[5] = tsum[1][1]; d[6] = tsum[1][2]; d[7] = tcnt[1];
  }
  if (p.prio == 99) return;  // (ablation from the host: main loop only; wrong results)

  // epilogue: bf16 tile through LDS (row pitch 256 + 8 elements), 16-byte stores
  constexpr int EP = 256 + 8;
  uint16_t* E = (uint16_t*)smem;
#pragma unroll
  for (int a = 0; a < 2 * AF; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int half = a / AF, aa = a % AF, hb = b >> 1, bb = b & 1;
      const int row0 = wm * (32 * AF) + half * (16 * AF) + aa * 16 + lq * 4;
      const int col = wn * 64 + hb * 32 + bb * 16 + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) E[(row0 + r) * EP + col] = f2bf(acc[a][b][r]);
    }
  __syncthreads();
  for (int idx = tid; idx < BM * 32; idx += 512) {
    const int row = idx >> 5, c8 = idx & 31;
    if (m0 + row < p.M && n0 + c8 * 8 < p.N)
      *(uint4*)(p.C + (long long)(m0 + row) * p.N + n0 + c8 * 8) = *(const uint4*)(E + row * EP + c8 * 8);
  }
}

static uint16_t h_f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float h_bf2f(uint16_t h) {
  unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

template <int AF, bool STAG = false>
static float run(const P& p, int grid, int reps) {
  hipFuncSetAttribute((const void*)gemm_deep<AF, STAG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t ring = 2 * (2 * AF * 4096 + 32768), epi = (size_t)64 * AF * (256 + 8) * 2;
  const size_t smem = ring > epi ? ring : epi;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_deep<AF, STAG>), dim3(grid), dim3(512), smem, 0, p);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_deep<AF, STAG>), dim3(grid), dim3(512), smem, 0, p);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms / reps < best) best = ms / reps;
  }
  return best;
}

int main(int argc, char** argv) {
  struct Shape { int M, N, K; const char* what; };
  const Shape shapes[] = {
      {4096, 4096, 4096, "4096^3 (guide's template shape)"},
      {50176, 256, 3072, "s4.a at 32 clips"},
      {12544, 256, 3072, "s4.a at 8 clips"},
      {25088, 256, 1536, "s4.a at 8 clips, K split 2 (emulated)"},
      {12544, 256, 2304, "s4.b at 8 clips (as a dense GEMM)"},
      {25088, 256, 1152, "s4.b at 8 clips, K split 2 (emulated)"},
      {50176, 256, 1920, "s4.b0.a at 8 clips"},
      {12544, 1024, 768, "s4.a dgrad at 8 clips"},
      {12544, 1024, 640, "s4 shortcut at 8 clips"},
      {3136, 512, 6144, "s5.a at 8 clips"},
      {12544, 512, 1536, "s5.a at 8 clips, K split 4 (emulated)"},
      {12544, 512, 3840, "s5.b0.a at 8 clips"},
      {3136, 2048, 1280, "s5 shortcut at 8 clips"},
  };
  for (const Shape& s : shapes) {
    const size_t na = (size_t)s.M * s.K, nb = (size_t)s.N * s.K, nc = (size_t)s.M * s.N;
    std::vector<uint16_t> ha(na), hb(nb), hc(nc);
    unsigned seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : ha) v = h_f2bf(rnd());
    for (auto& v : hb) v = h_f2bf(rnd() * 0.05f);
    uint16_t *dA, *dB, *dC;
    hipMalloc(&dA, na * 2); hipMalloc(&dB, nb * 2); hipMalloc(&dC, nc * 2);
    hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, hb.data(), nb * 2, hipMemcpyHostToDevice);
    hipMemset(dC, 0, nc * 2);
    P p;
    p.A = dA; p.B = dB; p.C = dC; p.M = s.M; p.N = s.N; p.K = s.K;
    p.a_bytes = (unsigned)(na * 2); p.b_bytes = (unsigned)(nb * 2);
    p.tilesN = (s.N + 255) / 256;
    p.dbg = nullptr;
    const double fl = 2.0 * s.M * s.N * s.K;
    const int reps = fl > 5e11 ? 5 : 20;
    const int g4 = ((s.M + 255) / 256) * p.tilesN, g2 = ((s.M + 127) / 128) * p.tilesN;
    p.prio = 0;
    const float t4 = run<4>(p, g4, reps), t2 = run<2>(p, g2, reps), ts = run<4, true>(p, g4, reps);
    p.prio = 99;
    const float l4 = run<4>(p, g4, reps), l2 = run<2>(p, g2, reps), ls = run<4, true>(p, g4, reps);
    printf("%-40s M%6d N%5d K%5d | 256x256: %4d blocks %7.1f us %5.0f TF/s (loop only %7.1f) | staggered all-wave issue %7.1f us %5.0f TF/s (loop only %7.1f) | 128x256: %4d blocks %7.1f us %5.0f TF/s (loop only %7.1f)\n",
           s.what, s.M, s.N, s.K, g4, t4 * 1e3, fl / t4 / 1e9, l4 * 1e3, ts * 1e3, fl / ts / 1e9, ls * 1e3, g2, t2 * 1e3, fl / t2 / 1e9, l2 * 1e3);
    if (s.M == 4096 || (s.M == 50176 && s.K == 3072)) {  // cycle budget of a phase (diagnostic build)
      unsigned long long* dbg;
      hipMalloc(&dbg, (size_t)g4 * 8 * 8 * 8);
      hipMemset(dbg, 0, (size_t)g4 * 8 * 8 * 8);
      p.dbg = dbg;
      p.prio = 0;
      hipFuncSetAttribute((const void*)gemm_deep<4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      const size_t smem = 256 * (256 + 8) * 2;
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_deep<4, false, true>), dim3(g4), dim3(512), smem, 0, p);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h((size_t)g4 * 64);
      hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
      for (int half = 0; half < 2; ++half) {
        double a[2][3] = {{0, 0, 0}, {0, 0, 0}}, c[2] = {0, 0};
        for (int b = 0; b < g4; ++b)
          for (int w = half * 4; w < half * 4 + 4; ++w) {
            const unsigned long long* d = &h[((size_t)b * 8 + w) * 8];
            for (int r = 0; r < 2; ++r) {
              for (int k = 0; k < 3; ++k) a[r][k] += (double)d[r * 4 + k];
              c[r] += (double)d[r * 4 + 3];
            }
          }
        printf("   stamps, waves %d-%d: multiplying phases: wait+barrier %6.0f, (no copies) %4.0f, reads+MFMA+lgkm %6.0f | loading phases: wait+barrier %6.0f, copy issue %6.0f, reads+MFMA+lgkm %6.0f   [cycles per phase, mean over %0.f phases]\n",
               half * 4, half * 4 + 3, a[0][0] / c[0], a[0][1] / c[0], a[0][2] / c[0], a[1][0] / c[1], a[1][1] / c[1], a[1][2] / c[1], c[0] + c[1]);
      }
      hipFree(dbg);
      p.dbg = nullptr;
    }
    // check sampled outputs against a host dot product (the full-epilogue variant ran last? no: rerun it)
    p.prio = 0;
    run<4, true>(p, g4, 1);
    hipMemcpy(hc.data(), dC, nc * 2, hipMemcpyDeviceToHost);
    double maxerr = 0.0;
    int bad = 0;
    for (int i = 0; i < 4000; ++i) {
      seed = seed * 1664525u + 1013904223u;
      const int m = (int)((seed >> 4) % (unsigned)s.M);
      seed = seed * 1664525u + 1013904223u;
      const int n = (int)((seed >> 4) % (unsigned)s.N);
      double ref = 0.0;
      for (int k = 0; k < s.K; ++k) ref += (double)h_bf2f(ha[(size_t)m * s.K + k]) * h_bf2f(hb[(size_t)n * s.K + k]);
      const double got = h_bf2f(hc[(size_t)m * s.N + n]);
      const double err = fabs(got - ref);
      if (err > maxerr) maxerr = err;
      if (err > 0.02 + 0.01 * fabs(ref)) ++bad;
    }
    printf("   check: 4000 samples, max abs err %.4f, bad %d\n", maxerr, bad);
    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  return 0;
}
