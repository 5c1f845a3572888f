// One launch for a unit's data gradient AND weight gradient.
//
// In the backward pass of a conv + BN unit the two gradients are independent; the trunk ran the weight gradient on
// a side stream beside the data gradient.  Inside a replayed hipGraph that fork + join costs ~17 us per unit on this
// system (tools/graph_edge_cost.py, profiles/r02_graph_edge_cost.txt) -- about what the overlap saves.  Here both
// kernels' blocks go into ONE grid: blocks [0, gd) run the tile kernel's body on the dgrad's ConvP, blocks
// [gd, gd + gw) the ring weight-gradient body on the WgradP; the dispatcher fills the chip with dgrad tiles first
// and weight-gradient blocks take the slots as they free up.  Same bodies, same per-convolution block indices:
// bitwise the two launches.
//
// Mechanism: between vs_conv_pair_begin() and vs_conv_pair_end() the launch functions of the two families RECORD an
// eligible launch (128 x 128 tile, two-stage LDS-DMA ring, no split-K / second BN unit) instead of issuing it;
// _end issues the pair as one launch (or whatever was recorded, alone), then the weight gradient's slab reduce.
// Everything else launched in between is issued as usual.  The two .hip files are compiled as part of this
// translation unit (the pair kernel needs both bodies).
#define VS_CONV_PAIR_TU 1
#include <atomic>
#include <mutex>

#include "conv_igemm.hip"
#include "conv_wgrad.hip"

namespace {
struct PairState {
  bool active = false;
  bool have_d = false, have_w = false, have_r = false;
  ConvP dp;
  WgradP wp;
  int d_grid = 0, w_grid = 0, d_mode = 0, w_mode = 0;
  bool d_bnb = false;
  size_t d_smem = 0, w_smem = 0;
  const float* slabs = nullptr;
  float* dw = nullptr;
  long long n = 0;
  int splits = 0;
  hipStream_t st = nullptr;
};
thread_local PairState g_pair;
std::atomic<long long> g_pairs_issued{0};  // launches that held both kernels (tests); any thread may issue one
}  // namespace

static bool pair_take_dgrad(const ConvP& p, int grid, size_t smem, int mode, bool bnb, hipStream_t st) {
  PairState& s = g_pair;
  if (!s.active || s.have_d || p.splitK != 1) return false;
  if (s.have_w && s.st != st) return false;  // one launch = one stream
  s.dp = p; s.d_grid = grid; s.d_smem = smem; s.d_mode = mode; s.d_bnb = bnb; s.st = st;
  s.have_d = true;
  return true;
}

static bool pair_take_wgrad(const WgradP& p, int grid, size_t smem, int mode, hipStream_t st) {
  PairState& s = g_pair;
  if (!s.active || s.have_w) return false;
  if (s.have_d && s.st != st) return false;
  s.wp = p; s.w_grid = grid; s.w_smem = smem; s.w_mode = mode; s.st = st;
  s.have_w = true;
  return true;
}

static bool pair_defer_reduce(const float* slabs, float* dw, long long n, int splits) {
  PairState& s = g_pair;
  if (!s.active || !s.have_w || s.have_r) return false;
  if ((const float*)s.wp.out != slabs) return false;  // only the reduce of the RECORDED weight gradient's slabs waits for _end
  s.slabs = slabs; s.dw = dw; s.n = n; s.splits = splits;
  s.have_r = true;
  return true;
}

// WAOL: the weight gradient's x operand goes through its producer's BN + ReLU on load (WgradP::in_scale; WMODE 0)
template <int DMODE, bool BNB, int WMODE, bool WAOL = false>
__global__ __launch_bounds__(256) void conv_pair_kernel(ConvP dp, WgradP wp, int gd) {
  if ((int)blockIdx.x < gd)
    conv_igemm_body<128, 128, 2, 2, DMODE, true, 0, 2, BNB, false>(dp, blockIdx.x, gd);
  else
    conv_wgrad_ring_body<128, 128, 2, 2, WMODE, 2, WAOL>(wp, blockIdx.x - gd, gridDim.x - gd);
}

template <int DMODE, bool BNB, int WMODE, bool WAOL = false>
static void pair_launch(const PairState& s) {
  static std::once_flag attr;  // the autograd thread and the main thread may both get here first
  std::call_once(attr, [] {
    (void)hipFuncSetAttribute((const void*)conv_pair_kernel<DMODE, BNB, WMODE, WAOL>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const size_t smem = s.d_smem > s.w_smem ? s.d_smem : s.w_smem;
  hipLaunchKernelGGL((conv_pair_kernel<DMODE, BNB, WMODE, WAOL>), dim3(s.d_grid + s.w_grid), dim3(256), smem, s.st,
                     s.dp, s.wp, s.d_grid);
}

template <int DMODE, bool BNB>
static void dgrad_alone(const PairState& s) {
  // a dgrad recorded inside a pair block never reached launch_one's own opt-in: > 64 KiB of dynamic LDS needs it
  static std::once_flag attr;
  std::call_once(attr, [] {
    (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<128, 128, 2, 2, DMODE, true, 0, 2, BNB, false>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 2, DMODE, true, 0, 2, BNB, false>), dim3(s.d_grid), dim3(256),
                     s.d_smem, s.st, s.dp);
}

extern "C" int vs_conv_pair_begin(void) {
  g_pair = PairState();
  g_pair.active = true;
  return VS_OK;
}

extern "C" int vs_conv_pair_end(void) {
  PairState s = g_pair;
  g_pair = PairState();
  if (s.have_d && s.have_w && s.wp.in_scale) {  // (recorded only for pointwise weight gradients: w_mode == 0)
    switch (s.d_mode * 2 + (s.d_bnb ? 1 : 0)) {
      case 0: pair_launch<0, false, 0, true>(s); break;  case 1: pair_launch<0, true, 0, true>(s); break;
      case 2: pair_launch<1, false, 0, true>(s); break;  case 3: pair_launch<1, true, 0, true>(s); break;
      case 4: pair_launch<2, false, 0, true>(s); break;  default: pair_launch<2, true, 0, true>(s); break;
    }
  } else if (s.have_d && s.have_w) {
#define VS_PAIR(DM, B, WM_) pair_launch<DM, B, WM_>(s)
    const int key = s.d_mode * 4 + (s.d_bnb ? 2 : 0) + s.w_mode;
    switch (key) {
      case 0: VS_PAIR(0, false, 0); break;   case 1: VS_PAIR(0, false, 1); break;
      case 2: VS_PAIR(0, true, 0); break;    case 3: VS_PAIR(0, true, 1); break;
      case 4: VS_PAIR(1, false, 0); break;   case 5: VS_PAIR(1, false, 1); break;
      case 6: VS_PAIR(1, true, 0); break;    case 7: VS_PAIR(1, true, 1); break;
      case 8: VS_PAIR(2, false, 0); break;   case 9: VS_PAIR(2, false, 1); break;
      case 10: VS_PAIR(2, true, 0); break;   default: VS_PAIR(2, true, 1); break;
    }
#undef VS_PAIR
  } else {
    if (s.have_d) {
      switch (s.d_mode * 2 + (s.d_bnb ? 1 : 0)) {
        case 0: dgrad_alone<0, false>(s); break;  case 1: dgrad_alone<0, true>(s); break;
        case 2: dgrad_alone<1, false>(s); break;  case 3: dgrad_alone<1, true>(s); break;
        case 4: dgrad_alone<2, false>(s); break;  default: dgrad_alone<2, true>(s); break;
      }
    }
    if (s.have_w) {
      if (s.wp.in_scale) {
        static std::once_flag aattr;
        std::call_once(aattr, [] {
          (void)hipFuncSetAttribute((const void*)conv_wgrad_ring_kernel<128, 128, 2, 2, 0, 2, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
        hipLaunchKernelGGL((conv_wgrad_ring_kernel<128, 128, 2, 2, 0, 2, true>), dim3(s.w_grid), dim3(256), s.w_smem,
                           s.st, s.wp);
      } else if (s.w_mode == 0)
        hipLaunchKernelGGL((conv_wgrad_ring_kernel<128, 128, 2, 2, 0, 2>), dim3(s.w_grid), dim3(256), s.w_smem, s.st, s.wp);
      else
        hipLaunchKernelGGL((conv_wgrad_ring_kernel<128, 128, 2, 2, 1, 2>), dim3(s.w_grid), dim3(256), s.w_smem, s.st, s.wp);
    }
  }
  if (s.have_r && !pending_stash(s.slabs, s.dw, s.n, s.splits, s.st)) {
    const long long grid = wgrad_reduce_vblocks(s.n, s.splits);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)grid), dim3(256), 0, s.st, s.slabs, s.dw, s.n, s.splits);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    vs_set_error("vs_conv_pair_end: launch failed: %s", hipGetErrorString(e));
    return VS_ERR_LAUNCH;
  }
  if (s.have_d && s.have_w) g_pairs_issued.fetch_add(1, std::memory_order_relaxed);
  return VS_OK;
}

extern "C" int64_t vs_conv_pair_count(void) { return g_pairs_issued.load(std::memory_order_relaxed); }
