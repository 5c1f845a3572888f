// Frame resize of the loader on the GPU (first half): PIL's
// `img.resize((224, 224))` of `VsituDS.read_img` (vidsitu_code/dat_loader.py:183-191), i.e. Pillow's
// two-pass bicubic resampling of 8-bit RGB (Resample.c: pillow=7.2.0, vsitu_pyt_env.yml:151), bit for bit:
// the 22-bit fixed-point weights are computed on the host in double precision exactly as Pillow does
// (vs_resize_coeffs), the device does the integer part -- per output pixel a window of source pixels
// times int weights, accumulated from 1 << 21, shifted, clipped to 0..255 after EACH pass.
#include "common.h"
#include <math.h>

#define RS_PRECISION_BITS (32 - 8 - 2)

static inline double rs_bicubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

extern "C" int vs_resize_ksize(int in_size, int out_size) {
  if (in_size <= 0 || out_size <= 0) return 0;
  double filterscale = (double)in_size / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  return (int)ceil(2.0 * filterscale) * 2 + 1;
}

// HOST tables: bounds[out][2] = (first source index, count), kk[out][ksize] fixed-point weights
extern "C" int vs_resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk) {
  VS_CHECK_ARG(in_size > 0 && out_size > 0 && bounds && kk, "bad args");
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double w[64 * 2 + 8];
    VS_CHECK_ARG(xmax <= (int)(sizeof(w) / sizeof(w[0])) && xmax <= ksize, "down-scaling factor too large");
    volatile double ww = 0.0;  // (volatile: keep the plain double sum, no reassociation)
    for (int x = 0; x < xmax; ++x) {
      w[x] = rs_bicubic((x + xmin - center + 0.5) * ss);
      ww = ww + w[x];
    }
    for (int x = 0; x < ksize; ++x) {
      double v = 0.0;
      if (x < xmax) v = ww != 0.0 ? w[x] / ww : w[x];
      kk[(size_t)xx * ksize + x] =
          v < 0 ? (int)(-0.5 + v * (double)(1 << RS_PRECISION_BITS)) : (int)(0.5 + v * (double)(1 << RS_PRECISION_BITS));
    }
    bounds[xx * 2 + 0] = xmin;
    bounds[xx * 2 + 1] = xmax;
  }
  return VS_OK;
}

__device__ __forceinline__ uint8_t rs_clip8(int v) {
  v >>= RS_PRECISION_BITS;
  return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

// horizontal pass over source rows y0 .. y0+Hy: tmp[f][yy][xo][3]
__global__ void resize_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp,
                                const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                int H0, int W0, int y0, int Hy, int Wo, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int xo = (int)(i % Wo);
  const long long r = i / Wo;
  const int yy = (int)(r % Hy);
  const long long f = r / Hy;
  const int xmin = bounds[xo * 2], cnt = bounds[xo * 2 + 1];
  const int* k = kk + (long long)xo * ksize;
  const uint8_t* p = src + ((f * H0 + y0 + yy) * W0 + xmin) * 3;
  int s0 = 1 << (RS_PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < cnt; ++x) {
    const int kx = k[x];
    s0 += p[x * 3 + 0] * kx;
    s1 += p[x * 3 + 1] * kx;
    s2 += p[x * 3 + 2] * kx;
  }
  uint8_t* o = tmp + i * 3;
  o[0] = rs_clip8(s0);
  o[1] = rs_clip8(s1);
  o[2] = rs_clip8(s2);
}

// vertical pass: dst[f][yo][xo][3] from in[f][Hin][Wo][3]; source row = bounds[yo].first - yshift
__global__ void resize_v_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ dst,
                                const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                int Hin, int yshift, int Ho, int Wo, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int xo = (int)(i % Wo);
  const long long r = i / Wo;
  const int yo = (int)(r % Ho);
  const long long f = r / Ho;
  const int ymin = bounds[yo * 2] - yshift, cnt = bounds[yo * 2 + 1];
  const int* k = kk + (long long)yo * ksize;
  const uint8_t* p = in + ((f * Hin + ymin) * Wo + xo) * 3;
  int s0 = 1 << (RS_PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int y = 0; y < cnt; ++y) {
    const int ky = k[y];
    const uint8_t* q = p + (long long)y * Wo * 3;
    s0 += q[0] * ky;
    s1 += q[1] * ky;
    s2 += q[2] * ky;
  }
  uint8_t* o = dst + i * 3;
  o[0] = rs_clip8(s0);
  o[1] = rs_clip8(s1);
  o[2] = rs_clip8(s2);
}

extern "C" size_t vs_resize_tmp_bytes(int64_t frames, int H0, int Wo) { return (size_t)frames * H0 * Wo * 3; }

extern "C" int vs_resize_bicubic_u8(const uint8_t* src, uint8_t* dst, uint8_t* tmp, int64_t frames, int H0,
                                    int W0, int Ho, int Wo, const int32_t* bounds_h, const int32_t* kk_h,
                                    int ksize_h, const int32_t* bounds_v, const int32_t* kk_v, int ksize_v,
                                    int y0, int y1, void* stream) {
  VS_CHECK_ARG(src && dst && frames > 0 && H0 > 0 && W0 > 0 && Ho > 0 && Wo > 0, "bad args");
  const bool need_h = W0 != Wo, need_v = H0 != Ho;
  VS_CHECK_ARG(!need_h || (bounds_h && kk_h && ksize_h > 0), "horizontal tables missing");
  VS_CHECK_ARG(!need_v || (bounds_v && kk_v && ksize_v > 0), "vertical tables missing");
  VS_CHECK_ARG(!(need_h && need_v) || tmp, "two passes need the tmp buffer");
  if (!need_v) { y0 = 0; y1 = H0; }
  VS_CHECK_ARG(y0 >= 0 && y1 > y0 && y1 <= H0, "row range of the vertical pass");
  hipStream_t st = (hipStream_t)stream;
  if (!need_h && !need_v) {
    (void)hipMemcpyAsync(dst, src, (size_t)frames * H0 * W0 * 3, hipMemcpyDeviceToDevice, st);
    VS_CHECK_LAUNCH();
    return VS_OK;
  }
  const uint8_t* vin = src;
  int Hin = H0, yshift = 0;
  if (need_h) {
    uint8_t* hout = need_v ? tmp : dst;
    const int Hy = y1 - y0;
    const long long total = (long long)frames * Hy * Wo;
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, hout,
                       bounds_h, kk_h, ksize_h, H0, W0, y0, Hy, Wo, total);
    vin = hout;
    Hin = Hy;
    yshift = y0;
  }
  if (need_v) {
    const long long total = (long long)frames * Ho * Wo;
    hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, vin, dst,
                       bounds_v, kk_v, ksize_v, Hin, yshift, Ho, Wo, total);
  }
  VS_CHECK_LAUNCH();
  return VS_OK;
}
