// HBM-bound gather kernels: crop+resize, ROI pyramid gather, bilinear upsample, layout changes.
// One work item = one output pixel x one 16-byte channel group; consecutive lanes walk the
// channel dimension first (NHWC), so every load/store instruction covers whole 64-1024 B runs.
// Grid = (chunks of one output row, output row, image): the row / image coordinates are scalar
// (blockIdx) and the only per-thread index arithmetic is one 32-bit divide -- the flat 64-bit
// index decomposition these kernels started with cost more ALU time than the memory traffic.
#include "common.h"

PRV2_NO_PACKED_FP32_BEGIN  // (common.h)

namespace prv2 {

// ---------------------------------------------------------------------------------------------
// crop + bilinear(align_corners) resize, CHW image -> NHWC patches, (v-mean)/std fused
// ---------------------------------------------------------------------------------------------
struct Norm3 {
  float mean[3];
  float std[3];
};

__global__ void __launch_bounds__(256) crop_resize_kernel(const float* __restrict__ img, int H, int W,
                                                          const int* __restrict__ tiles, int K, int ch, int cw, int oh,
                                                          int ow, float sy, float sx, Norm3 nrm, float* __restrict__ out,
                                                          int ldo) {
  const int ox = blockIdx.x * blockDim.x + threadIdx.x;
  const int oy = blockIdx.y, k = blockIdx.z;
  if (ox >= ow) return;
  const int64_t idx = ((int64_t)k * oh + oy) * ow + ox;
  int h0 = tiles[2 * k], w0 = tiles[2 * k + 1];
  AxisTap ty = ac_tap(oy, sy, ch), tx = ac_tap(ox, sx, cw);
  const float* base = img + (int64_t)(h0)*W + w0;
  float* o = out + idx * ldo;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* p = base + (int64_t)c * H * W;
    float v00 = p[(int64_t)ty.i0 * W + tx.i0], v01 = p[(int64_t)ty.i0 * W + tx.i1];
    float v10 = p[(int64_t)ty.i1 * W + tx.i0], v11 = p[(int64_t)ty.i1 * W + tx.i1];
    float v = ty.w0 * (tx.w0 * v00 + tx.w1 * v01) + ty.w1 * (tx.w0 * v10 + tx.w1 * v11);
    o[c] = (v - nrm.mean[c]) / nrm.std[c];
  }
}

// ---------------------------------------------------------------------------------------------
// image input stage: HWC uint8 (or fp32) RGB -> /255 -> bicubic(align_corners=True, A = -0.75) -> CHW fp32.
// The reference does this on the CPU in float64 (cv2 image / 255.0 is a float64 array; general_dataset.py:55-60), so the
// arithmetic here is double as well and rounded to fp32 once, tap order as in ATen's separable kernel (x then y).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void cubic_coeffs(double t, double (&c)[4]) {
  const double A = -0.75;
  const double x0 = t + 1.0, x3 = (1.0 - t) + 1.0, x2 = 1.0 - t;
  c[0] = ((A * x0 - 5.0 * A) * x0 + 8.0 * A) * x0 - 4.0 * A;
  c[1] = ((A + 2.0) * t - (A + 3.0)) * t * t + 1.0;
  c[2] = ((A + 2.0) * x2 - (A + 3.0)) * x2 * x2 + 1.0;
  c[3] = ((A * x3 - 5.0 * A) * x3 + 8.0 * A) * x3 - 4.0 * A;
}

template <typename T>
__global__ void __launch_bounds__(256) bicubic_resize_kernel(const T* __restrict__ src, int h, int w, float* __restrict__ dst,
                                                             int H, int W, double sy, double sx, double norm) {
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
  if (ox >= W) return;
  const double ry = sy * oy, rx = sx * ox;
  const int iy = (int)floor(ry), ix = (int)floor(rx);
  double cy[4], cx[4];
  cubic_coeffs(ry - iy, cy);
  cubic_coeffs(rx - ix, cx);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int yy = min(max(iy - 1 + i, 0), h - 1);
      double row = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int xx = min(max(ix - 1 + j, 0), w - 1);
        row += cx[j] * ((double)src[((int64_t)yy * w + xx) * 3 + c] / norm);
      }
      acc += cy[i] * row;
    }
    dst[((int64_t)c * H + oy) * W + ox] = (float)acc;
  }
}

// ---------------------------------------------------------------------------------------------
// roi_align(aligned=True, sampling_ratio=-1) from ONE feature map to K outputs (no repeat(K))
// ---------------------------------------------------------------------------------------------
template <int VEC>
struct VecT;
template <>
struct VecT<4> {
  using type = float4;
};
template <>
struct VecT<1> {
  using type = float;
};

__device__ __forceinline__ void vfma(float4& a, float w, const float4& v) {
  a.x += w * v.x;
  a.y += w * v.y;
  a.z += w * v.z;
  a.w += w * v.w;
}
__device__ __forceinline__ void vfma(float& a, float w, const float& v) { a += w * v; }
__device__ __forceinline__ float4 vzero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void vscale(float4& a, float s) {
  a.x *= s;
  a.y *= s;
  a.z *= s;
  a.w *= s;
}
__device__ __forceinline__ void vscale(float& a, float s) { a *= s; }
__device__ __forceinline__ void vdiv(float4& a, float s) {
  a.x /= s;
  a.y /= s;
  a.z /= s;
  a.w /= s;
}
__device__ __forceinline__ void vdiv(float& a, float s) { a /= s; }

// one thread's 4 consecutive channels c .. c + 3 of an output pixel: fp32, or (X2) the pre-split format of prv2_conv_desc.fmt --
// per 8 channels [8 x bf16 hi | 8 x bf16 lo]: the 4 channels are 8 bytes of the group's hi half and 8 bytes of its lo half
template <bool X2>
__device__ __forceinline__ void store4_fmt(float* pix_base, int c, const float4 v) {
  if constexpr (!X2) {
    *reinterpret_cast<float4*>(pix_base + c) = v;
  } else {
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const f32x4_t f = {v.x, v.y, v.z, v.w};
    const bf16x4_t hi = __builtin_convertvector(f, bf16x4_t);
    const bf16x4_t lo = __builtin_convertvector(f - __builtin_convertvector(hi, f32x4_t), bf16x4_t);
    char* g = reinterpret_cast<char*>(pix_base + (c & ~7)) + ((c >> 2) & 1) * 8;
    *reinterpret_cast<bf16x4_t*>(g) = hi;
    *reinterpret_cast<bf16x4_t*>(g + 16) = lo;
  }
}

template <int VEC, bool X2 = false>
__global__ void __launch_bounds__(256) roi_align_kernel(const float* __restrict__ feat, int H, int W, int C, int ldf,
                                                        const float* __restrict__ boxes, int K, float scale, int oh,
                                                        int ow, float* __restrict__ out, int ldo) {
  using V = typename VecT<VEC>::type;
  const unsigned cg = C / VEC;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (unsigned)ow * cg) return;
  const int px = (int)(t / cg), c = (int)(t - (unsigned)px * cg) * VEC;
  const int py = blockIdx.y, k = blockIdx.z;
  const int64_t pix = ((int64_t)k * oh + py) * ow + px;
  {
    const float* b = boxes + 4 * k;
    // torchvision roi_align_forward_kernel_impl, aligned=True
    float rsw = b[0] * scale - 0.5f, rsh = b[1] * scale - 0.5f;
    float rew = b[2] * scale - 0.5f, reh = b[3] * scale - 0.5f;
    float roi_w = rew - rsw, roi_h = reh - rsh;
    float bin_h = roi_h / (float)oh, bin_w = roi_w / (float)ow;
    int gh = (int)ceilf(roi_h / (float)oh), gw = (int)ceilf(roi_w / (float)ow);
    float count = (float)max(gh * gw, 1);
    V acc;
    if constexpr (VEC == 4) acc = vzero4(); else acc = 0.f;
    for (int iy = 0; iy < gh; ++iy) {
      float y = rsh + (float)py * bin_h + ((float)iy + .5f) * bin_h / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        float x = rsw + (float)px * bin_w + ((float)ix + .5f) * bin_w / (float)gw;
        if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
        float yy = y <= 0.f ? 0.f : y, xx = x <= 0.f ? 0.f : x;
        int yl = (int)yy, xl = (int)xx, yh, xh;
        if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
        if (xl >= W - 1) { xh = xl = W - 1; xx = (float)xl; } else xh = xl + 1;
        float ly = yy - (float)yl, lx = xx - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
        float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const V v1 = *reinterpret_cast<const V*>(feat + ((int64_t)yl * W + xl) * ldf + c);
        const V v2 = *reinterpret_cast<const V*>(feat + ((int64_t)yl * W + xh) * ldf + c);
        const V v3 = *reinterpret_cast<const V*>(feat + ((int64_t)yh * W + xl) * ldf + c);
        const V v4 = *reinterpret_cast<const V*>(feat + ((int64_t)yh * W + xh) * ldf + c);
        // val = w1*v1 + w2*v2 + w3*v3 + w4*v4 (left to right), then output_val += val
        V val;
        if constexpr (VEC == 4) val = vzero4(); else val = 0.f;
        vfma(val, w1, v1);
        vfma(val, w2, v2);
        vfma(val, w3, v3);
        vfma(val, w4, v4);
        vfma(acc, 1.0f, val);
      }
    }
    vdiv(acc, count);
    if constexpr (VEC == 4) store4_fmt<X2>(out + pix * ldo, c, acc);
    else *reinterpret_cast<V*>(out + pix * ldo + c) = acc;
  }
}

// The upsampling regime of the path (an ROI is 1/S of the coarse map and comes out at the map's own size: bin < 1 pixel, one
// sample per bin): R output rows per thread.  Consecutive output rows mostly share their two source rows (a x4 zoom: 4 output
// rows per source row), so the four taps stay in registers while (y_low, y_high) does not change -- ~1.3 instead of 4 tap loads
// per output, the SAME arithmetic per output as roi_align_kernel (w1 v1 + w2 v2 + w3 v3 + w4 v4, left to right).  Boxes whose
// sampling grid is larger than 1 x 1 (down-sampling ROIs) take the general per-pixel loop.
template <int R, bool X2 = false>
__global__ void __launch_bounds__(256) roi_align_rows_kernel(const float* __restrict__ feat, int H, int W, int C, int ldf,
                                                             const float* __restrict__ boxes, int K, float scale, int oh, int ow,
                                                             float* __restrict__ out, int ldo) {
  const unsigned cg = C / 4;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (unsigned)ow * cg) return;
  const int px = (int)(t / cg), c = (int)(t - (unsigned)px * cg) * 4;
  const int py0 = blockIdx.y * R, k = blockIdx.z;
  const float* b = boxes + 4 * k;
  const float rsw = b[0] * scale - 0.5f, rsh = b[1] * scale - 0.5f;
  const float rew = b[2] * scale - 0.5f, reh = b[3] * scale - 0.5f;
  const float roi_w = rew - rsw, roi_h = reh - rsh;
  const float bin_h = roi_h / (float)oh, bin_w = roi_w / (float)ow;
  const int gh = (int)ceilf(roi_h / (float)oh), gw = (int)ceilf(roi_w / (float)ow);
  if (gh != 1 || gw != 1) {  // block-uniform (a property of box k): the general sampling grid, one output row at a time
    // (gh or gw == 0: a degenerate box with roi_h / roi_w <= 0 -- zero samples, output 0, like roi_align_kernel and torchvision)
#pragma unroll 1
    for (int r = 0; r < R && py0 + r < oh; ++r) {
      const int py = py0 + r;
      const float count = (float)max(gh * gw, 1);
      float4 acc = vzero4();
      for (int iy = 0; iy < gh; ++iy) {
        const float y = rsh + (float)py * bin_h + ((float)iy + .5f) * bin_h / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
          const float x = rsw + (float)px * bin_w + ((float)ix + .5f) * bin_w / (float)gw;
          if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
          float yy = y <= 0.f ? 0.f : y, xx = x <= 0.f ? 0.f : x;
          int yl = (int)yy, xl = (int)xx, yh, xh;
          if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
          if (xl >= W - 1) { xh = xl = W - 1; xx = (float)xl; } else xh = xl + 1;
          const float ly = yy - (float)yl, lx = xx - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
          float4 val = vzero4();
          vfma(val, hy * hx, *reinterpret_cast<const float4*>(feat + ((int64_t)yl * W + xl) * ldf + c));
          vfma(val, hy * lx, *reinterpret_cast<const float4*>(feat + ((int64_t)yl * W + xh) * ldf + c));
          vfma(val, ly * hx, *reinterpret_cast<const float4*>(feat + ((int64_t)yh * W + xl) * ldf + c));
          vfma(val, ly * lx, *reinterpret_cast<const float4*>(feat + ((int64_t)yh * W + xh) * ldf + c));
          vfma(acc, 1.0f, val);
        }
      }
      vdiv(acc, count);
      store4_fmt<X2>(out + (((int64_t)k * oh + py) * ow + px) * ldo, c, acc);
    }
    return;
  }
  // one sample per bin: x is the thread's own, shared by its R rows
  const float x = rsw + (float)px * bin_w + .5f * bin_w / 1.0f;
  const bool x_in = !(x < -1.0f || x > (float)W);
  float xx = x <= 0.f ? 0.f : x;
  int xl = (int)xx, xh;
  if (xl >= W - 1) { xh = xl = W - 1; xx = (float)xl; } else xh = xl + 1;
  const float lx = xx - (float)xl, hx = 1.f - lx;
  int cyl = -1, cyh = -1;  // rows whose taps are in registers (block-uniform: the block's threads share their output rows)
  float4 v1 = vzero4(), v2 = vzero4(), v3 = vzero4(), v4 = vzero4();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int py = py0 + r;
    if (py >= oh) break;
    const float y = rsh + (float)py * bin_h + .5f * bin_h / 1.0f;
    float4 acc = vzero4();
    if (x_in && !(y < -1.0f || y > (float)H)) {
      float yy = y <= 0.f ? 0.f : y;
      int yl = (int)yy, yh;
      if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
      const float ly = yy - (float)yl, hy = 1.f - ly;
      if (yl != cyl || yh != cyh) {
        v1 = *reinterpret_cast<const float4*>(feat + ((int64_t)yl * W + xl) * ldf + c);
        v2 = *reinterpret_cast<const float4*>(feat + ((int64_t)yl * W + xh) * ldf + c);
        v3 = *reinterpret_cast<const float4*>(feat + ((int64_t)yh * W + xl) * ldf + c);
        v4 = *reinterpret_cast<const float4*>(feat + ((int64_t)yh * W + xh) * ldf + c);
        cyl = yl;
        cyh = yh;
      }
      float4 val = vzero4();
      vfma(val, hy * hx, v1);
      vfma(val, hy * lx, v2);
      vfma(val, ly * hx, v3);
      vfma(val, ly * lx, v4);
      vfma(acc, 1.0f, val);
    }
    vdiv(acc, 1.0f);
    store4_fmt<X2>(out + (((int64_t)k * oh + py) * ow + px) * ldo, c, acc);
  }
}

// ---------------------------------------------------------------------------------------------
// bilinear align_corners=True upsample, NHWC
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256) upsample_bilinear_kernel(const float* __restrict__ x, int N, int H, int W, int C,
                                                                int ldx, int oh, int ow, float sy, float sx,
                                                                float* __restrict__ y, int ldy) {
  using V = typename VecT<VEC>::type;
  const unsigned cg = C / VEC;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (unsigned)ow * cg) return;
  const int ox = (int)(t / cg), c = (int)(t - (unsigned)ox * cg) * VEC;
  const int oy = blockIdx.y, n = blockIdx.z;
  const int64_t pix = ((int64_t)n * oh + oy) * ow + ox;
  {
    AxisTap ty = ac_tap(oy, sy, H), tx = ac_tap(ox, sx, W);
    const float* p = x + (int64_t)n * H * W * ldx + c;
    const V v00 = *reinterpret_cast<const V*>(p + ((int64_t)ty.i0 * W + tx.i0) * ldx);
    const V v01 = *reinterpret_cast<const V*>(p + ((int64_t)ty.i0 * W + tx.i1) * ldx);
    const V v10 = *reinterpret_cast<const V*>(p + ((int64_t)ty.i1 * W + tx.i0) * ldx);
    const V v11 = *reinterpret_cast<const V*>(p + ((int64_t)ty.i1 * W + tx.i1) * ldx);
    V top, bot, r;
    if constexpr (VEC == 4) { top = vzero4(); bot = vzero4(); r = vzero4(); } else { top = 0.f; bot = 0.f; r = 0.f; }
    vfma(top, tx.w0, v00);
    vfma(top, tx.w1, v01);
    vfma(bot, tx.w0, v10);
    vfma(bot, tx.w1, v11);
    vfma(r, ty.w0, top);
    vfma(r, ty.w1, bot);
    *reinterpret_cast<V*>(y + pix * ldy + c) = r;
  }
}

// The same, R output rows per thread (float4 over channels): consecutive output rows share source rows (a x2 upsample
// needs ~3 source rows for 4 output rows), so the horizontally interpolated source rows are kept in a two-entry cache --
// 1.5 instead of 4 tap loads per output, identical arithmetic per output (top / bot are formed exactly as above).
template <int R>
__global__ void __launch_bounds__(256) upsample_bilinear_rows_kernel(const float* __restrict__ x, int N, int H, int W, int C, int ldx,
                                                                     int oh, int ow, float sy, float sx, float* __restrict__ y,
                                                                     int ldy) {
  const unsigned cg = C / 4;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (unsigned)ow * cg) return;
  const int ox = (int)(t / cg), c = (int)(t - (unsigned)ox * cg) * 4;
  const int oy0 = blockIdx.y * R, n = blockIdx.z;
  const AxisTap tx = ac_tap(ox, sx, W);
  const float* p = x + (int64_t)n * H * W * ldx + c;
  auto hrow = [&](int i) {  // horizontally interpolated source row i at ox
    const float4 a = *reinterpret_cast<const float4*>(p + ((int64_t)i * W + tx.i0) * ldx);
    const float4 b = *reinterpret_cast<const float4*>(p + ((int64_t)i * W + tx.i1) * ldx);
    float4 h = vzero4();
    vfma(h, tx.w0, a);
    vfma(h, tx.w1, b);
    return h;
  };
  int ia = -1, ib = -1;  // cached source rows (block-uniform: every thread of a block works on the same output rows)
  float4 ha = vzero4(), hb = vzero4();
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int oy = oy0 + k;
    if (oy >= oh) break;
    const AxisTap ty = ac_tap(oy, sy, H);
    float4 top, bot;
    if (ty.i0 == ia) top = ha;
    else if (ty.i0 == ib) top = hb;
    else {
      top = hrow(ty.i0);
      ia = ib; ha = hb; ib = ty.i0; hb = top;
    }
    if (ty.i1 == ia) bot = ha;
    else if (ty.i1 == ib) bot = hb;
    else {
      bot = hrow(ty.i1);
      ia = ib; ha = hb; ib = ty.i1; hb = bot;
    }
    float4 r = vzero4();
    vfma(r, ty.w0, top);
    vfma(r, ty.w1, bot);
    *reinterpret_cast<float4*>(y + (((int64_t)n * oh + oy) * ow + ox) * ldy + c) = r;
  }
}

// Two dense 1-channel maps (the two depth predictions every fusion level appends to its features: fusion_model.py:91-118,
// bi_directional_fusion_model.py:424-436) resized bilinear(align_corners=True) and written as channels c, c + 1 of an NHWC
// buffer whose last four channels are [pred1 | pred2 | pad | pad]: ONE 16-byte store per pixel instead of two scattered
// 4-byte stores per pixel plus the pad-zeroing pass.  Arithmetic identical to upsample_bilinear_kernel<1> per map.
__global__ void __launch_bounds__(256) depth_pair_fill_kernel(const float* __restrict__ p1, const float* __restrict__ p2, int H, int W,
                                                              int oh, int ow, float sy, float sx, float* __restrict__ y, int ldy) {
  const int ox = blockIdx.x * blockDim.x + threadIdx.x;
  if (ox >= ow) return;
  const int oy = blockIdx.y, n = blockIdx.z;
  const AxisTap ty = ac_tap(oy, sy, H), tx = ac_tap(ox, sx, W);
  auto interp = [&](const float* p) {
    p += (int64_t)n * H * W;
    const float v00 = p[(int64_t)ty.i0 * W + tx.i0], v01 = p[(int64_t)ty.i0 * W + tx.i1];
    const float v10 = p[(int64_t)ty.i1 * W + tx.i0], v11 = p[(int64_t)ty.i1 * W + tx.i1];
    float top = 0.f, bot = 0.f, r = 0.f;
    vfma(top, tx.w0, v00);
    vfma(top, tx.w1, v01);
    vfma(bot, tx.w0, v10);
    vfma(bot, tx.w1, v11);
    vfma(r, ty.w0, top);
    vfma(r, ty.w1, bot);
    return r;
  };
  const float4 o = make_float4(interp(p1), interp(p2), 0.f, 0.f);
  *reinterpret_cast<float4*>(y + (((int64_t)n * oh + oy) * ow + ox) * ldy) = o;
}

// y[n, oy, ox, :c] -= sum of bt[tap][:] over the taps (ky, kx) of a 3 x 3 / pad 1 conv that fall outside the image at (oy, ox): the
// border correction of a conv whose bias was folded from an upstream constant (conv3x3(x + b) = conv3x3(x) + sum_taps W_tap b holds
// only where all nine taps see b; with zero padding the border pixels see fewer).  One thread = one border pixel x 4 channels.
__global__ void __launch_bounds__(256) conv_border_bias_kernel(float* __restrict__ y, int N, int H, int W, int C, int ldy,
                                                               const float* __restrict__ bt /* [9][C] */) {
  const int per_img = 2 * W + 2 * (H - 2 > 0 ? H - 2 : 0), c4n = C / 4;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)N * per_img * c4n) return;
  const int c = (int)(t % c4n) * 4;
  const int b = (int)((t / c4n) % per_img), n = (int)(t / c4n / per_img);
  int oy, ox;
  if (b < W) { oy = 0; ox = b; }
  else if (b < 2 * W) { oy = H - 1; ox = b - W; }
  else { const int r = b - 2 * W; oy = 1 + (r >> 1); ox = (r & 1) ? W - 1 : 0; }
  if (H == 1 && b >= W) return;  // (one row: top == bottom, counted once)
  float4 s = vzero4();
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy + ky - 1, ix = ox + kx - 1;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) continue;
      vfma(s, 1.0f, *reinterpret_cast<const float4*>(bt + (ky * 3 + kx) * C + c));
    }
  float4* q = reinterpret_cast<float4*>(y + (((long long)n * H + oy) * W + ox) * ldy + c);
  float4 v = *q;
  v.x -= s.x; v.y -= s.y; v.z -= s.z; v.w -= s.w;
  *q = v;
}

__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ x, int N, int C, int H, int W,
                                                           float* __restrict__ y, int ldy) {
  int64_t total = (int64_t)N * H * W * C;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(idx % C);
    int64_t pix = idx / C;
    int64_t hw = pix % ((int64_t)H * W);
    int n = (int)(pix / ((int64_t)H * W));
    y[pix * ldy + c] = x[((int64_t)n * C + c) * H * W + hw];
  }
}

__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const float* __restrict__ x, int N, int C, int H, int W,
                                                           int ldx, float* __restrict__ y) {
  int64_t total = (int64_t)N * H * W * C;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t hw = idx % ((int64_t)H * W);
    int c = (int)((idx / ((int64_t)H * W)) % C);
    int n = (int)(idx / ((int64_t)H * W * C));
    y[idx] = x[((int64_t)n * H * W + hw) * ldx + c];
  }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace prv2

using namespace prv2;

extern "C" int prv2_crop_resize(const float* img, int32_t H, int32_t W, const int32_t* tiles, int32_t K, int32_t ch,
                                int32_t cw, int32_t oh, int32_t ow, const float* mean3, const float* std3, float* out,
                                int32_t ldo, void* stream) {
  PRV2_REQUIRE(img && tiles && out, "crop_resize: null pointer");
  PRV2_REQUIRE(K > 0 && ch > 0 && cw > 0 && oh > 0 && ow > 0 && ch <= H && cw <= W && ldo >= 3,
               "crop_resize: bad geometry K=%d crop=%dx%d out=%dx%d img=%dx%d ldo=%d", K, ch, cw, oh, ow, H, W, ldo);
  Norm3 n;
  for (int i = 0; i < 3; ++i) {
    n.mean[i] = mean3 ? mean3[i] : 0.f;
    n.std[i] = std3 ? std3[i] : 1.f;
  }
  PRV2_REQUIRE(oh <= 65535 && K <= 65535, "crop_resize: grid too large");
  hipLaunchKernelGGL(crop_resize_kernel, dim3((unsigned)cdiv(ow, 256), oh, K), dim3(256), 0, (hipStream_t)stream, img, H, W, tiles,
                     K, ch, cw, oh, ow, ac_scale(ch, oh), ac_scale(cw, ow), n, out, ldo);
  PRV2_LAUNCH_CHECK("crop_resize");
  return 0;
}

extern "C" int prv2_bicubic_resize(const void* src_hwc, int32_t src_is_u8, int32_t h, int32_t w, float* dst_chw, int32_t H,
                                   int32_t W, void* stream) {
  PRV2_REQUIRE(src_hwc && dst_chw && h > 0 && w > 0 && H > 0 && W > 0 && H <= 65535, "bicubic_resize: bad arguments");
  const double sy = H > 1 ? (double)(h - 1) / (double)(H - 1) : 0.0, sx = W > 1 ? (double)(w - 1) / (double)(W - 1) : 0.0;
  const dim3 grid((unsigned)cdiv(W, 256), H);
  if (src_is_u8)
    hipLaunchKernelGGL(bicubic_resize_kernel<uint8_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src_hwc, h, w,
                       dst_chw, H, W, sy, sx, 255.0);
  else
    hipLaunchKernelGGL(bicubic_resize_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src_hwc, h, w, dst_chw,
                       H, W, sy, sx, 1.0);
  PRV2_LAUNCH_CHECK("bicubic_resize");
  return 0;
}

static int roi_align_impl(const float* feat, int32_t h, int32_t w, int32_t c, int32_t ldf, const float* boxes, int32_t k, float spatial_scale,
                          int32_t oh, int32_t ow, float* out, int32_t ldo, void* stream, bool x2);

extern "C" int prv2_roi_align(const float* feat, int32_t h, int32_t w, int32_t c, int32_t ldf, const float* boxes,
                              int32_t k, float spatial_scale, int32_t oh, int32_t ow, float* out, int32_t ldo,
                              void* stream) {
  return roi_align_impl(feat, h, w, c, ldf, boxes, k, spatial_scale, oh, ow, out, ldo, stream, false);
}

extern "C" int prv2_roi_align_x2(const float* feat, int32_t h, int32_t w, int32_t c, int32_t ldf, const float* boxes,
                                 int32_t k, float spatial_scale, int32_t oh, int32_t ow, float* out, int32_t ldo,
                                 void* stream) {
  PRV2_REQUIRE(feat && out && c > 0 && c % 8 == 0 && ldf % 4 == 0 && ldo % 8 == 0 && aligned16(feat) && (reinterpret_cast<uintptr_t>(out) & 31) == 0,
               "roi_align_x2: c %% 8 == 0, ldo %% 8 == 0, 32-byte aligned output (c=%d ldo=%d)", c, ldo);
  return roi_align_impl(feat, h, w, c, ldf, boxes, k, spatial_scale, oh, ow, out, ldo, stream, true);
}

static int roi_align_impl(const float* feat, int32_t h, int32_t w, int32_t c, int32_t ldf, const float* boxes, int32_t k, float spatial_scale,
                          int32_t oh, int32_t ow, float* out, int32_t ldo, void* stream, bool x2) {
  PRV2_REQUIRE(feat && boxes && out, "roi_align: null pointer");
  PRV2_REQUIRE(h > 0 && w > 0 && c > 0 && k > 0 && oh > 0 && ow > 0 && ldf >= c && ldo >= c, "roi_align: bad geometry");
  bool vec = (c % 4 == 0) && (ldf % 4 == 0) && (ldo % 4 == 0) && aligned16(feat) && aligned16(out);
  PRV2_REQUIRE(oh <= 65535 && k <= 65535, "roi_align: grid too large");
  const dim3 grid((unsigned)cdiv((int64_t)ow * (vec ? c / 4 : c), 256), oh, k);
  if (x2) {  // (vec holds: checked by the caller)
    if (oh >= 16) {
      constexpr int R = 4;
      const dim3 grid_r((unsigned)cdiv((int64_t)ow * (c / 4), 256), (unsigned)cdiv(oh, R), k);
      hipLaunchKernelGGL((roi_align_rows_kernel<R, true>), grid_r, dim3(256), 0, (hipStream_t)stream, feat, h, w, c, ldf, boxes, k, spatial_scale, oh, ow,
                         out, ldo);
    } else
      hipLaunchKernelGGL((roi_align_kernel<4, true>), grid, dim3(256), 0, (hipStream_t)stream, feat, h, w, c, ldf, boxes, k, spatial_scale, oh, ow, out, ldo);
    PRV2_LAUNCH_CHECK("roi_align_x2");
    return 0;
  }
  if (vec && oh >= 16) {
    constexpr int R = 4;
    const dim3 grid_r((unsigned)cdiv((int64_t)ow * (c / 4), 256), (unsigned)cdiv(oh, R), k);
    hipLaunchKernelGGL(roi_align_rows_kernel<R>, grid_r, dim3(256), 0, (hipStream_t)stream, feat, h, w, c, ldf, boxes, k, spatial_scale, oh, ow,
                       out, ldo);
  } else if (vec)
    hipLaunchKernelGGL(roi_align_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, feat, h, w,
                       c, ldf, boxes, k, spatial_scale, oh, ow, out, ldo);
  else
    hipLaunchKernelGGL(roi_align_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, feat, h, w,
                       c, ldf, boxes, k, spatial_scale, oh, ow, out, ldo);
  PRV2_LAUNCH_CHECK("roi_align");
  return 0;
}

extern "C" int prv2_upsample_bilinear(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ldx,
                                      int32_t oh, int32_t ow, float* y, int32_t ldy, void* stream) {
  PRV2_REQUIRE(x && y, "upsample_bilinear: null pointer");
  PRV2_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && oh > 0 && ow > 0 && ldx >= c && ldy >= c,
               "upsample_bilinear: bad geometry");
  bool vec = (c % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && aligned16(x) && aligned16(y);
  PRV2_REQUIRE(oh <= 65535 && n <= 65535, "upsample_bilinear: grid too large");
  const dim3 grid((unsigned)cdiv((int64_t)ow * (vec ? c / 4 : c), 256), oh, n);
  if (vec && oh >= 16) {
    constexpr int R = 4;
    const dim3 grid_r((unsigned)cdiv((int64_t)ow * (c / 4), 256), (oh + R - 1) / R, n);
    hipLaunchKernelGGL(upsample_bilinear_rows_kernel<R>, grid_r, dim3(256), 0, (hipStream_t)stream, x, n, h, w, c, ldx, oh, ow,
                       ac_scale(h, oh), ac_scale(w, ow), y, ldy);
  } else if (vec)
    hipLaunchKernelGGL(upsample_bilinear_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, x, n,
                       h, w, c, ldx, oh, ow, ac_scale(h, oh), ac_scale(w, ow), y, ldy);
  else
    hipLaunchKernelGGL(upsample_bilinear_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, n,
                       h, w, c, ldx, oh, ow, ac_scale(h, oh), ac_scale(w, ow), y, ldy);
  PRV2_LAUNCH_CHECK("upsample_bilinear");
  return 0;
}

extern "C" int prv2_conv_border_bias(float* y, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ldy, const float* tap_bias, void* stream) {
  PRV2_REQUIRE(y && tap_bias && n > 0 && h > 0 && w > 1 && c > 0, "conv_border_bias: bad arguments");
  PRV2_REQUIRE(c % 4 == 0 && ldy % 4 == 0 && ldy >= c && aligned16(y) && aligned16(tap_bias), "conv_border_bias: 16-byte channel groups (c %d ldy %d)", c, ldy);
  const long long items = (long long)n * (2 * w + 2 * (h > 2 ? h - 2 : 0)) * (c / 4);
  hipLaunchKernelGGL(conv_border_bias_kernel, dim3((unsigned)cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, y, n, h, w, c, ldy, tap_bias);
  PRV2_LAUNCH_CHECK("conv_border_bias");
  return 0;
}

extern "C" int prv2_depth_pair_fill(const float* p1, const float* p2, int32_t n, int32_t h, int32_t w, int32_t oh, int32_t ow, float* y,
                                    int32_t ldy, void* stream) {
  PRV2_REQUIRE(p1 && p2 && y && n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "depth_pair_fill: bad arguments");
  PRV2_REQUIRE(ldy % 4 == 0 && aligned16(y), "depth_pair_fill: the four destination channels must be 16-byte aligned (ldy %d)", ldy);
  PRV2_REQUIRE(oh <= 65535 && n <= 65535, "depth_pair_fill: grid too large");
  hipLaunchKernelGGL(depth_pair_fill_kernel, dim3((unsigned)cdiv(ow, 256), oh, n), dim3(256), 0, (hipStream_t)stream, p1, p2, h, w, oh, ow,
                     ac_scale(h, oh), ac_scale(w, ow), y, ldy);
  PRV2_LAUNCH_CHECK("depth_pair_fill");
  return 0;
}

extern "C" int prv2_nchw_to_nhwc(const float* x, int32_t n, int32_t c, int32_t h, int32_t w, float* y, int32_t ldy,
                                 void* stream) {
  PRV2_REQUIRE(x && y && n > 0 && c > 0 && h > 0 && w > 0 && ldy >= c, "nchw_to_nhwc: bad arguments");
  int64_t total = (int64_t)n * c * h * w;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, n, c, h, w, y,
                     ldy);
  PRV2_LAUNCH_CHECK("nchw_to_nhwc");
  return 0;
}

extern "C" int prv2_nhwc_to_nchw(const float* x, int32_t n, int32_t c, int32_t h, int32_t w, int32_t ldx, float* y,
                                 void* stream) {
  PRV2_REQUIRE(x && y && n > 0 && c > 0 && h > 0 && w > 0 && ldx >= c, "nhwc_to_nchw: bad arguments");
  int64_t total = (int64_t)n * c * h * w;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, n, c, h, w,
                     ldx, y);
  PRV2_LAUNCH_CHECK("nhwc_to_nchw");
  return 0;
}

PRV2_NO_PACKED_FP32_END
