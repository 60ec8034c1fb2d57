// The coarse half of a ``cat([fine, coarse_roi])`` 3x3 convolution, computed ONCE PER FRAME at coarse resolution.
//
//   GatedConvUnit.forward        estimator/models/blocks/bi_directional_fusion_model.py:70-73   fusion_conv(cat([out, c_feat]))
//   BiDirectionalFusion.forward  ...bi_directional_fusion_model.py:424-426                      fusion_layers_1(cat([c, f]))
//   coarse_postprocess_test      estimator/models/patchrefinerplus.py:263-283                   c_feat = roi_align(feat.repeat(K), boxes, (h, w), h / P)
//
// ``c_feat`` is a bilinear zoom (roi_align with one sample per bin) of a window of the per-frame pyramid level F, and it enters the
// conv with no activation in front.  The conv is linear, so its coarse half is
//     B(p) = sum_tap [p + d_tap inside the tile] * Bil(G_tap; s(p + d_tap)),      G_tap = W_coarse[:, :, tap] . F   (1x1 GEMM at coarse resolution)
// with s(.) roi_align's sample position and Bil its clamped bilinear sample.  The reference evaluates 9 * Cin * Cout MACs per OUTPUT
// pixel of every one of the frame's 81 tiles; here the MFMA work is 9 * Cin * Cout MACs per COARSE pixel per frame.
//
// Tiles of one frame share their size, so consecutive output pixels are ``b`` = 1 / split apart in coarse coordinates on either axis,
// and U(y, x) = sum_tap Bil(G_tap; y + dy b_h, x + dx b_w) -- the sum WITHOUT the border mask -- is piecewise bilinear with knots at
// {k - b, k, k + b}, k integer.  It is therefore fixed by its values on that knot grid (``V``: 3H x 3W knots, the size of G), and a
// tile's B is a 4-tap gather from V -- the cost of the roi_align it replaces -- minus, on the tile's border pixels only, the taps the
// conv's zero padding hides (3 per edge pixel, 5 per corner, sampled from G).  Exact algebra; fp32 rounding differs (1e-7 relative).
//
// BUILD NOTE: this file is compiled with packed fp32 math OFF (PRV2_NO_PACKED_FP32_BEGIN below: a function attribute, common.h).  With
// hipcc 7.2's v_pk_fma_f32 / v_pk_add_f32 code for tap_gather_kernel the border pixels of a tile came out wrong INTERMITTENTLY --
// always the low halves of the packed pairs (channels 4k and 4k + 2) of the last quarter-wave (lanes 48-63), values off by one tap
// term -- but only inside a frame with two tile streams and a third stream busy; never in isolation, and independent of every
// s_waitcnt / s_nop added around the loads and the store (tools/probes/taps_stage_checksums.py, profiles/r04_experiments.txt).
// Scalar v_fma_f32 code is bit-stable under the same load; the kernels are HBM-bound, the flag costs nothing measurable.
#include <cstdlib>

#include "common.h"

PRV2_NO_PACKED_FP32_BEGIN  // (common.h)

namespace prv2 {

__device__ __forceinline__ void fma4(float4& a, float w, const float4& v) {
  a.x = fmaf(w, v.x, a.x);
  a.y = fmaf(w, v.y, a.y);
  a.z = fmaf(w, v.z, a.z);
  a.w = fmaf(w, v.w, a.w);
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// weights of the rows (k - 1, k, k + 1) for the position k + o * b, o in -2 .. 2, b <= 1/2 (clamped rows replicate: roi_align's
// ``x <= 0 -> 0`` / ``lo >= size - 1 -> lo = hi = size - 1`` rules)
__device__ __forceinline__ void offset_weights(int o, float b, float (&w)[3]) {
  const float t = (float)o * b;
  if (o < 0) { w[0] = -t; w[1] = 1.0f + t; w[2] = 0.f; }
  else { w[0] = 0.f; w[1] = 1.0f - t; w[2] = t; }
}

// V[3k + a][3m + c][ch] = sum_tap Bil(G_tap; k + (a - 1 + dy) b_h, m + (c - 1 + dx) b_w).
// G: [H, W, ldg], channel = tap * C + ch at ``g_off``; thread = (coarse pixel, 4 channels): per tap the 3 x 3 neighbourhood is
// loaded once (9 loads) and serves the 9 knots of the pixel.
__global__ void __launch_bounds__(256) tap_knots_kernel(const float* __restrict__ G, int H, int W, int C, int ldg, float bh, float bw,
                                                        float* __restrict__ V, int ldv) {
  const unsigned cg = C / 4;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (unsigned)W * cg) return;
  const int m = (int)(t / cg), ch = (int)(t - (unsigned)m * cg) * 4;
  const int k = blockIdx.y;
  const int rows[3] = {max(k - 1, 0), k, min(k + 1, H - 1)};
  const int cols[3] = {max(m - 1, 0), m, min(m + 1, W - 1)};
  float4 acc[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[a][c] = zero4();
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int tap = ky * 3 + kx;
      float4 g[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) g[r][q] = *reinterpret_cast<const float4*>(G + ((int64_t)rows[r] * W + cols[q]) * ldg + tap * C + ch);
      // horizontal first: hz[r][c] = sum_q wx[c][q] g[r][q]
      float4 hz[3][3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float wx[3];
        offset_weights(c - 1 + kx - 1, bw, wx);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          float4 s = zero4();
          fma4(s, wx[0], g[r][0]);
          fma4(s, wx[1], g[r][1]);
          fma4(s, wx[2], g[r][2]);
          hz[r][c] = s;
        }
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        float wy[3];
        offset_weights(a - 1 + ky - 1, bh, wy);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          fma4(acc[a][c], wy[0], hz[0][c]);
          fma4(acc[a][c], wy[1], hz[1][c]);
          fma4(acc[a][c], wy[2], hz[2][c]);
        }
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) *reinterpret_cast<float4*>(V + ((int64_t)(3 * k + a) * (3 * W) + 3 * m + c) * ldv + ch) = acc[a][c];
}

// the cell of the knot grid {k - b, k, k + b} (index 3k, 3k + 1, 3k + 2; n coarse samples) that holds position v, and the weight of
// its upper knot.  Below the first / above the last knot U is constant (every shifted sample is clamped there).
struct KnotCell {
  int i0, i1;
  float t;
};
__device__ __forceinline__ KnotCell knot_cell(float v, int n, float b) {
  const float kf = floorf(v), f = v - kf;
  const int k = (int)kf;
  KnotCell c;
  if (f < b) { c.i0 = 3 * k + 1; c.t = f / b; }
  else if (f < 1.0f - b) { c.i0 = 3 * k + 2; c.t = (f - b) / (1.0f - 2.0f * b); }
  else { c.i0 = 3 * k + 3; c.t = (f - (1.0f - b)) / b; }
  c.i1 = c.i0 + 1;
  if (c.i0 < 0) { c.i0 = 0; c.i1 = 0; }
  if (c.i1 > 3 * n - 1) { c.i1 = 3 * n - 1; c.i0 = min(c.i0, 3 * n - 1); }
  return c;
}

// roi_align's clamped axis sample: (lo, hi, weight of hi)
__device__ __forceinline__ void roi_axis(float v, int n, int& lo, int& hi, float& l) {
  float vv = v <= 0.f ? 0.f : v;
  lo = (int)vv;
  if (lo >= n - 1) { hi = lo = n - 1; vv = (float)lo; } else hi = lo + 1;
  l = vv - (float)lo;
}

// B[k, i, j, ch] = U(ys(i), xs(j)) - (taps hidden by the zero padding at the tile border).  Thread = (column j, 4 channels) x R
// output rows; the x-interpolated knot rows of the previous output row are kept (consecutive rows share one knot row).
template <int R, bool NT = false>
__global__ void __launch_bounds__(256) tap_gather_kernel(const float* __restrict__ V, const float* __restrict__ G, int H, int W, int C, int ldv,
                                                         int ldg, float kbh, float kbw, const float* __restrict__ boxes, float scale, int oh,
                                                         int ow, float* __restrict__ out, int ldo) {
  const unsigned cg = C / 4;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (unsigned)ow * cg) return;
  const int px = (int)(t / cg), ch = (int)(t - (unsigned)px * cg) * 4;
  const int py0 = blockIdx.y * R, k = blockIdx.z;
  const float* b = boxes + 4 * k;
  // torchvision roi_align_forward_kernel_impl, aligned=True (as gather.hip::roi_align_kernel)
  const float rsw = b[0] * scale - 0.5f, rsh = b[1] * scale - 0.5f;
  const float rew = b[2] * scale - 0.5f, reh = b[3] * scale - 0.5f;
  const float bin_h = (reh - rsh) / (float)oh, bin_w = (rew - rsw) / (float)ow;
  const float x = rsw + (float)px * bin_w + .5f * bin_w;
  const KnotCell cx = knot_cell(x, W, kbw);
  const int W3 = 3 * W;
  const bool xedge = px == 0 || px == ow - 1;
  int c0 = -1, c1 = -1;  // knot rows whose x-interpolated values are in registers
  float4 h0 = zero4(), h1 = zero4();
  auto hrow = [&](int i) {
    const float4 a = *reinterpret_cast<const float4*>(V + ((int64_t)i * W3 + cx.i0) * ldv + ch);
    const float4 c = *reinterpret_cast<const float4*>(V + ((int64_t)i * W3 + cx.i1) * ldv + ch);
    float4 r = zero4();
    fma4(r, 1.0f - cx.t, a);
    fma4(r, cx.t, c);
    return r;
  };
  auto tap_sample = [&](int tap, float yy, float xx) {  // Bil(G_tap; yy, xx)
    int yl, yh, xl, xh;
    float ly, lx;
    roi_axis(yy, H, yl, yh, ly);
    roi_axis(xx, W, xl, xh, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* g = G + tap * C + ch;
    float4 r = zero4();
    const float4 a0 = *reinterpret_cast<const float4*>(g + ((int64_t)yl * W + xl) * ldg);
    const float4 a1 = *reinterpret_cast<const float4*>(g + ((int64_t)yl * W + xh) * ldg);
    const float4 a2 = *reinterpret_cast<const float4*>(g + ((int64_t)yh * W + xl) * ldg);
    const float4 a3 = *reinterpret_cast<const float4*>(g + ((int64_t)yh * W + xh) * ldg);
    fma4(r, hy * hx, a0);
    fma4(r, hy * lx, a1);
    fma4(r, ly * hx, a2);
    fma4(r, ly * lx, a3);
    return r;
  };
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int py = py0 + r;
    if (py >= oh) break;
    const float y = rsh + (float)py * bin_h + .5f * bin_h;
    const KnotCell cy = knot_cell(y, H, kbh);
    float4 lo, hi;
    if (cy.i0 == c0) lo = h0; else if (cy.i0 == c1) lo = h1; else lo = hrow(cy.i0);
    if (cy.i1 == cy.i0) hi = lo; else if (cy.i1 == c1) hi = h1; else if (cy.i1 == c0) hi = h0; else hi = hrow(cy.i1);
    c0 = cy.i0; h0 = lo; c1 = cy.i1; h1 = hi;
    float4 acc = zero4();
    fma4(acc, 1.0f - cy.t, lo);
    fma4(acc, cy.t, hi);
    const bool top = py == 0, bot = py == oh - 1;
    if (top || bot || xedge) {  // (xedge is wave-uniform: a wave is one pixel column; top / bot are block-uniform)
      const bool left = px == 0, right = px == ow - 1;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const bool hidden = (ky == 0 && top) || (ky == 2 && bot) || (kx == 0 && left) || (kx == 2 && right);
          if (hidden) {
            const float4 s = tap_sample(ky * 3 + kx, y + (float)(ky - 1) * kbh, x + (float)(kx - 1) * kbw);
            acc.x -= s.x; acc.y -= s.y; acc.z -= s.z; acc.w -= s.w;
          }
        }
    }
    float4* dst = reinterpret_cast<float4*>(out + (((int64_t)k * oh + py) * ow + px) * ldo + ch);
    if constexpr (NT) {
      typedef float f32x4n __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(f32x4n{acc.x, acc.y, acc.z, acc.w}, reinterpret_cast<f32x4n*>(dst));
    } else {
      *dst = acc;
    }
  }
}

static inline bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace prv2

using namespace prv2;

extern "C" int prv2_coarse_tap_knots(const float* g, int32_t h, int32_t w, int32_t c, int32_t ldg, float knot_bh, float knot_bw, float* v,
                                     int32_t ldv, void* stream) {
  PRV2_REQUIRE(g && v, "coarse_tap_knots: null pointer");
  PRV2_REQUIRE(h > 0 && w > 0 && c > 0 && c % 4 == 0 && ldg >= 9 * c && ldg % 4 == 0 && ldv >= c && ldv % 4 == 0 && aligned16(g) && aligned16(v),
               "coarse_tap_knots: c %% 4 == 0, ldg >= 9 c, 16-byte aligned rows (c=%d ldg=%d ldv=%d)", c, ldg, ldv);
  PRV2_REQUIRE(knot_bh > 0.f && knot_bh <= 0.5f && knot_bw > 0.f && knot_bw <= 0.5f,
               "coarse_tap_knots: the knot offsets (tile size / frame size per axis) must be in (0, 1/2] (got %g, %g)", (double)knot_bh, (double)knot_bw);
  PRV2_REQUIRE(h <= 65535, "coarse_tap_knots: grid too large");
  const dim3 grid((unsigned)cdiv((int64_t)w * (c / 4), 256), (unsigned)h);
  hipLaunchKernelGGL(tap_knots_kernel, grid, dim3(256), 0, (hipStream_t)stream, g, h, w, c, ldg, knot_bh, knot_bw, v, ldv);
  PRV2_LAUNCH_CHECK("coarse_tap_knots");
  return 0;
}

extern "C" int prv2_coarse_tap_gather(const float* v, const float* g, int32_t h, int32_t w, int32_t c, int32_t ldv, int32_t ldg, float knot_bh,
                                      float knot_bw, const float* boxes, int32_t k, float spatial_scale, int32_t oh, int32_t ow, float* out,
                                      int32_t ldo, void* stream) {
  PRV2_REQUIRE(v && g && boxes && out, "coarse_tap_gather: null pointer");
  PRV2_REQUIRE(h > 0 && w > 0 && k > 0 && oh > 0 && ow > 0 && c > 0 && c % 4 == 0 && ldg >= 9 * c && ldg % 4 == 0 && ldv >= c && ldv % 4 == 0 &&
                   ldo >= c && ldo % 4 == 0 && aligned16(g) && aligned16(v) && aligned16(out),
               "coarse_tap_gather: c %% 4 == 0, ldg >= 9 c, 16-byte aligned rows (c=%d ldg=%d ldv=%d ldo=%d)", c, ldg, ldv, ldo);
  PRV2_REQUIRE(knot_bh > 0.f && knot_bh <= 0.5f && knot_bw > 0.f && knot_bw <= 0.5f,
               "coarse_tap_gather: the knot offsets must be in (0, 1/2] (got %g, %g)", (double)knot_bh, (double)knot_bw);
  static const int variant = getenv("PRV2_TAPG") ? atoi(getenv("PRV2_TAPG")) : 0;  // A/B switch: bit 0 = 8 rows per thread, bit 1 = nontemporal stores
  const int R = (variant & 1) && oh >= 32 ? 8 : 4;
  PRV2_REQUIRE(cdiv(oh, 4) <= 65535 && k <= 65535, "coarse_tap_gather: grid too large");
  const dim3 grid((unsigned)cdiv((int64_t)ow * (c / 4), 256), (unsigned)cdiv(oh, R), (unsigned)k);
#define PRV2_TG(R_, NT_) hipLaunchKernelGGL((tap_gather_kernel<R_, NT_>), grid, dim3(256), 0, (hipStream_t)stream, v, g, h, w, c, ldv, ldg, knot_bh, knot_bw, boxes, \
                                            spatial_scale, oh, ow, out, ldo)
  if (R == 8) { if (variant & 2) PRV2_TG(8, true); else PRV2_TG(8, false); }
  else { if (variant & 2) PRV2_TG(4, true); else PRV2_TG(4, false); }
#undef PRV2_TG
  PRV2_LAUNCH_CHECK("coarse_tap_gather");
  return 0;
}

PRV2_NO_PACKED_FP32_END
