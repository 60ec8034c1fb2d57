// Shared helpers for the gfx950 kernels (internal; the public surface is include/prv2.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/prv2.h"

// Packed fp32 math (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32) OFF for every function between PRV2_NO_PACKED_FP32_BEGIN / _END, in the
// SOURCE rather than in a build flag (so that no build route can bring the packed code back): hipcc 7.2's packed code for
// tap_gather_kernel gave intermittently wrong border pixels inside multi-stream frames (coarse_taps.hip, BUILD NOTE); the other
// HBM-bound gather / pointwise / blend kernels have the same code shape (float4 arithmetic under exec-masked branches) and take the
// attribute as a precaution (v_pk_mul / v_pk_add are IEEE-identical to their scalar forms: same bits, no measurable cost).
// Device pass only: the host pass does not know the feature.  tests/test_isa_hazards.py asserts the ISA of these files has no v_pk_*_f32.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PRV2_TAPS_PK)
#define PRV2_NO_PACKED_FP32_BEGIN _Pragma("clang attribute push(__attribute__((target(\"no-packed-fp32-ops\"))), apply_to = function)")
#define PRV2_NO_PACKED_FP32_END _Pragma("clang attribute pop")
#else
#define PRV2_NO_PACKED_FP32_BEGIN
#define PRV2_NO_PACKED_FP32_END
#endif

namespace prv2 {

void set_error(const char* fmt, ...);
void set_kernel(const char* name, int bn, int prec);  // what prv2_last_kernel() reports

#define PRV2_REQUIRE(cond, ...)       \
  do {                                \
    if (!(cond)) {                    \
      prv2::set_error(__VA_ARGS__);   \
      return 1;                       \
    }                                 \
  } while (0)

#define PRV2_LAUNCH_CHECK(name)                                                      \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      prv2::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));        \
      return 2;                                                                      \
    }                                                                                \
  } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t roundup(int64_t a, int64_t b) { return cdiv(a, b) * b; }

// memory-bound grids: cap at 256 CUs x 8 blocks and grid-stride the rest
static inline int flat_grid(int64_t work_items, int block) {
  int64_t g = cdiv(work_items, block);
  if (g > 2048 * 4) g = 2048 * 4;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case PRV2_ACT_RELU: return v > 0.f ? v : 0.f;
    case PRV2_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));  // exact-erf GELU (mlp.py:31)
    case PRV2_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case PRV2_ACT_SOFTPLUS: return v > 20.f ? v : log1pf(expf(v));  // nn.Softplus(beta=1, threshold=20)
    case PRV2_ACT_SILU: return v / (1.0f + expf(-v));                  // nn.SiLU: x * sigmoid(x)
    default: return v;
  }
}

// GELU for the store loops of the bf16-mode conv kernels (16 rows x 4 channels per thread: VALU-bound; the exact-erf GELU above
// is ocml's two-branch erff, ~45 instructions under divergence -- +4..9 % on the short-K decoder layers): Abramowitz-Stegun 7.1.26
// on v_rcp_f32 / v_exp_f32, ~14 instructions.  Max |error| against float64 GELU over [-8, 8]: 4.7e-7 -- the same as the exact
// formula evaluated in fp32 (4.5e-7).  The f32 mode and the generic kernels keep act_apply (the ViT linears -- gemm16_kernel's lean store loop and
// gemm_ss.hip -- moved to gelu_fast in round 6, together, so that they stay bit-equal to each other).
__device__ __forceinline__ float gelu_fast(float v) {
  const float x = v * 0.70710678118654752440f, ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  float p = 1.061405429f * t;
  p = (p - 1.453152027f) * t;
  p = (p + 1.421413741f) * t;
  p = (p - 0.284496736f) * t;
  p = (p + 0.254829592f) * t;
  const float r = 1.0f - p * __builtin_amdgcn_exp2f(ax * ax * -1.44269504088896340736f);
  return 0.5f * v * (1.0f + copysignf(r, x));
}
__device__ __forceinline__ float act_apply_bf(float v, int act) { return act == PRV2_ACT_GELU ? gelu_fast(v) : act_apply(v, act); }

// bilinear / align_corners=True source coordinate (float32, PyTorch's area_pixel_compute_scale)
struct AxisTap {
  int i0, i1;
  float w0, w1;
};
__device__ __forceinline__ AxisTap ac_tap(int dst, float scale, int n_in) {
  float src = scale * (float)dst;
  int i0 = (int)src;
  if (i0 > n_in - 1) i0 = n_in - 1;
  int i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  float w1 = src - (float)i0;
  AxisTap t = {i0, i1, 1.0f - w1, w1};
  return t;
}
static inline float ac_scale(int n_in, int n_out) { return n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.0f; }

// legacy 'nearest': min(floor(dst * float(in/out)), in-1)
__device__ __forceinline__ int nearest_src(int dst, float scale, int n_in) {
  int s = (int)floorf((float)dst * scale);
  return s < n_in - 1 ? s : n_in - 1;
}

}  // namespace prv2
