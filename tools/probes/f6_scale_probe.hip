// GPU probe: which lane's E8M0 scale does v_mfma_scale_f32_16x16x128_f8f6f4 apply to which 32-k block of which row?
// Elements = 1.0 (e2m3 0b001000) in ONE lane group g0 of operand A (zeros elsewhere), all ones in B; A scale of lane L = 2^(L >> 4)
// (+ 8 if L & 1), B scales 1: D[m][n] / 32 = the scale the hardware applied to A's block held by lanes 16 g0 + m.
//   hipcc -O2 --offload-arch=gfx950 f6_scale_probe.hip -o f6_scale_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out) {
  const int lane = threadIdx.x;
  unsigned w[6];
  for (int i = 0; i < 6; ++i) w[i] = 0;
  for (int e = 0; e < 32; ++e) { const int bit = 6 * e + 3; w[bit / 32] |= 1u << (bit % 32); }
  const i32x8 ones = i32x8{(int)w[0], (int)w[1], (int)w[2], (int)w[3], (int)w[4], (int)w[5], 0, 0}, zeros = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
  for (int side = 0; side < 2; ++side)
    for (int g0 = 0; g0 < 4; ++g0) {
      const int sl = 127 + (lane >> 4) + ((lane & 1) ? 8 : 0), s1 = 127;
      const i32x8 sel = (lane >> 4) == g0 ? ones : zeros;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (side == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(sel, ones, acc, 2, 2, 0, sl, 0, s1);
      else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, sel, acc, 2, 2, 0, s1, 0, sl);
      for (int e = 0; e < 4; ++e) out[(side * 4 + g0) * 256 + (4 * (lane >> 4) + e) * 16 + (lane & 15)] = acc[e];
    }
}
int main() {
  float* d; static float h[8 * 256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int side = 0; side < 2; ++side)
    for (int g0 = 0; g0 < 4; ++g0) {
      printf("%s: ones in lane group %d only; log2(applied scale) per %s 0..15 (expected %d, +8 on odd):", side == 0 ? "A" : "B", g0, side == 0 ? "row" : "col", g0);
      for (int i = 0; i < 16; ++i) {
        const float v = (side == 0 ? h[(side * 4 + g0) * 256 + i * 16 + 0] : h[(side * 4 + g0) * 256 + 0 * 16 + i]) / 32.f;
        printf(" %g", v > 0 ? __builtin_log2f(v) : -999.f);
      }
      printf("\n");
    }
  return 0;
}
