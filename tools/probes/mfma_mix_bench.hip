// GPU probe: what the MFMA pipes deliver, at the clock the chip holds, for two ways of computing an fp32-grade product
//   bf16x3     x = h + l (bf16): h*h' + h*l' + l*h' -- three v_mfma_f32_16x16x32_bf16 per 32 k            (the shipped arithmetic)
//   f16+f6x2   fp16(x)*fp16(w) on v_mfma_f32_16x16x32_f16 + the two first-order corrections  x*(w - fp16 w), (x - fp16 x)*w
//              K-concatenated on the block-scaled fp6 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, e2m3): one fp6 instruction covers the
//              corrections of TWO 32-k steps -- 1.5 instructions per 32 k   (accuracy: tools/studies/split_arith_study.py)
// Wave tile = 4 x 4 accumulators (the conv kernels' tile), operands in registers (4 + 4 fragments per k-step, two alternating sets so
// that consecutive MFMAs see other operand bits), 2 waves per SIMD, every CU busy; random operand bits vs all zeros.
//   hipcc -O3 --offload-arch=gfx950 mfma_mix_bench.hip -o mfma_mix_bench && ./mfma_mix_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int SETS = 2;  // operand sets alternating per k-step

template <int MODE>  // 0: bf16x3, 1: f16 + fp6 corrections, 2: f16 alone, 3: bf16 alone, 4: f16 + one fp6 instruction PER k-step (half of its K unused),
                     // 5: f16 + fp8 (e4m3, block-scaled) corrections, one instruction per two k-steps
__global__ void __launch_bounds__(512, 2) k(const i32x4* __restrict__ src, float* __restrict__ out, int iters, long long* clk) {
  const int lane = threadIdx.x & 63;
  // operands: per set 4 A + 4 B fragments, hi / lo (16 B per lane each); fp6: 4 A + 4 B fragments of 24 B (+ 8 B unused) per TWO k-steps
  i32x4 ah[SETS][4], al[SETS][4], bh[SETS][4], bl[SETS][4];
  i32x8 a6[4], b6[4];
  const i32x4* p = src + (blockIdx.x & 7) * 4096 + lane;
#pragma unroll
  for (int s = 0; s < SETS; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[s][i] = p[((s * 4 + i) * 4 + 0) * 64];
      al[s][i] = p[((s * 4 + i) * 4 + 1) * 64];
      bh[s][i] = p[((s * 4 + i) * 4 + 2) * 64];
      bl[s][i] = p[((s * 4 + i) * 4 + 3) * 64];
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const i32x4 u0 = p[(32 + i * 4 + 0) * 64], u1 = p[(32 + i * 4 + 1) * 64], v0 = p[(32 + i * 4 + 2) * 64], v1 = p[(32 + i * 4 + 3) * 64];
    a6[i] = i32x8{u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, 0, 0};
    b6[i] = i32x8{v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, 0, 0};
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int scale = 0x7f7f7f7f;  // E8M0 127 = 2^0 in every byte
  long long t0 = 0, r0 = 0;
  if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < SETS; ++s) {  // one 32-k step per set
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (MODE == 0) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al[s][i]), __builtin_bit_cast(bf16x8, bh[s][j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[s][i]), __builtin_bit_cast(bf16x8, bl[s][j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[s][i]), __builtin_bit_cast(bf16x8, bh[s][j]), acc[i][j], 0, 0, 0);
          } else if constexpr (MODE == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[s][i]), __builtin_bit_cast(bf16x8, bh[s][j]), acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[s][i]), __builtin_bit_cast(f16x8, bh[s][j]), acc[i][j], 0, 0, 0);
          }
        }
    }
    if constexpr (MODE == 1 || MODE == 4 || MODE == 5) {  // the corrections of the two k-steps above: K = 128 = 2 steps x (32 + 32)
#pragma unroll
      for (int rep = 0; rep < (MODE == 4 ? 2 : 1); ++rep)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if constexpr (MODE == 5) {
              i32x8 a8 = a6[i], b8 = b6[j];
              a8[6] = a6[(i + 1) & 3][0]; a8[7] = a6[(i + 1) & 3][1]; b8[6] = b6[(j + 1) & 3][0]; b8[7] = b6[(j + 1) & 3][1];
              acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i][j], 0, 0, 0, scale, 0, scale);
            } else {
              acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6[i], b6[j], acc[i][j], 2, 2, 0, scale, 0, scale);
            }
          }
    }
    // rotate the operand registers so that no MFMA repeats its predecessor's operands from one iteration to the next (cheap VALU)
    if ((it & 15) == 15) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ah[0][i].x ^= it; bh[0][i].y ^= it; a6[i][1] ^= it; b6[i][2] ^= it;
      }
    }
  }
  if (threadIdx.x == 0) {
    clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
    clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += acc[i][j];
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = sum.x + sum.y + sum.z + sum.w;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int blocks = 256, iters = argc > 1 ? atoi(argv[1]) : 20000, reps = argc > 2 ? atoi(argv[2]) : 3;  // (short kernels launched back to back: iters 1500 ~ 1.2 ms)
  const size_t nsrc = 8 * 4096 * 4;  // ints
  std::vector<int> h(nsrc);
  i32x4* src; float* out; long long* clk;
  CK(hipMalloc(&src, nsrc * 4)); CK(hipMalloc(&out, (size_t)blocks * 512 * 4)); CK(hipMalloc(&clk, blocks * 16));
  const char* names[6] = {"bf16x3 (3 bf16 MFMAs per 32 k)", "f16 + fp6 x 2 corrections (1.5 per 32 k)", "f16 alone (1 per 32 k)", "bf16 alone (1 per 32 k)",
                          "f16 + fp6, one fp6 instruction per k-step (2)", "f16 + fp8 e4m3 scaled corrections (1.5 instr)"};
  for (int zero = 0; zero < 2; ++zero) {
    srand(1);
    for (size_t i = 0; i < nsrc; ++i) {
      // random finite 16-bit floats in both halves (exponent field kept in the middle of the range: |v| ~ 2^-3 .. 2^3 for bf16 and fp16 alike
      // is not needed for timing; any finite pattern toggles the multipliers): random mantissas, small exponents
      unsigned lo = (rand() & 0x83ff) | 0x3800, hi = (rand() & 0x83ff) | 0x3800;
      h[i] = zero ? 0 : (int)((hi << 16) | lo);
    }
    CK(hipMemcpy(src, h.data(), nsrc * 4, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 6; ++mode) {
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float ms = 0.f;
      for (int rep = 0; rep < reps; ++rep) {  // the last repetition counts (the chip has settled)
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, src, out, iters, clk);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, src, out, iters, clk);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, src, out, iters, clk);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, src, out, iters, clk);
        if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(512), 0, 0, src, out, iters, clk);
        if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(512), 0, 0, src, out, iters, clk);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      std::vector<long long> c(blocks * 2);
      CK(hipMemcpy(c.data(), clk, blocks * 16, hipMemcpyDeviceToHost));
      double ghz = 0;
      for (int b = 0; b < blocks; ++b) ghz += (double)c[2 * b] / (double)c[2 * b + 1] * 0.1;
      ghz /= blocks;
      // algorithmic work: 2 k-steps of 32 per iteration, 16 accumulators of 16 x 16 per wave, 8 waves per block
      const double flop = 2.0 * 16 * 16 * 32 * 2 /*k-steps*/ * 16 * 8.0 * blocks * (double)iters;
      printf("%s  %-48s %8.2f ms   %7.1f fp32-grade TFLOP/s   in-kernel clock %.2f GHz\n", zero ? "zeros " : "random", names[mode], ms, flop / ms / 1e9, ghz);
    }
  }
  return 0;
}
