// GPU probe (one wave): pins the semantics the fp16 + block-scaled-fp6 arithmetic of DESIGN.md section 9 item 0 relies on --
//   * v_cvt_scalef32_2xpk16_fp6_f32: 2 x 16 floats / scale -> 32 x e2m3, round to nearest even, the two sources INTERLEAVED (source 0
//     element i at 6-bit position 2 i, source 1 element i at 2 i + 1); the builtin must not be used as is (see quant_block)
//   * v_mfma_scale_f32_16x16x128_f8f6f4 with fp6 operands: lane (row = lane & 15, block = lane >> 4) holds k = 32 block .. + 31 and the
//     E8M0 scale of that block (value = element x 2^(scale - 127)); result layout as v_mfma_f32_16x16x32_*
//   * the whole scheme on K = 128 (two 32-k steps): 4 x v_mfma_f32_16x16x32_f16 + ONE fp6 instruction against the exact product
//   hipcc -O2 --offload-arch=gfx950 f16f6_probe.hip -o f16f6_probe && ./f16f6_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));

#ifndef NOPS
#define NOPS 0
#endif
// one 32-element block -> fp6 e2m3 + E8M0 scale (MX rule: scale exponent = floor(log2 max|v|) - 2, e2m3's largest exponent)
__device__ inline void quant_block(const float (&v)[32], i32x8& out, int& scale_e8m0) {
  float m = 0.f;
  for (int i = 0; i < 32; ++i) m = fmaxf(m, fabsf(v[i]));
  int e = m > 0.f ? (int)((__float_as_uint(m) >> 23) & 0xff) - 127 - 2 : -127;  // (denormal maxima: treated as 2^-127 blocks)
  e = max(e, -127);
  const float scale = __uint_as_float((unsigned)(e + 127) << 23);
  f32x16 a, b;
  for (int i = 0; i < 16; ++i) { a[i] = v[i]; b[i] = v[16 + i]; }
  // (inline asm with an early-clobber destination: hipcc (ROCm 7.2) allocates the builtin's destination over its SCALE operand -- seen:
  // v[50:55] <- v[34:49], v[18:33], v50 -- and the second half of the result is then converted with a clobbered scale)
  u32x6 q;
  asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(q) : "v"(a), "v"(b), "v"(scale));
  out = i32x8{(int)q[0], (int)q[1], (int)q[2], (int)q[3], (int)q[4], (int)q[5], 0, 0};
  scale_e8m0 = e + 127;
}

// A [16][K], B [16][K] row-major fp32 (K = 128); out: D_fp6 [16][16] (corrections-style product of the two quantised operands),
// D_mix [16][16] (the full scheme), raw fp6 words of A for the host to decode
__global__ void probe(const float* A, const float* B, float* d_fp6, float* d_mix, unsigned* a_words, int* a_scales, float* d_main) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  // ---- test 1: D = q6(A) q6(B)^T over K = 128
  float va[32], vb[32];
  for (int i = 0; i < 32; ++i) { va[i] = A[r * 128 + 32 * g + i]; vb[i] = B[r * 128 + 32 * g + i]; }
  i32x8 qa, qb;
  int sa, sb;
  quant_block(va, qa, sa);
  quant_block(vb, qb, sb);
  for (int i = 0; i < 6; ++i) a_words[lane * 6 + i] = (unsigned)qa[i];
  a_scales[lane] = sa;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(qa, qb, acc, 2, 2, 0, sa, 0, sb);
  for (int e = 0; e < 4; ++e) d_fp6[(4 * g + e) * 16 + r] = acc[e];  // D[m = 4 g + e][n = r]: rows from A, columns from B

  // ---- test 2: the scheme.  x = A row (pixels), w = B row (output channels); two 32-k steps s = 0, 1 use k = 64 s .. 64 s + 31 (the rest
  // of the arrays is ignored): main product fp16(x) fp16(w) over the 64 k, corrections K-concatenated: block g of the fp6 instruction =
  // (step g >> 1, g & 1 == 0: q6(x) with q6(w - f16 w);  g & 1 == 1: q6(x - f16 x) with q6(w))
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < 2; ++s) {
    f16x8 xa, wb;
    for (int i = 0; i < 8; ++i) { xa[i] = (_Float16)A[r * 128 + 64 * s + 8 * g + i]; wb[i] = (_Float16)B[r * 128 + 64 * s + 8 * g + i]; }
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa, wb, acc2, 0, 0, 0);
  }
  for (int e = 0; e < 4; ++e) d_main[(4 * g + e) * 16 + r] = acc2[e];
  {
    const int s = g >> 1, second = g & 1;
    float cx[32], cw[32];
    for (int i = 0; i < 32; ++i) {
      const float x = A[r * 128 + 64 * s + i], w = B[r * 128 + 64 * s + i];
      const float x2 = x - (float)(_Float16)x, w2 = w - (float)(_Float16)w;
      cx[i] = second ? x2 : x;
      cw[i] = second ? w : w2;
    }
    i32x8 qx, qw;
    int sx, sw;
    quant_block(cx, qx, sx);
    quant_block(cw, qw, sw);
    float part = 0.f, sumx = 0.f, sumw = 0.f;
    for (int i = 0; i < 32; ++i) { part += cx[i] * cw[i]; sumx += fabsf(cx[i]); sumw += fabsf(cw[i]); }
    d_main[256 + lane * 4 + 0] = part; d_main[256 + lane * 4 + 1] = sumx; d_main[256 + lane * 4 + 2] = sumw; d_main[256 + lane * 4 + 3] = (float)(sx * 1000 + sw);
    for (int i = 0; i < 6; ++i) { a_words[384 + lane * 12 + i] = (unsigned)qx[i]; a_words[384 + lane * 12 + 6 + i] = (unsigned)qw[i]; }
    f32x4 accc = {0.f, 0.f, 0.f, 0.f};
    accc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(qx, qw, accc, 2, 2, 0, sx, 0, sw);
    acc2 += accc;
  }
  for (int e = 0; e < 4; ++e) d_mix[(4 * g + e) * 16 + r] = acc2[e];
}

static double e2m3(unsigned bits) {
  const int s = (bits >> 5) & 1, e = (bits >> 3) & 3, m = bits & 7;
  const double v = e == 0 ? m / 8.0 : (1.0 + m / 8.0) * std::ldexp(1.0, e - 1);
  return s ? -v : v;
}
static double quant_host(double v, double scale) {  // RNE onto the e2m3 grid (ties to even mantissa)
  static std::vector<double> grid;
  if (grid.empty()) for (unsigned b = 0; b < 32; ++b) grid.push_back(e2m3(b));
  const double a = std::fabs(v) / scale;
  int best = 0;
  for (int i = 1; i < 32; ++i) {
    const double d = std::fabs(grid[i] - a), db = std::fabs(grid[best] - a);
    if (d < db || (d == db && (i & 1) == 0)) best = i;
  }
  return (v < 0 ? -1 : 1) * grid[best] * scale;
}

int main() {
  std::vector<float> A(16 * 128), B(16 * 128);
  srand(3);
  for (auto& v : A) v = (float)((rand() / (double)RAND_MAX - 0.5) * 4.0 * std::exp((rand() / (double)RAND_MAX - 0.5) * 3));
  for (auto& v : B) v = (float)((rand() / (double)RAND_MAX - 0.5) * 0.2);
  if (getenv("T2DATA")) {  // test 1 on the operand blocks test 2 builds: row r, block g = (step g >> 1; even: x | w - f16 w, odd: x - f16 x | w)
    std::vector<float> A2(16 * 128), B2(16 * 128);
    for (int r = 0; r < 16; ++r)
      for (int g = 0; g < 4; ++g)
        for (int i = 0; i < 32; ++i) {
          const float x = A[r * 128 + 64 * (g >> 1) + i], w = B[r * 128 + 64 * (g >> 1) + i];
          A2[r * 128 + 32 * g + i] = (g & 1) ? x - (float)(_Float16)x : x;
          B2[r * 128 + 32 * g + i] = (g & 1) ? w : w - (float)(_Float16)w;
        }
    A = A2; B = B2;
  }
  if (getenv("SMALL")) { for (auto& v : A) v *= 1.3e-4f; for (auto& v : B) v *= 0.7e-5f; }
  float *dA, *dB, *d1, *d2, *d3; unsigned* dw; int* ds; hipMalloc(&d3, 2048);
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&d1, 1024); hipMalloc(&d2, 1024); hipMalloc(&dw, 64 * 18 * 4); hipMalloc(&ds, 256);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, d1, d2, dw, ds, d3);
  std::vector<float> D1(256), D2(256), D3(512); std::vector<unsigned> W(64 * 18); std::vector<int> S(64);
  hipMemcpy(D1.data(), d1, 1024, hipMemcpyDeviceToHost); hipMemcpy(D2.data(), d2, 1024, hipMemcpyDeviceToHost); hipMemcpy(D3.data(), d3, 2048, hipMemcpyDeviceToHost);
  hipMemcpy(W.data(), dw, 64 * 18 * 4, hipMemcpyDeviceToHost); hipMemcpy(S.data(), ds, 256, hipMemcpyDeviceToHost);
  // (a) the conversion: decode the device's words against the host's RNE quantisation
  int bad_q = 0;
  for (int lane = 0; lane < 64; ++lane) {
    const int r = lane & 15, g = lane >> 4;
    const double scale = std::ldexp(1.0, S[lane] - 127);
    for (int i = 0; i < 32; ++i) {
      const int bit = 6 * (i < 16 ? 2 * i : 2 * (i - 16) + 1);  // element i of the first source at position 2 i, of the second at 2 i + 1
      unsigned long long two = W[lane * 6 + bit / 32] | ((unsigned long long)(bit / 32 + 1 < 6 ? W[lane * 6 + bit / 32 + 1] : 0) << 32);
      const unsigned bits = (unsigned)(two >> (bit % 32)) & 63;
      const double dev = e2m3(bits) * scale, host = quant_host(A[r * 128 + 32 * g + i], scale);
      if (dev != host) { if (bad_q < 5) printf("  quant mismatch lane %d elem %d: value %g device %g host %g (scale 2^%d)\n", lane, i, A[r * 128 + 32 * g + i], dev, host, S[lane] - 127); ++bad_q; }
    }
  }
  {  // where does element i land?  (lane 0: match host-quantised values against the decoded positions)
    const double scale = std::ldexp(1.0, S[0] - 127);
    printf("  lane 0: host element i -> device positions holding that value:");
    for (int i = 0; i < 32; ++i) {
      const double host = quant_host(A[i], scale);
      printf(" %d:[", i);
      for (int pos = 0; pos < 32; ++pos) {
        const int bit = 6 * pos;
        unsigned long long two = W[bit / 32] | ((unsigned long long)(bit / 32 + 1 < 6 ? W[bit / 32 + 1] : 0) << 32);
        if (e2m3((unsigned)(two >> (bit % 32)) & 63) * scale == host) printf("%d ", pos);
      }
      printf("]");
    }
    printf("\n");
  }
  printf("conversion (v_cvt_scalef32_2xpk16_fp6_f32: source 0 element i at position 2 i, source 1 element i at 2 i + 1; RNE): %d of 2048 elements differ from the host quantiser\n", bad_q);
  // (b) the fp6 MFMA: D1 against sum of the host-quantised products
  double e1 = 0, n1 = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      double ref = 0;
      for (int k = 0; k < 128; ++k) {
        auto sc = [&](const std::vector<float>& M, int row, int kk) {
          double mx = 0; for (int i = 0; i < 32; ++i) mx = std::max(mx, (double)std::fabs(M[row * 128 + (kk & ~31) + i]));
          return std::ldexp(1.0, (int)std::floor(std::log2(mx)) - 2);
        };
        ref += quant_host(A[m * 128 + k], sc(A, m, k)) * quant_host(B[n * 128 + k], sc(B, n, k));
      }
      e1 = std::max(e1, std::fabs(D1[m * 16 + n] - ref)); n1 = std::max(n1, std::fabs(ref));
    }
  printf("fp6 scaled MFMA, K = 128: max |device - host model| = %.3g (max |result| %.3g)\n", e1, n1);
  // (c) the scheme on the 2 x 32 k it covers
  double e2 = 0, ef = 0, n2 = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      double ref = 0, f16only = 0;
      for (int s = 0; s < 2; ++s)
        for (int i = 0; i < 32; ++i) {
          const double x = A[m * 128 + 64 * s + i], w = B[n * 128 + 64 * s + i];
          ref += x * w;
          f16only += (double)(float)(_Float16)(float)x * (double)(float)(_Float16)(float)w;
        }
      e2 += (D2[m * 16 + n] - ref) * (D2[m * 16 + n] - ref); ef += (f16only - ref) * (f16only - ref); n2 += ref * ref;
    }
  {  // host model of the scheme (quantiser as above)
    double em = 0, eh = 0, nn = 0;
    for (int m = 0; m < 16; ++m)
      for (int n = 0; n < 16; ++n) {
        double ref = 0, model = 0;
        for (int s = 0; s < 2; ++s) {
          double mx[4] = {0, 0, 0, 0};
          for (int i = 0; i < 32; ++i) {
            const float x = A[m * 128 + 64 * s + i], w = B[n * 128 + 64 * s + i];
            const float x2 = x - (float)(_Float16)x, w2 = w - (float)(_Float16)w;
            mx[0] = std::max(mx[0], (double)std::fabs(x)); mx[1] = std::max(mx[1], (double)std::fabs(w2));
            mx[2] = std::max(mx[2], (double)std::fabs(x2)); mx[3] = std::max(mx[3], (double)std::fabs(w));
          }
          double sc[4];
          for (int q = 0; q < 4; ++q) sc[q] = std::ldexp(1.0, (int)std::floor(std::log2(mx[q])) - 2);
          for (int i = 0; i < 32; ++i) {
            const float x = A[m * 128 + 64 * s + i], w = B[n * 128 + 64 * s + i];
            const float x1 = (float)(_Float16)x, w1 = (float)(_Float16)w;
            ref += (double)x * w;
            model += (double)x1 * w1 + quant_host(x, sc[0]) * quant_host(w - w1, sc[1]) + quant_host(x - x1, sc[2]) * quant_host(w, sc[3]);
          }
        }
        em += (model - ref) * (model - ref); eh += (D2[m * 16 + n] - model) * (D2[m * 16 + n] - model); nn += ref * ref;
        if (m == 0 && n < 3) {
          double mainh = 0;
          for (int s = 0; s < 2; ++s) for (int i = 0; i < 32; ++i) mainh += (double)(float)(_Float16)A[m * 128 + 64 * s + i] * (double)(float)(_Float16)B[n * 128 + 64 * s + i];
          printf("  D[0][%d]: exact %.8f  host model %.8f  device %.8f   main: host %.8f device %.8f   corrections: host %.3g device %.3g\n", n, ref, model, D2[m * 16 + n], mainh,
                 D3[m * 16 + n], model - mainh, D2[m * 16 + n] - D3[m * 16 + n]);
        }
      }
    printf("host model of the scheme vs exact: %.3g   device vs host model: %.3g (rms / rms result)\n", std::sqrt(em / nn), std::sqrt(eh / nn));
  }
  for (int lane : {0, 16, 17}) {
    printf("  lane %2d: test-1 A words %08x %08x %08x %08x %08x %08x scale %d | test-2 x words %08x %08x %08x %08x %08x %08x scale %d\n", lane, W[lane * 6], W[lane * 6 + 1], W[lane * 6 + 2],
           W[lane * 6 + 3], W[lane * 6 + 4], W[lane * 6 + 5], S[lane], W[384 + lane * 12], W[384 + lane * 12 + 1], W[384 + lane * 12 + 2], W[384 + lane * 12 + 3], W[384 + lane * 12 + 4],
           W[384 + lane * 12 + 5], ((int)D3[256 + lane * 4 + 3]) / 1000);
  }
  {  // decode test 2's operand words (element i of in1 at position 2 i, of in2 at 2 i + 1) and redo the MFMA on the host
    auto elem = [&](const unsigned* w6, int i) {
      const int pos = i < 16 ? 2 * i : 2 * (i - 16) + 1, bit = 6 * pos;
      unsigned long long two = w6[bit / 32] | ((unsigned long long)(bit / 32 + 1 < 6 ? w6[bit / 32 + 1] : 0) << 32);
      return e2m3((unsigned)(two >> (bit % 32)) & 63);
    };
    double worst = 0;
    for (int m = 0; m < 16; ++m)
      for (int n = 0; n < 16; ++n) {
        double acc = 0;
        for (int g = 0; g < 4; ++g) {
          const int la = 16 * g + m, lb = 16 * g + n;
          const double sa = std::ldexp(1.0, ((int)D3[256 + la * 4 + 3]) / 1000 - 127), sb = std::ldexp(1.0, ((int)D3[256 + lb * 4 + 3]) % 1000 - 127);
          for (int i = 0; i < 32; ++i) acc += elem(&W[384 + la * 12], i) * sa * elem(&W[384 + lb * 12 + 6], i) * sb;
        }
        worst = std::max(worst, std::fabs(acc - (D2[m * 16 + n] - D3[m * 16 + n])));
        if (m == 0 && n < 3) printf("  corrections D[0][%d]: host MFMA on the device's operand words %.4g, device %.4g\n", n, acc, D2[m * 16 + n] - D3[m * 16 + n]);
      }
    printf("  test 2 correction MFMA vs host MFMA on the same words: max diff %.3g\n", worst);
  }
  for (int m = 0; m < 3; ++m) {
    double dev_part = 0;
    for (int g = 0; g < 4; ++g) dev_part += D3[256 + (16 * g + m) * 4];
    printf("  diagonal D[%d][%d]: device corrections %.4g, device-side unquantised sum over the 4 lanes %.4g;", m, m, D2[m * 16 + m] - D3[m * 16 + m], dev_part);
    for (int g = 0; g < 4; ++g) printf("  lane g=%d: part %.3g sum|x| %.3g sum|w| %.3g scales %d", g, D3[256 + (16 * g + m) * 4], D3[256 + (16 * g + m) * 4 + 1], D3[256 + (16 * g + m) * 4 + 2], (int)D3[256 + (16 * g + m) * 4 + 3]);
    printf("\n");
  }
  printf("fp16 + fp6 x 2 over 64 k: rms error / rms result = %.3g   (fp16 product alone: %.3g)\n", std::sqrt(e2 / n2), std::sqrt(ef / n2));
  return 0;
}
