// Two consecutive 32-channel 3x3 convolutions of the FULL-RESOLUTION tail of BiDirectionalFusion as ONE kernel, tiles LDS-resident:
//
//   MODE_C2F   C2FModule, output_conv2_fusion (GatedFusionBlock, one input, upscale=False) + output_conv3
//              bi_directional_fusion_model.py:56-82 (GatedConvUnit), :116-146 (block), :171-180 (definitions), :203-204 (use)
//                 o     = conv3x3(relu(x); W1) + b1 + x                                   GateresConfUnit2.conv + skip_add
//                 f     = relu(LN(conv3x3(o; W2) + b2 + pre))                             fusion_conv.0-.2 over cat([o, c_feat]): ``pre`` = its coarse half
//                 y     = o * sigmoid(Wg f [+ bg])                                        fusion_conv.3 + gate
//                 last  = Wo y + bo                                                       out_conv (1x1)
//                 depth = w3 . last + b3                                                  output_conv3 (1x1 -> 1)
//   MODE_ENC   fusion_layers_1[0] + fusion_layers_2[0]  (SingleConvCNNLN, convs.py:58-72)   bi_directional_fusion_model.py:424-431
//                 f  = gelu(LN(conv3x3(x; W1) + b1 + pre))                                over cat([c, x]): ``pre`` = the coarse half
//                 y  = gelu(LN(conv3x3(cat([f, p1, p2]); W2) + b2))                       p1 / p2: the two depth maps at the level's size
//
// Unfused these are five resp. three launches that each stream a 32-channel full-resolution map (81 tiles x 384 x 512 x 128 B = 2 GB)
// in and out at ~3 TB/s with the matrix pipe idle half of the time (profiles/r04_bf16x3_layers_v2_zoe_4k_r32.csv: 7.4 + 5.4 ms per frame).
// Here a workgroup owns an 8 x 16 output tile: the 12 x 20 input window is staged once (fp32 -> bf16 hi / lo), the first conv is
// evaluated on the 10 x 18 window the second one needs (zero outside the image = the second conv's zero padding) and stays in LDS.
//
// Design: WEIGHT-STATIONARY, WAVE-SPECIALISED.  At 32 -> 32 channels a tap is ONE k = 32 MFMA step, so any scheme that streams
// weights through LDS spends more LDS bandwidth on them than on the pixels.  Instead waves 0-3 hold the first conv's 9 x 2 x (hi, lo)
// A-operand fragments in registers (144 VGPRs) for the whole (persistent) kernel and waves 4-7 the second conv's; the two groups
// form a two-stage pipeline over the workgroup's tile sequence -- stage 1 computes tile t while stage 2 finishes tile t - 1 from the
// other half of a double-buffered LDS tile -- with ONE barrier per tile.  A SIMD hosts one wave of each role (waves w, w + 4), so the
// matrix pipe sees one stage's MFMAs while the other stage's wave is in its VALU epilogue / loads.  Weights are the MFMA's A operand:
// an accumulator holds four consecutive CHANNELS of one pixel, a pixel's 32 channels sit in the four lanes (m16, g = 0..3), so
// LayerNorm, the gate GEMM, the product and out_conv run on registers (the K order of the second-stage GEMMs is permuted to the
// accumulator order at pack time: prv2_pack_chain32_weight) and the results leave as 16-byte stores.
// Arithmetic: the split products (hi*lo, lo*hi, hi*hi) and fp32 accumulation of the other bf16x3 kernels; ``o`` counts as hi + lo
// in the gate product (as ``mul`` does in conv3x3_gate.hip); LayerNorm two-pass (convs.py:25-27), GELU / sigmoid as in the bf16 store
// loops (common.h gelu_fast, sigmoid on v_exp_f32 / v_rcp_f32).  Not bit-identical to the unfused sequence (the LayerNorm row sums are
// reduced in another order): tests compare against fp64 / the unfused kernels with the mode's tolerance.
#include <cstdlib>

#include "igemm.h"

namespace prv2 {

namespace c32 {
constexpr int TH = 8, TW = 16;                   // output tile
constexpr int IN_H = TH + 4, IN_W = TW + 4;      // 12 x 20 input window
constexpr int MID_H = TH + 2, MID_W = TW + 2;    // 10 x 18 window of the first conv's output
constexpr int IN_PX = IN_H * IN_W, MID_PX = MID_H * MID_W;  // 240, 180
constexpr int PIX = 160;                         // bytes per pixel in LDS: [32 bf16 hi | 32 bf16 lo | 32 B pad] (conflict-free ds_read_b128 fragments)
constexpr int IN_BYTES = IN_PX * PIX;            // 38 400
constexpr int MID_BYTES = MID_PX * PIX;          // 28 800
constexpr int PRED_BYTES = MID_PX * 8;           // (p1, p2) per pixel of the mid window (MODE_ENC)
constexpr int NCONST = 9 * 32;                   // per-channel constants (floats): see ConstSlot
constexpr int OFF_IN = 0, OFF_MID = 2 * IN_BYTES, OFF_PRED = OFF_MID + 2 * MID_BYTES, OFF_CONST = OFF_PRED + 2 * PRED_BYTES,
              OFF_W1X1 = OFF_CONST + NCONST * 4, SMEM_BYTES = OFF_W1X1 + 3 * 4096;
static_assert(SMEM_BYTES <= 160 * 1024, "LDS budget");
constexpr int RUNS1_PER_WAVE = 3;    // first conv: 180 pixels -> 12 runs of 16 (the last one 4 valid)
constexpr int NPF = 8;                           // input window items per loader thread (240 px x 8 chunks / 256 threads = 7.5)
enum ConstSlot { C_B1 = 0, C_LN1W, C_LN1B, C_B2, C_BG, C_BO, C_W3, C_LN2W, C_LN2B };
constexpr int FRAG1x1 = 2 * 2 * 64;
}  // namespace c32

struct Chain32Params {
  const float* x;  // NHWC input, 32 channels
  int ldx;
  long long x_bstride;
  int N, H, W;
  const void* w1;  // fragment images (prv2_pack_chain32_weight)
  const void* w2;
  const void* wg;  // MODE_C2F: gate 1x1 (K permuted);  MODE_ENC: the tail tile of W2 (pred channels)
  const void* wo;  // MODE_C2F: out_conv 1x1 (K permuted)
  const float* consts;  // [9][32] fp32: b1, ln1 w, ln1 b, b2, bg, bo, w3, ln2 w, ln2 b (unused rows zero)
  float b3, eps;
  const float* pre;  // pre-LayerNorm addend (coarse half of the conv that has one), NHWC >= 32 channels
  int ld_pre;
  const float* p1;  // MODE_ENC: dense [N, H, W]
  const float* p2;
  float* y;  // NHWC output, 32 channels
  int ldy;
  long long y_bstride;
  float* depth;  // MODE_C2F: dense [N, H, W]
  int tiles_x, tiles_y;
  long long ntiles;
  long long* stamps;  // -DC32_STAMPS builds (tools/ab_chain32.sh): per workgroup and wave [compute, window store / -, barrier wait] cycles
};

__device__ __forceinline__ float c32_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f)); }

// 8 fp32 (two accumulator quads of a lane) -> bf16 hi / lo fragments
__device__ __forceinline__ void c32_split8(const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& lo) {
  bf16x4 h0, l0, h1, l1;
  split_bf16(a, h0, l0);
  split_bf16(b, h1, l1);
  hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
  lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ void c32_mma(f32x4& c, const bf16x8& wh, const bf16x8& wl, const bf16x8& xh, const bf16x8& xl) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, c, 0, 0, 0);
}

// sum over the four lanes (m16, g = 0..3) that hold one pixel's channels
__device__ __forceinline__ float c32_sum_g(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// channels-first LayerNorm over the 32 channels of a pixel held as (a, b) in its four lanes; two passes (convs.py:25-27)
__device__ __forceinline__ void c32_layernorm(f32x4& a, f32x4& b, const f32x4 w0, const f32x4 w1, const f32x4 b0, const f32x4 b1, float eps) {
  float s = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
  const float mean = c32_sum_g(s) * (1.0f / 32.0f);
  a -= mean;
  b -= mean;
  float q = ((a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w)) + ((b.x * b.x + b.y * b.y) + (b.z * b.z + b.w * b.w));
  const float rstd = 1.0f / sqrtf(c32_sum_g(q) * (1.0f / 32.0f) + eps);
  a = a * rstd * w0 + b0;
  b = b * rstd * w1 + b1;
}

__device__ __forceinline__ f32x4 c32_gelu4(f32x4 v) { return f32x4{gelu_fast(v.x), gelu_fast(v.y), gelu_fast(v.z), gelu_fast(v.w)}; }

template <int MODE>
__global__ void __launch_bounds__(512) chain32_kernel(const Chain32Params p) {
  using namespace c32;
  __shared__ __attribute__((aligned(1024))) char smem[SMEM_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15, g = lane >> 4;

  // ---- this workgroup's tile sequence: an XCD takes a contiguous range of tiles, its workgroups walk it side by side (neighbouring
  //      tiles share their halo rows in that XCD's L2) ----
  const int per = max((int)gridDim.x >> 3, 1), xcd = blockIdx.x & 7, jwg = blockIdx.x >> 3;
  const int ntiles = (int)p.ntiles, chunk = (ntiles + 7) / 8;  // (32-bit: 64-bit divisions cost hundreds of instructions per tile)
  const int t_begin = xcd * chunk + jwg, t_end = min((xcd + 1) * chunk, ntiles);
  const int K = t_begin < t_end ? (t_end - t_begin + per - 1) / per : 0;
  const int tiles_xy = p.tiles_x * p.tiles_y;
  auto tile_of = [&](int k, int& n, int& y0, int& x0) {
    const int t = t_begin + k * per;
    n = t / tiles_xy;
    const int r = t - n * tiles_xy, ty = r / p.tiles_x;
    y0 = ty * TH;
    x0 = (r - ty * p.tiles_x) * TW;
  };
  const int HW = p.H * p.W;

  // ---- constants and the 1x1 fragment images -> LDS (read at their use: they would cost ~80 registers per wave) ----
  float* const cst = reinterpret_cast<float*>(smem + OFF_CONST);
  if (tid < NCONST) cst[tid] = p.consts[tid];
  {
    const f32x4* src0 = reinterpret_cast<const f32x4*>(p.wg);
    const f32x4* src1 = reinterpret_cast<const f32x4*>(p.wo);
    f32x4* dst = reinterpret_cast<f32x4*>(smem + OFF_W1X1);
    if (tid < FRAG1x1 && src0) dst[tid] = src0[tid];
    if (tid >= 256 && tid < 256 + FRAG1x1 && src1) dst[tid] = src1[tid - 256];
  }
  auto cvec = [&](int slot, int blk) { return *reinterpret_cast<const f32x4*>(cst + slot * 32 + blk * 16 + 4 * g); };

  if (wave < 4) {
    // ============================== stage 1: input window -> first conv -> mid window ==============================
    bf16x8 wh[9][2], wl[9][2];
    {
      const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(p.w1);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          wh[t][b] = wsrc[((t * 2 + b) * 2 + 0) * 64 + lane];
          wl[t][b] = wsrc[((t * 2 + b) * 2 + 1) * 64 + lane];
        }
    }
    // (a "use" of every weight register: the compiler then waits for the loads HERE and its counted vmcnt waits inside the tile loop
    //  only see that loop's own loads -- otherwise every tap waited for "the weights", i.e. for everything older, incl. the prefetch)
#pragma unroll
    for (int t = 0; t < 9; ++t) asm volatile("" ::"v"(wh[t][0]), "v"(wh[t][1]), "v"(wl[t][0]), "v"(wl[t][1]));
    // loader roles: item i = it * 256 + tid -> (pixel i >> 3 of the 12 x 20 window, 4-channel chunk i & 7).  The byte offset of an item
    // relative to the window's first pixel does not depend on the tile: a tile whose window lies inside the image (90 % of them) is
    // fetched as uniform base + 32-bit offset without bounds arithmetic (the kernel is VALU-issue bound, not latency bound)
    f32x4 pf[NPF];
    const int ld4 = p.ldx * 4;
    auto load_window = [&](int k) {
      int n, y0, x0;
      tile_of(k, n, y0, x0);
      const float* img = p.x + (long long)n * p.x_bstride;
      // (the item offsets are recomputed per tile from an OPAQUE copy of the thread id: hoisted out of the tile loop -- as hipcc does when it
      //  can -- they occupy sixteen registers next to the weights and spill)
      int tid_o = tid;
      asm volatile("" : "+v"(tid_o));
      if (y0 >= 2 && x0 >= 2 && y0 + TH + 2 <= p.H && x0 + TW + 2 <= p.W) {  // block-uniform
        const char* base = reinterpret_cast<const char*>(img + ((y0 - 2) * p.W + (x0 - 2)) * p.ldx);
        const int px0 = tid_o >> 3, ch16 = (tid_o & 7) * 16;
#pragma unroll
        for (int it = 0; it < NPF; ++it)
          if (it < NPF - 1 || tid < 128) {
            const int px = it * 32 + px0, r = px / IN_W, c = px - r * IN_W;
            pf[it] = *reinterpret_cast<const f32x4*>(base + (unsigned)((r * p.W + c) * ld4 + ch16));
          }
        return;
      }
#pragma unroll
      for (int it = 0; it < NPF; ++it) {
        const int i = it * 256 + tid_o, px = i >> 3, ch = (i & 7) * 4;
        const int r = px / IN_W, c = px - r * IN_W;
        const int gy = y0 - 2 + r, gx = x0 - 2 + c;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (px < IN_PX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
          v = *reinterpret_cast<const f32x4*>(img + (gy * p.W + gx) * p.ldx + ch);
        pf[it] = v;
      }
    };
    auto store_window = [&](int buf) {
      int tid_o = tid;
      asm volatile("" : "+v"(tid_o));
#pragma unroll
      for (int it = 0; it < NPF; ++it) {
        const int i = it * 256 + tid_o, px = i >> 3, chunk = i & 7;
        if (px < IN_PX) {
          f32x4 v = pf[it];
          if (MODE == 0) v = relu4(v);  // GatedConvUnit: conv(activation(x))
          bf16x4 hi, lo;
          split_bf16(v, hi, lo);
          char* dst = smem + OFF_IN + buf * IN_BYTES + px * PIX + chunk * 8;
          *reinterpret_cast<bf16x4*>(dst) = hi;
          *reinterpret_cast<bf16x4*>(dst + 64) = lo;
        }
      }
    };
    // compute roles: run rr = 3 wave + a; lane (m16, g) = pixel q = 16 rr + m16 of the 10 x 18 mid window, k-slice / channel quad g
    // (the run loop is NOT unrolled and its addresses are recomputed per run: three runs' accumulators, operands and addresses next to
    //  the 144 weight registers and the prefetched window spill)

    // the epilogue's global operand of a run (MODE_C2F: the residual x, MODE_ENC: the coarse half ``pre``) is requested ONE RUN AHEAD -- the
    // first run of a tile during the last run of the previous one: waited for at its use it is an HBM round trip per run
    auto e_load = [&](int kk, int a, f32x4& e0, f32x4& e1) {
      e0 = e1 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kk >= K) return;
      int n, y0, x0;
      tile_of(kk, n, y0, x0);
      const int q = (wave * RUNS1_PER_WAVE + a) * 16 + m16, qq = min(q, MID_PX - 1);
      const int qr = qq / MID_W, qc = qq - qr * MID_W;
      const int gy = y0 - 1 + qr, gx = x0 - 1 + qc;
#ifdef C32_ABL_NOE1  // (timing ablations: results wrong)
      if (p.b3 != 12345.f) return;
#endif
      if (q < MID_PX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) {
        const float* src = MODE == 0 ? p.x + (long long)n * p.x_bstride + (gy * p.W + gx) * p.ldx
                                     : p.pre + (long long)n * HW * p.ld_pre + (gy * p.W + gx) * p.ld_pre;
        e0 = *reinterpret_cast<const f32x4*>(src + 4 * g);
        e1 = *reinterpret_cast<const f32x4*>(src + 16 + 4 * g);
      }
    };
    f32x4 e0, e1;
    if (K > 0) {
      load_window(0);
      store_window(0);
    }
    e_load(0, 0, e0, e1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef C32_STAMPS
    long long st_c = 0, st_s = 0, st_b = 0;
#endif
    for (int k = 0; k < K; ++k) {
#ifdef C32_STAMPS
      const long long s0 = clock64();
#endif
      const int buf = k & 1;
      int n, y0, x0;
      tile_of(k, n, y0, x0);
#ifndef C32_ABL_NOWIN
      if (k + 1 < K) load_window(k + 1);  // lands while this tile is convolved
#endif
      float2 pv = make_float2(0.f, 0.f);  // MODE_ENC: the two depth maps on the mid window (zero outside the image), requested in front of the runs
      if (MODE == 1 && tid < MID_PX) {
        const int r = tid / MID_W, c = tid - r * MID_W, gy = y0 - 1 + r, gx = x0 - 1 + c;
        if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) {
          const long long o = (long long)n * HW + (gy * p.W + gx);
          pv = make_float2(p.p1[o], p.p2[o]);
        }
      }
#pragma unroll 1
      for (int a = 0; a < RUNS1_PER_WAVE; ++a) {
        const int q = (wave * RUNS1_PER_WAVE + a) * 16 + m16, qq = min(q, MID_PX - 1);
        const int qr = qq / MID_W, qc = qq - qr * MID_W;
        const int in_off = OFF_IN + (qr * IN_W + qc) * PIX + g * 16;
        const int mid_off = q < MID_PX ? OFF_MID + q * PIX + g * 16 : -1;
        const int gy = y0 - 1 + qr, gx = x0 - 1 + qc;
        const bool valid = mid_off >= 0 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        f32x4 en0, en1;  // the next run's operand (of the next tile's first run behind this tile's last)
        if (a + 1 < RUNS1_PER_WAVE) e_load(k, a + 1, en0, en1);
        else e_load(k + 1, 0, en0, en1);
        f32x4 acc0 = cvec(C_B1, 0), acc1 = cvec(C_B1, 1);
        const char* const ib = smem + buf * IN_BYTES + in_off;
        // the next tap's fragments are requested in front of this tap's MFMAs (left alone hipcc reads each fragment right in front of its
        // first use and every tap waits two LDS round trips)
        bf16x8 xh[2], xl[2];
        xh[0] = *reinterpret_cast<const bf16x8*>(ib);
        xl[0] = *reinterpret_cast<const bf16x8*>(ib + 64);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          if (t + 1 < 9) {
            const int toff = (((t + 1) / 3) * IN_W + ((t + 1) % 3)) * PIX;
            xh[(t + 1) & 1] = *reinterpret_cast<const bf16x8*>(ib + toff);
            xl[(t + 1) & 1] = *reinterpret_cast<const bf16x8*>(ib + toff + 64);
          }
          __builtin_amdgcn_sched_barrier(0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[t][0], xl[t & 1], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[t][1], xl[t & 1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[t][0], xh[t & 1], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[t][1], xh[t & 1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[t][0], xh[t & 1], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[t][1], xh[t & 1], acc1, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        acc0 += e0;
        acc1 += e1;
        e0 = en0;
        e1 = en1;
        if (MODE == 1) {
          c32_layernorm(acc0, acc1, cvec(C_LN1W, 0), cvec(C_LN1W, 1), cvec(C_LN1B, 0), cvec(C_LN1B, 1), p.eps);
          acc0 = c32_gelu4(acc0);
          acc1 = c32_gelu4(acc1);
        }
        acc0 = zero_unless(acc0, valid);  // outside the image: the second conv's zero padding
        acc1 = zero_unless(acc1, valid);
        bf16x8 hi, lo;
        c32_split8(acc0, acc1, hi, lo);
        if (mid_off >= 0) {
          char* dst = smem + buf * MID_BYTES + mid_off;
          *reinterpret_cast<bf16x8*>(dst) = hi;
          *reinterpret_cast<bf16x8*>(dst + 64) = lo;
        }
      }
      if (MODE == 1 && tid < MID_PX) *reinterpret_cast<float2*>(smem + OFF_PRED + buf * PRED_BYTES + tid * 8) = pv;
#ifdef C32_STAMPS
      const long long s1 = clock64();
#endif
      if (k + 1 < K) store_window(buf ^ 1);
#ifdef C32_STAMPS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const long long s2 = clock64();
#endif
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef C32_STAMPS
      const long long s3 = clock64();
      st_c += s1 - s0; st_s += s2 - s1; st_b += s3 - s2;
#endif
    }
#ifdef C32_STAMPS
    if (p.stamps && lane == 0) {
      long long* o = p.stamps + ((long long)blockIdx.x * 8 + wave) * 4;
      o[0] = st_c; o[1] = st_s; o[2] = st_b; o[3] = K;
    }
#endif
  } else {
    // ============================== stage 2: mid window -> second conv -> epilogue -> HBM ==============================
    const int w2i = wave - 4;
    bf16x8 wh[9][2], wl[9][2];
    {
      const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(p.w2);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          wh[t][b] = wsrc[((t * 2 + b) * 2 + 0) * 64 + lane];
          wl[t][b] = wsrc[((t * 2 + b) * 2 + 1) * 64 + lane];
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) asm volatile("" ::"v"(wh[t][0]), "v"(wh[t][1]), "v"(wl[t][0]), "v"(wl[t][1]));
    auto w1x1 = [&](int which, int blk, int hl) {
      return *reinterpret_cast<const bf16x8*>(smem + OFF_W1X1 + which * 4096 + ((blk * 2 + hl) * 64 + lane) * 16);
    };
    // MODE_ENC tail step: lane (m16, g) supplies k-slots 8 g + j = (tap 4 g + (j >> 1), map j & 1); taps >= 9 are zero slots
    int pred_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int tap = 4 * g + j;
      pred_off[j] = tap < 9 ? ((tap / 3) * MID_W + (tap % 3)) * 8 : -1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef C32_STAMPS
    long long st_c = 0, st_b = 0;
#endif
    // the pre-LayerNorm addend of this wave's two rows (MODE_C2F), requested ONE TILE AHEAD: in the epilogue it would be a full HBM round
    // trip in the middle of a serial chain (the rows are this wave's only outstanding loads: nothing else to wait behind)
    f32x4 pe[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
    auto load_pre = [&](int kk) {
      if (MODE != 0 || !p.pre || kk >= K) return;
      int n, y0, x0;
      tile_of(kk, n, y0, x0);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int gy = y0 + 2 * w2i + a, gx = x0 + m16;
#ifdef C32_ABL_NOE2
        if (p.b3 != 12345.f) continue;
#endif
        if (gy < p.H && gx < p.W) {
          const float* src = p.pre + (long long)n * HW * p.ld_pre + (gy * p.W + gx) * p.ld_pre;
          pe[a][0] = *reinterpret_cast<const f32x4*>(src + 4 * g);
          pe[a][1] = *reinterpret_cast<const f32x4*>(src + 16 + 4 * g);
        }
      }
    };
    load_pre(0);
    for (int k = 0; k <= K; ++k) {
#ifdef C32_STAMPS
      const long long s0 = clock64();
#endif
      if (k > 0) {
        const int buf = (k - 1) & 1;
        int n, y0, x0;
        tile_of(k - 1, n, y0, x0);
        // the wave's two rows TOGETHER: two independent accumulator pairs through the taps (a row's next fragments are in flight under the
        // other row's MFMAs) and two independent epilogue chains (LayerNorm reductions, gate GEMM, sigmoid, out_conv) for the scheduler
        const int gx = x0 + m16;
        const char* const mb0 = smem + OFF_MID + buf * MID_BYTES + ((2 * w2i) * MID_W + m16) * PIX + g * 16;
        f32x4 acc[2][2];
        bf16x8 xh[2], xl[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          acc[a][0] = cvec(C_B2, 0);
          acc[a][1] = cvec(C_B2, 1);
          xh[a] = *reinterpret_cast<const bf16x8*>(mb0 + a * MID_W * PIX);
          xl[a] = *reinterpret_cast<const bf16x8*>(mb0 + a * MID_W * PIX + 64);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            __builtin_amdgcn_sched_barrier(0);
            acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[t][0], xl[a], acc[a][0], 0, 0, 0);
            acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[t][1], xl[a], acc[a][1], 0, 0, 0);
            acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[t][0], xh[a], acc[a][0], 0, 0, 0);
            acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[t][1], xh[a], acc[a][1], 0, 0, 0);
            acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[t][0], xh[a], acc[a][0], 0, 0, 0);
            acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[t][1], xh[a], acc[a][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < 9) {  // this row's next fragments: in flight under the other row's six MFMAs
              const int toff = (((t + 1) / 3 + a) * MID_W + ((t + 1) % 3)) * PIX;
              xh[a] = *reinterpret_cast<const bf16x8*>(mb0 + toff);
              xl[a] = *reinterpret_cast<const bf16x8*>(mb0 + toff + 64);
            }
          }
        }
        if (MODE == 1) {
          // the two depth-map channels of cat([f, p1, p2]): 9 taps x 2 maps as ONE k = 32 step
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const char* const pb = smem + OFF_PRED + buf * PRED_BYTES + ((2 * w2i + a) * MID_W + m16) * 8;
            const float2 v0 = pred_off[0] >= 0 ? *reinterpret_cast<const float2*>(pb + pred_off[0]) : make_float2(0.f, 0.f);
            const float2 v1 = pred_off[1] >= 0 ? *reinterpret_cast<const float2*>(pb + pred_off[1]) : make_float2(0.f, 0.f);
            const float2 v2 = pred_off[2] >= 0 ? *reinterpret_cast<const float2*>(pb + pred_off[2]) : make_float2(0.f, 0.f);
            const float2 v3 = pred_off[3] >= 0 ? *reinterpret_cast<const float2*>(pb + pred_off[3]) : make_float2(0.f, 0.f);
            bf16x8 th, tl;
            c32_split8(f32x4{v0.x, v0.y, v1.x, v1.y}, f32x4{v2.x, v2.y, v3.x, v3.y}, th, tl);
            c32_mma(acc[a][0], w1x1(0, 0, 0), w1x1(0, 0, 1), th, tl);
            c32_mma(acc[a][1], w1x1(0, 1, 0), w1x1(0, 1, 1), th, tl);
          }
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            c32_layernorm(acc[a][0], acc[a][1], cvec(C_LN2W, 0), cvec(C_LN2W, 1), cvec(C_LN2B, 0), cvec(C_LN2B, 1), p.eps);
            acc[a][0] = c32_gelu4(acc[a][0]);
            acc[a][1] = c32_gelu4(acc[a][1]);
          }
        } else {
          f32x4 gt[2][2];
          bf16x8 fh[2], fl[2];
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            acc[a][0] += pe[a][0];
            acc[a][1] += pe[a][1];
            c32_layernorm(acc[a][0], acc[a][1], cvec(C_LN1W, 0), cvec(C_LN1W, 1), cvec(C_LN1B, 0), cvec(C_LN1B, 1), p.eps);
            c32_split8(relu4(acc[a][0]), relu4(acc[a][1]), fh[a], fl[a]);
            gt[a][0] = cvec(C_BG, 0);
            gt[a][1] = cvec(C_BG, 1);
          }
          load_pre(k);  // (the next tile's rows: the registers are free again)
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            c32_mma(gt[a][0], w1x1(0, 0, 0), w1x1(0, 0, 1), fh[a], fl[a]);
            c32_mma(gt[a][1], w1x1(0, 1, 0), w1x1(0, 1, 1), fh[a], fl[a]);
          }
          bf16x8 yh[2], yl[2];
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            // o = hi + lo of the centre pixel (what the matrix pipe saw of it); element j of the fragment = channel 4 g + j (j < 4), 16 + 4 g + j - 4
            // (the centre tap's fragments = this lane's own channels of ``o``; read again rather than kept across the taps)
            typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
            const char* const cb = mb0 + ((a + 1) * MID_W + 1) * PIX;
            const u16x8 chu = __builtin_bit_cast(u16x8, *reinterpret_cast<const bf16x8*>(cb)), clu = __builtin_bit_cast(u16x8, *reinterpret_cast<const bf16x8*>(cb + 64));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float o0 = __builtin_bit_cast(float, (unsigned)chu[e] << 16) + __builtin_bit_cast(float, (unsigned)clu[e] << 16);
              const float o1 = __builtin_bit_cast(float, (unsigned)chu[4 + e] << 16) + __builtin_bit_cast(float, (unsigned)clu[4 + e] << 16);
              gt[a][0][e] = o0 * c32_sigmoid(gt[a][0][e]);
              gt[a][1][e] = o1 * c32_sigmoid(gt[a][1][e]);
            }
            c32_split8(gt[a][0], gt[a][1], yh[a], yl[a]);
            acc[a][0] = cvec(C_BO, 0);
            acc[a][1] = cvec(C_BO, 1);
          }
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            c32_mma(acc[a][0], w1x1(1, 0, 0), w1x1(1, 0, 1), yh[a], yl[a]);
            c32_mma(acc[a][1], w1x1(1, 1, 0), w1x1(1, 1, 1), yh[a], yl[a]);
          }
          const f32x4 w30 = cvec(C_W3, 0), w31 = cvec(C_W3, 1);
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const int gy = y0 + 2 * w2i + a;
            float d = ((acc[a][0].x * w30.x + acc[a][0].y * w30.y) + (acc[a][0].z * w30.z + acc[a][0].w * w30.w)) +
                      ((acc[a][1].x * w31.x + acc[a][1].y * w31.y) + (acc[a][1].z * w31.z + acc[a][1].w * w31.w));
            d = c32_sum_g(d) + p.b3;
            if (gy < p.H && gx < p.W && g == 0 && p.depth) {
              float* dd = p.depth + (long long)n * HW + (gy * p.W + gx);
              asm volatile("global_store_dword %0, %1, off\n\ts_nop 0" ::"v"(dd), "v"(d) : "memory");
            }
          }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int gy = y0 + 2 * w2i + a;
#ifndef C32_ABL_NOSTORE
          if (gy < p.H && gx < p.W) {
#else
          if (gy < p.H && gx < p.W && p.b3 == 12345.f) {
#endif
            // (inline asm: a store the compiler knows about makes its counted waits in the next tile's taps wait for the store's acknowledgement)
            float* dst = p.y + (long long)n * p.y_bstride + (gy * p.W + gx) * p.ldy + 4 * g;
            asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:64\n\ts_nop 1" ::"v"(dst), "v"(acc[a][0]), "v"(acc[a][1]) : "memory");
          }
        }
      }
#ifdef C32_STAMPS
      const long long s1 = clock64();
#endif
      if (k < K) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef C32_STAMPS
      const long long s2 = clock64();
      st_c += s1 - s0; st_b += s2 - s1;
#endif
    }
#ifdef C32_STAMPS
    if (p.stamps && lane == 0) {
      long long* o = p.stamps + ((long long)blockIdx.x * 8 + wave) * 4;
      o[0] = st_c; o[1] = 0; o[2] = st_b; o[3] = K;
    }
#endif
  }
}

// ---- weight images ---------------------------------------------------------------------------------------------------------------
// fragment (tap, blk, hl, lane = (m16, g)) = bf16 hi (hl = 0) / lo (hl = 1) of W[16 blk + m16][kmap(8 g + j)][tap], j = 0..7
//   kind 0: kmap(s) = s                                        3x3 / 1x1 whose B operand comes from an fp32 NHWC row (first conv)
//   kind 1: kmap(8 g + j) = j < 4 ? 4 g + j : 16 + 4 g + j - 4  B operand = a lane's own accumulator channels (second conv, gate, out_conv)
//   kind 2: the pred tail of a 3x3 over cin = 34: slot s -> (tap s >> 1, channel 32 + (s & 1)), s < 18; one "tap"
__global__ void chain32_pack_kernel(const float* __restrict__ w, int cin_total, int taps, int kind, unsigned short* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one bf16 element
  const int ntap = kind == 2 ? 1 : taps;
  if (i >= ntap * 2 * 2 * 64 * 8) return;
  const int j = i & 7, lane = (i >> 3) & 63, hl = (i >> 9) & 1, blk = (i >> 10) & 1, tap = i >> 11;
  const int m16 = lane & 15, g = lane >> 4, s = 8 * g + j, o = 16 * blk + m16;
  float v = 0.f;
  if (kind == 2) {
    if (s < 18) v = w[((long long)o * cin_total + 32 + (s & 1)) * taps + (s >> 1)];
  } else {
    const int c = kind == 0 ? s : (j < 4 ? 4 * g + j : 16 + 4 * g + j - 4);
    v = w[((long long)o * cin_total + c) * taps + tap];
  }
  const __bf16 hi = (__bf16)v;
  const __bf16 r = hl == 0 ? hi : (__bf16)(v - (float)hi);
  out[i] = __builtin_bit_cast(unsigned short, r);
}

}  // namespace prv2

using namespace prv2;

static inline bool c32_al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

extern "C" int64_t prv2_chain32_weight_bytes(int32_t kind, int32_t taps) { return (int64_t)(kind == 2 ? 1 : taps) * 2 * 2 * 64 * 16; }

extern "C" int prv2_pack_chain32_weight(const float* w_src, int32_t cin_total, int32_t taps, int32_t kind, void* w_packed, void* stream) {
  PRV2_REQUIRE(w_src && w_packed && c32_al16(w_packed), "pack_chain32_weight: null / unaligned pointer");
  PRV2_REQUIRE((taps == 9 || taps == 1) && kind >= 0 && kind <= 2 && cin_total >= (kind == 2 ? 34 : 32) && (kind != 2 || taps == 9),
               "pack_chain32_weight: taps 9 | 1, kind 0..2, 32 output x >= 32 (kind 2: 34) input channels (taps=%d kind=%d cin=%d)", taps, kind, cin_total);
  const int n = (kind == 2 ? 1 : taps) * 2 * 2 * 64 * 8;
  hipLaunchKernelGGL(chain32_pack_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w_src, cin_total, taps, kind,
                     reinterpret_cast<unsigned short*>(w_packed));
  PRV2_LAUNCH_CHECK("pack_chain32_weight");
  return 0;
}

static int chain32_launch(int mode, const prv2_chain32_desc* d, void* stream) {
  PRV2_REQUIRE(d && d->x && d->y && d->w1 && d->w2 && d->consts, "chain32: null pointer");
  PRV2_REQUIRE(d->n > 0 && d->h >= 1 && d->w >= 1 && d->ldx >= 32 && d->ldx % 4 == 0 && d->ldy >= 32 && d->ldy % 4 == 0 && c32_al16(d->x) && c32_al16(d->y) &&
                   c32_al16(d->w1) && c32_al16(d->w2) && d->x_bstride % 4 == 0 && d->y_bstride % 4 == 0,
               "chain32: 32-channel NHWC rows, 16-byte aligned (ldx=%d ldy=%d)", d->ldx, d->ldy);
  PRV2_REQUIRE(!d->pre || (d->ld_pre >= 32 && d->ld_pre % 4 == 0 && c32_al16(d->pre)), "chain32: pre rows must be 16-byte aligned (ld_pre=%d)", d->ld_pre);
  if (mode == 0) PRV2_REQUIRE(d->wg && d->wo && c32_al16(d->wg) && c32_al16(d->wo), "chain32_c2f: gate / out_conv fragment images missing");
  else PRV2_REQUIRE(d->wg && c32_al16(d->wg) && d->p1 && d->p2 && d->pre, "chain32_enc: tail fragment image / depth maps / pre missing");
  PRV2_REQUIRE((long long)d->h * d->w * (d->ldx > d->ldy ? d->ldx : d->ldy) < (1LL << 31), "chain32: image too large");
  Chain32Params p = {};
  p.x = d->x; p.ldx = d->ldx; p.x_bstride = d->x_bstride ? d->x_bstride : (long long)d->h * d->w * d->ldx;
  p.N = d->n; p.H = d->h; p.W = d->w;
  p.w1 = d->w1; p.w2 = d->w2; p.wg = d->wg; p.wo = d->wo; p.consts = d->consts; p.b3 = d->b3; p.eps = d->ln_eps;
  p.pre = d->pre; p.ld_pre = d->ld_pre; p.p1 = d->p1; p.p2 = d->p2;
  p.y = d->y; p.ldy = d->ldy; p.y_bstride = d->y_bstride ? d->y_bstride : (long long)d->h * d->w * d->ldy;
  p.depth = d->depth;
  p.tiles_x = (int)cdiv(d->w, c32::TW); p.tiles_y = (int)cdiv(d->h, c32::TH);
  p.ntiles = (long long)d->n * p.tiles_x * p.tiles_y;
  // persistent: one workgroup per CU (8 XCDs x 32), fewer when the tiles do not fill them
  static const int wgs = getenv("PRV2_CHAIN32_WGS") ? atoi(getenv("PRV2_CHAIN32_WGS")) : 256;  // A/B switch
  long long grid = (p.ntiles + 7) / 8;
  grid = (grid > wgs / 8 ? wgs / 8 : grid) * 8;
#ifdef C32_STAMPS
  p.stamps = getenv("PRV2_C32_STAMPS") ? reinterpret_cast<long long*>(strtoull(getenv("PRV2_C32_STAMPS"), nullptr, 0)) : nullptr;
#endif
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL((chain32_kernel<0>), dim3((unsigned)grid), dim3(512), 0, s, p);
  else hipLaunchKernelGGL((chain32_kernel<1>), dim3((unsigned)grid), dim3(512), 0, s, p);
  set_kernel(mode == 0 ? "chain32_c2f_kernel" : "chain32_enc_kernel", 32, PRV2_PREC_BF16X3);
  PRV2_LAUNCH_CHECK("chain32");
  return 0;
}

extern "C" int prv2_chain32_c2f(const prv2_chain32_desc* d, void* stream) { return chain32_launch(0, d, stream); }
extern "C" int prv2_chain32_enc(const prv2_chain32_desc* d, void* stream) { return chain32_launch(1, d, stream); }
