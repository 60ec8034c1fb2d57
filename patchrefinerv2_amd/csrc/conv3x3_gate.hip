// GatedConvUnit tail in ONE kernel (estimator/models/blocks/bi_directional_fusion_model.py:44-51, 70-80):
//
//   fused = act(LayerNorm_channels(conv3x3(x) + bias))          fusion_conv.0 (512 -> 256) . LayerNorm . ReLU
//   y     = mul * sigmoid(conv1x1(fused) + gate_bias) (+ res)   fusion_conv.3 (256 -> 256) . Sigmoid; out * gate (+ xs[0])
//
// Unfused this is the dominant 3x3 conv writing 256 channels, a LayerNorm pass over them (the fused-LN epilogue of
// conv3x3_m16.hip needs the whole channel row in one workgroup: Cout <= 128), and a 1x1 GEMM that re-reads them together with
// ``mul`` -- 9.2 KB of HBM traffic per pixel against 4.8 KB here, and the gate GEMM (HBM-bound at 135 TFLOP/s as a kernel of
// its own) becomes 8 more MFMA steps on a tile that is already on chip.
//
// Main loop = conv3x3_m16.hip's pipeline (halo slab staged once per 32 input channels and reused by the nine taps, weight
// tile of every tap by LDS-DMA into three rotating buffers, one barrier per tap, counted waits) with a workgroup tile of
// 8 x 16 pixels x ALL 256 output channels: 8 waves = 2 (rows 0-3 / 4-7) x 4 (64-channel quarters), a wave = 4 pixel runs of
// 16 x 4 columns of 16 = the same 4 x 4 accumulators of v_mfma_f32_16x16x32_bf16 and the same 16 ds_read_b128 per 48 MFMAs.
// LDS: 2 halo buffers of 10 x 18 pixels x 160 B (57.6 KB) + 3 weight buffers of 256 x 128 B (96 KB).
//
// Epilogue: C tile (conv + bias, fp32, 128 x 260 floats) -> LDS; two-pass row statistics (mean, centred variance:
// convs.py:25-27), four threads per pixel; then, per wave, the gate GEMM [64 pixels x 256] x [256 x 64]: A fragments are read
// from the C tile, normalised, activated and split to bf16 hi / lo on the fly; B fragments come straight from global memory
// (L2-resident 256 KB, packed fragment-major by prv2_pack_gate_weight: one 1 KB coalesced load per fragment); same split
// products in the same order as the stand-alone GEMM (lo*hi, hi*lo, hi*hi per 32-channel slab).  The gate accumulators go
// back through the C tile and leave through the common store loop (sigmoid, * mul, + res; 1 KB row stores).
#include <cstdlib>
#include <type_traits>

#include "igemm.h"

// Pre-split activations ("X2", round 3).  The halo stream costs the main loop 6-8 % (ablation): every fp32 element is ReLU'd, rounded
// to bf16 hi, subtracted, rounded to bf16 lo and written to LDS as two 8-byte halves.  A tensor whose ONLY consumers are these
// kernels is therefore stored by its producer in the operand format itself: per 8 channels [8 x bf16 hi | 8 x bf16 lo] = 32 bytes,
// the bytes of 8 floats -- same pixel stride, same buffers, channel slices at multiples of 8 stay valid.  hi = RNE bf16 of v,
// lo = RNE bf16 of v - hi (split_bf16): exactly what the consumer's loader computes from fp32, so a layer's result does not depend on
// which format its input arrived in.  The loader then only copies (two 16-byte loads -> two ds_write_b128 per 8 channels); operands of
// the final stage read in this format are hi + lo (the gate's ``mul`` is ALWAYS taken as hi + lo, whatever format it arrives in, for
// the same reason).  In this file: x / mul of the gate kernel (the [out | coarse ROI] concat of a GatedConvUnit), y of the plain
// 256-column conv (GatedConvUnit.conv writes ``out`` into that concat); prv2_roi_align_x2 writes the other half.
namespace prv2 {

namespace g256 {
constexpr int BN = 256, TH = 8, TW = 16, HW_ = TW + 2;
constexpr int HALO = (TH + 2) * HW_;            // 180 halo pixels
constexpr int A_IT = (HALO * 8 + 511) / 512;    // float4 loads per thread per slab (3)
constexpr int A_IT2 = (HALO * 4 + 511) / 512;   // X2 input: 32-byte items (8 channels: hi, lo) per thread per slab (2)
constexpr int AROW = 160;                       // bytes per halo pixel in LDS (32 bf16 hi | 32 bf16 lo | 32 B pad)
constexpr int A_BYTES = HALO * AROW, B_BYTES = BN * 128, NBUF = 3;
constexpr int CLD = BN + 4;                     // C tile row pitch (floats): rows shift by 16 B over the banks
constexpr int ROWS = TH * TW;                   // 128 pixels
constexpr int MAIN_BYTES = 2 * A_BYTES + NBUF * B_BYTES;
constexpr int EPI_FLOATS = ROWS * CLD + 2 * ROWS + 2 * BN;  // C tile + (mean, rstd) + (ln weight, ln bias)
constexpr int SMEM_FLOATS = MAIN_BYTES / 4 > EPI_FLOATS ? MAIN_BYTES / 4 : EPI_FLOATS;
static_assert(SMEM_FLOATS * 4 <= 160 * 1024, "LDS budget");
constexpr int NA = 4, NJ = 4, ND = 4;           // pixel runs / 16-channel columns / DMA pieces (8 rows x 128 B) per wave
}  // namespace g256

// 1 / (1 + 2^(-v log2 e)) on v_exp_f32 / v_rcp_f32 (1 ulp each): four instructions instead of expf + an IEEE division (~20) in a
// store loop that is VALU-bound (16 rows x 4 channels per thread); within 3e-7 of act_apply(PRV2_ACT_SIGMOID)
__device__ __forceinline__ float sigmoid_fast(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}

struct GateConvParams {
  IgemmParams c;           // the 3x3 conv: x, w, bias, ln_w, ln_b, ln_eps, act; final stage: mul, res, y (+ their strides)
  const void* gate_w;      // fragment-major packed 256 x 256 gate weights, or null: y = act(LN(conv + bias))
  const float* gate_bias;
  int x_x2, mul_x2, y_x2;  // operand formats (X2 = pre-split, see the head of this file); y_x2: the no-gate, no-LayerNorm kernel
  const float* pre;        // pre-LayerNorm addend [n, h, w, ld_pre] (prv2_conv3x3_ln_gate_pre: the conv's coarse half, coarse_taps.hip), or null
  int ld_pre;
  long long* stamps;       // -DPRV2_GATE_STAMPS builds (tools/probes/gate_phase_stamps.sh): 10 s_memtime stamps per wave
};

#ifdef PRV2_GATE_STAMPS
#define PRV2_STAMP(i)                                                                                      \
  do {                                                                                                     \
    if (gp.stamps && (threadIdx.x & 63) == 0) gp.stamps[((long long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 10 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
// in-kernel clock (MI355X_MICROARCH.md 'DVFS give-back' item 6): s_memtime / s_memrealtime at the start and the end of a workgroup,
// into a region of the stamp buffer of their own
#define PRV2_CLK_STAMP(i)                                                                                  \
  do {                                                                                                     \
    if (gp.stamps && threadIdx.x == 0) {                                                                   \
      long long* q_ = gp.stamps + 8000000 + ((long long)blockIdx.x * 2 + (i)) * 2;                         \
      q_[0] = __builtin_amdgcn_s_memtime();                                                                \
      q_[1] = __builtin_amdgcn_s_memrealtime();                                                            \
    }                                                                                                      \
  } while (0)
#else
#define PRV2_STAMP(i)
#define PRV2_CLK_STAMP(i)
#endif

template <int PREC, bool GATE, bool X2IN = false>
__device__ __forceinline__ void c256_body(const GateConvParams& gp, float* smem) {
  using namespace g256;
  constexpr int NIT = X2IN ? A_IT2 : A_IT;  // halo items per thread per slab
  constexpr int LPI = X2IN ? 2 : 1;         // 16-byte loads per item
  const IgemmParams& p = gp.c;
  char* const As_b = reinterpret_cast<char*>(smem);
  char* const Bs_b = As_b + 2 * A_BYTES;
  float* const csm = smem;

  // ---- XCD-aware block -> pixel tile -------------------------------------------------------------------
  const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;  // (ragged right / bottom edges: zero-filled halo, guarded stores)
  const int ntiles = gridDim.x;
  int t = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = t & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
  }
  const int tx = t % tiles_x;
  const int ty = (t / tiles_x) % tiles_y;
  const int n_img = t / (tiles_x * tiles_y);
  const int y0 = ty * TH, x0 = tx * TW;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;  // rows 4wm..4wm+3 of the tile, channels 64wn..64wn+63 (waves w, w + 4 share a SIMD)
  const int m16 = lane & 15, g = lane >> 4;
  const int chunk = tid & 7, prow_lin = tid >> 3;
  const int prow = (prow_lin & ~3) | ((prow_lin & 1) << 1) | ((prow_lin >> 1) & 1);  // (bank spread of the ds_write pairs)

  // ---- halo loader: buffer loads with hardware zero fill (see conv3x3_m16.hip) -----------------------------------
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  constexpr unsigned OOB = 0x80000000u;
  i32x4 rsrc;
  unsigned hoff[NIT];
  {
    const unsigned long long img_base = (unsigned long long)(size_t)(p.x + (long long)n_img * p.x_bstride);
    rsrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)img_base);
    rsrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((img_base >> 32) & 0xffffu));
    rsrc.z = __builtin_amdgcn_readfirstlane((int)(unsigned)((((long long)p.H * p.W - 1) * p.ldx + p.Cin) * 4));
    rsrc.w = 0x00020000;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      // fp32 input: item = (halo pixel prow + 64 it, 4 channels `chunk`); X2: item tid + 512 it = (halo pixel, 8-channel group)
      const int hp = X2IN ? (tid + 512 * it) >> 2 : prow + 64 * it;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = hp < HALO && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      hoff[it] = ok ? (unsigned)(((iy * p.W + ix) * p.ldx + (X2IN ? (tid & 3) * 8 : chunk * 4)) * 4) : OOB;
    }
  }
  const long long w_row_stride = 9LL * p.Cin_pad;
  const int cslabs = p.Cin_pad / BK;  // (Cin % 32 == 0: no partial slab, no tail tile)
  const int nsteps = 9 * cslabs;
  const int relu_floor = p.relu_in ? 0 : (int)0x80000000;

  f32x4 ra[NIT][LPI];
  auto load_a_async = [&](int cc, int it) {
    const unsigned voff = hoff[it] + (unsigned)(cc * BK * 4);  // (2^31 + cc*128 stays out of range: no wrap)
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ra[it][0]) : "v"(voff), "s"(rsrc) : "memory");
    if constexpr (X2IN) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(ra[it][LPI - 1]) : "v"(voff), "s"(rsrc) : "memory");
  };
#define PRV2_WAIT_A(newer, it)                                                                                              \
  do {                                                                                                                    \
    if constexpr (!X2IN) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ra[it][0]) : "n"(newer) : "memory");                    \
    else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(ra[it][0]), "+v"(ra[it][LPI - 1]) : "n"(newer) : "memory");            \
  } while (0)
  auto store_a = [&](int abuf, int it) {
    if constexpr (X2IN) {  // pre-split input: the 8 channels' hi / lo halves go to the two planes of the LDS row as they are
      const int item = tid + 512 * it, hp = item >> 2;
      if (hp >= HALO) return;
      const unsigned addr = (unsigned)(size_t)(As_b + abuf * A_BYTES + hp * AROW) + (item & 3) * 16;
      asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:64" ::"v"(addr), "v"(ra[it][0]), "v"(ra[it][LPI - 1]) : "memory");
      return;
    }
    const int hp = prow + 64 * it;
    if (hp >= HALO) return;
    typedef int i32x4v __attribute__((ext_vector_type(4)));
    i32x4v vi = __builtin_bit_cast(i32x4v, ra[it][0]);
    vi.x = max(vi.x, relu_floor);
    vi.y = max(vi.y, relu_floor);
    vi.z = max(vi.z, relu_floor);
    vi.w = max(vi.w, relu_floor);
    const f32x4 v = __builtin_bit_cast(f32x4, vi);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const bf16x4 hi = __builtin_convertvector(v, bf16x4);
    const u32x2 hw = __builtin_bit_cast(u32x2, hi);
    f32x4 hf;
    hf.x = __builtin_bit_cast(float, hw.x << 16);
    hf.y = __builtin_bit_cast(float, hw.x & 0xffff0000u);
    hf.z = __builtin_bit_cast(float, hw.y << 16);
    hf.w = __builtin_bit_cast(float, hw.y & 0xffff0000u);
    const bf16x4 lo = __builtin_convertvector(v - hf, bf16x4);
    const unsigned addr = (unsigned)(size_t)(As_b + abuf * A_BYTES + hp * AROW) + chunk * 8;
    const unsigned long long h = __builtin_bit_cast(unsigned long long, hi), l = __builtin_bit_cast(unsigned long long, lo);
    if constexpr (PREC == PRV2_PREC_BF16X3) asm volatile("ds_write2_b64 %0, %1, %2 offset1:8" ::"v"(addr), "v"(h), "v"(l) : "memory");
    else asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(h) : "memory");
  };
  // weight tile of step s: 32 pieces of 8 rows x 128 B, wave w moves pieces 4w..4w+3 (linear in LDS, pre-swizzled in HBM)
  const int dma_row = lane >> 3, dma_slot = lane & 7;
  const float* wdma = reinterpret_cast<const float*>(p.w) + (long long)dma_row * w_row_stride + dma_slot * 4;
  auto dma_src = [&](int s, int i) {
    const int cc = s / 9, tap = s - cc * 9;
    return wdma + (long long)tap * p.Cin_pad + cc * BK + (long long)((wave * ND + i) * 8) * w_row_stride;
  };
  auto dma_dst = [&](int bbuf, int i) { return Bs_b + bbuf * B_BYTES + (wave * ND + i) * 1024; };
  auto dma_b = [&](int s, int bbuf, int i) {  // prologue: compiler-visible
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)dma_src(s, i),
                                     (__attribute__((address_space(3))) void*)dma_dst(bbuf, i), 16, 0, 0);
  };
  auto dma_b_async = [&](int s, int bbuf, int i) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dma_dst(bbuf, i));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(dma_src(s, i)) : "memory");
  };

  f32x4 acc[NA][NJ];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addressing: run a = tile row 4wm + a, lane m = pixel m of it, lane group g = channels 8g..8g+7 --------
  const char* const a_lane = As_b + ((4 * wm) * HW_ + m16) * AROW + g * 16;
  auto run_off = [](int a, int ky, int kx) { return (a + ky) * HW_ + kx; };
  const int b_key = (m16 >> 1) & 7;
  const char* const b_lane_hi = Bs_b + (wn * 64 + m16) * 128 + ((g ^ b_key) << 4);
  const char* const b_lane_lo = Bs_b + (wn * 64 + m16) * 128 + (((4 + g) ^ b_key) << 4);
  bf16x8 ah[NA], al[NA], bh[2], bl[2];
  auto read_a = [&](int a, int abuf, int tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const char* q = a_lane + abuf * A_BYTES + run_off(a, ky, kx) * AROW;
    ah[a] = *reinterpret_cast<const bf16x8*>(q);
    if constexpr (PREC == PRV2_PREC_BF16X3) al[a] = *reinterpret_cast<const bf16x8*>(q + 64);
  };
  auto read_b = [&](int slot, int bbuf, int j) {
    bh[slot] = *reinterpret_cast<const bf16x8*>(b_lane_hi + bbuf * B_BYTES + j * 16 * 128);
    if constexpr (PREC == PRV2_PREC_BF16X3) bl[slot] = *reinterpret_cast<const bf16x8*>(b_lane_lo + bbuf * B_BYTES + j * 16 * 128);
  };
  constexpr int NP = PREC == PRV2_PREC_BF16X3 ? 3 : 1;
  auto mma = [&](f32x4& c, const bf16x8& xh, const bf16x8& xl, const bf16x8& wh, const bf16x8& wl, int pr) {
    if constexpr (PREC == PRV2_PREC_BF16X3) {
      if (pr == 0) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, wh, c, 0, 0, 0);
      if (pr == 1) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wl, c, 0, 0, 0);
      if (pr == 2) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wh, c, 0, 0, 0);
    } else {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wh, c, 0, 0, 0);
    }
  };

  // ---- prologue ---------------------------------------------------------------------------------------------
  PRV2_STAMP(0);
  PRV2_CLK_STAMP(0);
#pragma unroll
  for (int it = 0; it < NIT; ++it) load_a_async(0, it);
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    dma_b(0, 0, i);
    dma_b(1, 1, i);
    dma_b(2, 2, i);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    PRV2_WAIT_A(0, it);
    store_a(0, it);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int a = 0; a < NA; ++a) read_a(a, 0, 0);
  read_b(0, 0, 0);
  PRV2_STAMP(1);

  for (int cc = 0; cc < cslabs; ++cc) {
    const int ccn = cc + 1 < cslabs ? cc + 1 : cc;  // behind the last slab: clamped data nobody reads
    const int ab = cc & 1;
    auto step = [&](auto tap_c) {
      constexpr int tap = decltype(tap_c)::value;
      constexpr int L0 = tap < NIT ? LPI : 0, Lm1 = (tap >= 1 && tap - 1 < NIT) ? LPI : 0;  // halo loads issued at this / the previous tap
      const int s = cc * 9 + tap;
      const int s3 = s + 3 < nsteps ? s + 3 : nsteps - 1;
      constexpr int bb = tap % 3;
#pragma unroll
      for (int j = 0; j < NJ - 1; ++j) {
        read_b((j + 1) & 1, bb, j + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < NA; a += 2) {
#pragma unroll
          for (int pr = 0; pr < NP; ++pr) {
            mma(acc[a][j], ah[a], al[a], bh[j & 1], bl[j & 1], pr);
            mma(acc[a + 1][j], ah[a + 1], al[a + 1], bh[j & 1], bl[j & 1], pr);
          }
          __builtin_amdgcn_sched_barrier(0);
#ifndef PRV2_ABL_NOA  // (timing ablations of tools/probes/gate_phase_stamps.sh: results are wrong with any of them)
          if (j == 0 && a == 0) {
            if constexpr (tap < NIT) load_a_async(ccn, tap);
          }
#endif
          if (j == 1 && a == NA - 2) {
#ifndef PRV2_ABL_NOA
            if constexpr (tap >= 2 && tap - 2 < NIT) {
              // VMEM instructions issued since load_a_async(tap - 2): DMAs of steps tap-2 and tap-1, loads tap-1, tap
              constexpr int newer = 2 * ND + Lm1 + L0;
              PRV2_WAIT_A(newer, tap - 2);
              store_a(ab ^ 1, tap - 2);
            }
#endif
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#ifdef PRV2_ABL_NOBAR
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(ND + Lm1 + L0) : "memory");
#else
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(ND + Lm1 + L0) : "memory");
#endif
      read_b(NJ & 1, (tap + 1) % 3, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < NA; a += 2) {
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
          mma(acc[a][NJ - 1], ah[a], al[a], bh[(NJ - 1) & 1], bl[(NJ - 1) & 1], pr);
          mma(acc[a + 1][NJ - 1], ah[a + 1], al[a + 1], bh[(NJ - 1) & 1], bl[(NJ - 1) & 1], pr);
        }
        __builtin_amdgcn_sched_barrier(0);
        read_a(a, tap == 8 ? ab ^ 1 : ab, (tap + 1) % 9);
        read_a(a + 1, tap == 8 ? ab ^ 1 : ab, (tap + 1) % 9);
#ifndef PRV2_ABL_NOB
        dma_b_async(s3, tap % 3, a);      // this step's tile buffer is free since the barrier
        dma_b_async(s3, tap % 3, a + 1);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{});
    step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{});
    step(std::integral_constant<int, 7>{});
    step(std::integral_constant<int, 8>{});
  }
  // the clamped DMAs of the last steps are inline asm: drain them by hand before the C tile overwrites the buffers
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  PRV2_STAMP(2);

  // ---- pre-LayerNorm addend (the conv's coarse half, gathered per tile by coarse_taps.hip): 64 threads x float4 = one 1 KB pixel
  // row, 16 rows per thread, all requested here -- they land while the C tile is written -- and added to the C tile in LDS
  constexpr int PC4 = BN / 4, PRPP = 512 / PC4, PNR = ROWS / PRPP;
  const bool has_pre = gp.pre != nullptr;  // block-uniform
  f32x4 pv[PNR];
  if (has_pre) {
    const __amdgpu_buffer_rsrc_t pre_rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(gp.pre + (long long)n_img * p.H * p.W * gp.ld_pre), 0, (int)(((unsigned)(p.H * p.W - 1) * gp.ld_pre + BN) * 4u), 0x00020000);
#pragma unroll
    for (int i = 0; i < PNR; ++i) {
      const int rr = tid / PC4 + i * PRPP;
      const int pix = min(y0 + rr / TW, p.H - 1) * p.W + min(x0 + (rr & (TW - 1)), p.W - 1);
      pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pre_rs, (pix * gp.ld_pre + (tid % PC4) * 4) * 4, 0, 0));
    }
  }
  // ---- C tile (conv + bias) -> LDS; LayerNorm parameters beside it ----------------------------------------------
  float* const ln_stats = csm + ROWS * CLD;   // [mean | rstd]
  float* const ln_par = ln_stats + 2 * ROWS;  // [weight | bias]
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = wn * 64 + j * 16 + m16;
    const float b = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int e = 0; e < 4; ++e) csm[((4 * wm + a) * TW + 4 * g + e) * CLD + col] = acc[a][j][e] + b;
  }
  const bool has_ln = GATE || p.ln_w != nullptr;  // block-uniform
  if (has_ln && tid < BN) {
    ln_par[tid] = p.ln_w[tid];
    ln_par[BN + tid] = p.ln_b[tid];
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (bare barriers from here on: __syncthreads() also drains vmcnt)
  if (has_pre) {
#pragma unroll
    for (int i = 0; i < PNR; ++i) {
      f32x4* q = reinterpret_cast<f32x4*>(&csm[(tid / PC4 + i * PRPP) * CLD + (tid % PC4) * 4]);
      *q = *q + pv[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  PRV2_STAMP(3);
  // Gate weights: wave w multiplies ALL 128 pixels with gate columns 32w .. 32w + 31, so that every weight fragment is fetched
  // by exactly one wave (256 KB per tile from L2; 64 x 64 wave tiles fetched 512 KB and the GEMM ran at half the MFMA rate).
  // All 32 fragments of the wave (128 registers, free between the two GEMMs) are requested HERE and land during the row
  // statistics and the normalisation pass: a weight fetch issued inside the GEMM costs the wave an L2 round trip per slab --
  // the two waves of a SIMD then ran one after the other, 21 k cycles for 12.3 k of MFMAs (per-wave stamps,
  // tools/probes/gate_phase_stamps.sh).
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr int NA2 = 8, NJ2 = 2, KS2 = BN / 32;
  u32x4 wf[GATE ? KS2 : 1][NJ2][2];  // [slab][column][hi / lo]
  if constexpr (GATE) {
    const u32x4* const gw = reinterpret_cast<const u32x4*>(gp.gate_w) + lane;
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
      for (int j = 0; j < NJ2; ++j) {
        wf[ks][j][0] = gw[gate_frag_index(BN, wave * NJ2 + j, ks, 0)];
        if constexpr (PREC == PRV2_PREC_BF16X3) wf[ks][j][1] = gw[gate_frag_index(BN, wave * NJ2 + j, ks, 1)];
      }
  }
  // row statistics, two passes like convs.py:25-27; threads 4r..4r+3 share pixel r: thread `part` takes the 16-channel
  // groups 64k + 16 part (k = 0..3) -- the 16 lanes (4 rows x 4 parts) of a ds_read_b128 group then hit 16 distinct bank slots
  if (has_ln) {
    const int r = tid >> 2, part = tid & 3;
    const float* q = csm + r * CLD + part * 16;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 64; c += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(q + (c >> 4) * 64 + (c & 15));
      s += (v.x + v.y) + (v.z + v.w);
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    const float mean = s / (float)BN;
    float d2 = 0.f;
#pragma unroll
    for (int c = 0; c < 64; c += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(q + (c >> 4) * 64 + (c & 15));
      const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
      d2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    d2 += __shfl_xor(d2, 1);
    d2 += __shfl_xor(d2, 2);
    if (part == 0) {
      ln_stats[r] = mean;
      ln_stats[ROWS + r] = 1.0f / sqrtf(d2 / (float)BN + p.ln_eps);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  PRV2_STAMP(4);

  if constexpr (GATE) {
    // ---- gate GEMM: wave w = all 128 pixels (8 runs) x all 256 normalised channels x gate columns 32w .. 32w + 31 ---------
    f32x4 acc2[NA2][NJ2];
#pragma unroll
    for (int a = 0; a < NA2; ++a)
#pragma unroll
      for (int j = 0; j < NJ2; ++j) acc2[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // normalise + activate + split the C tile IN PLACE, once: the 32 bytes of 8 fp32 channels become [8 bf16 hi | 8 bf16 lo] --
    // exactly one A fragment of the gate GEMM.  Thread = pixel (tid & 127) x every 4th 8-channel chunk: the 16 lanes of a
    // ds_read_b128 group are 16 consecutive rows (row pitch 65 x 16 B: 16 distinct bank slots), for this pass and for the
    // fragment reads below alike.
    {
      const int r = tid & (ROWS - 1);
      const float mean = ln_stats[r], rstd = ln_stats[ROWS + r];
      const float act_floor = p.act == PRV2_ACT_RELU ? 0.f : -__builtin_inff();  // (host: ReLU or none in front of the gate)
#pragma unroll
      for (int i = 0; i < BN / 32; ++i) {
        const int c8 = (tid >> 7) + 4 * i;
        float* q = csm + r * CLD + c8 * 8;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(q), v1 = *reinterpret_cast<const f32x4*>(q + 4);
        const f32x4 lw0 = *reinterpret_cast<const f32x4*>(ln_par + c8 * 8), lw1 = *reinterpret_cast<const f32x4*>(ln_par + c8 * 8 + 4);
        const f32x4 lb0 = *reinterpret_cast<const f32x4*>(ln_par + BN + c8 * 8), lb1 = *reinterpret_cast<const f32x4*>(ln_par + BN + c8 * 8 + 4);
        v0 = (v0 - mean) * rstd * lw0 + lb0;  // (vector form: v_pk_add / v_pk_mul_f32, two channels per instruction)
        v1 = (v1 - mean) * rstd * lw1 + lb1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v0[e] = fmaxf(v0[e], act_floor);
          v1[e] = fmaxf(v1[e], act_floor);
        }
        bf16x4 h0, l0, h1, l1;
        split_bf16(v0, h0, l0);
        split_bf16(v1, h1, l1);
        *reinterpret_cast<bf16x8*>(q) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        *reinterpret_cast<bf16x8*>(q + 4) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (bare: the weight fragments stay in flight)
    PRV2_STAMP(5);
    // ---- operands of the final stage y = mul * sigmoid(gate + bias) + res, ALL requested long before their use: with one workgroup
    // per CU nothing else hides this traffic, and a CU only reaches its memory rate with a whole operand tile (128 KB) in flight
    // (4 rows in flight: 19 k cycles for the store loop; phase stamps of tools/probes/gate_phase_stamps.sh).  Thread = NC channels of
    // a pixel: fp32 mul: 4 channels (64 threads x float4 = one 1 KB pixel row, 16 rows per thread); pre-split (X2) mul: 8 channels =
    // one [8 hi | 8 lo] group (32 threads per row, 8 rows per thread) -- 16-byte accesses either way.
    // buffer loads (per-image base, 32-bit offsets; an absent operand is a resource of zero records: the hardware returns zeros,
    // no branches around the loads)
    const long long img_m = (long long)n_img * p.H * p.W;
    const unsigned img_px = (unsigned)(p.H * p.W - 1);
    const __amdgpu_buffer_rsrc_t mul_rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.mul ? p.mul + img_m * p.ld_mul : p.x), 0, p.mul ? (int)((img_px * p.ld_mul + BN) * 4u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.res ? p.res + img_m * p.ld_res : p.x), 0, p.res ? (int)((img_px * p.ld_res + BN) * 4u) : 0, 0x00020000);
    const bool has_mul = p.mul != nullptr;  // block-uniform
    // Row of the 16-pixel run that MFMA row m16 stands for in the gate GEMM: lanes 4..11 take the EVEN rows, lanes 0..3 / 12..15 the odd ones.
    // ds_read_b128 serves the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): with row = m16 and
    // the 65-slot row pitch the k-slice g = 1 lanes of rows 10 / 11 met rows 12 / 13 of slice 0 on the same banks -- every fragment read
    // of the gate GEMM was 2-way (SQ_LDS_BANK_CONFLICT 22-33M per launch, the 3x3 main loop alone: 0.7M).  Any row permutation is legal:
    // the accumulator rows come back through the same map (grow below).
    const int m16r = (m16 >= 4 && m16 < 12) ? 2 * (m16 - 4) : 2 * (m16 & 3) + 1 + (m16 >= 12 ? 8 : 0);
    const int grow = g == 0 ? 1 : (g == 3 ? 9 : 8 * (g - 1));  // accumulator element e of lane group g = row grow + 2 e
    auto final_stage = [&](auto nc_c) {
      constexpr int NC = decltype(nc_c)::value, NV = NC / 4;  // channels per thread, float4 per thread and row
      constexpr bool MX2 = NC == 8;                            // mul arrives pre-split
      constexpr int CG = BN / NC, RPP = 512 / CG, NR = ROWS / RPP;
      const int colg = tid % CG;
      auto pix_of = [&](int i) {  // (rows below the image: clamped address, never stored)
        const int rr = tid / CG + i * RPP;
        return min(y0 + rr / TW, p.H - 1) * p.W + min(x0 + (rr & (TW - 1)), p.W - 1);
      };
      f32x4 mv[NR][NV], rv[NR][NV];
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) {
        if (ks == KS2 - 3) {  // five slabs of weight registers are free again: the mul rows fly during the rest of the GEMM
#pragma unroll
          for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int v = 0; v < NV; ++v)  // (X2: v = 0 the group's hi half, v = 1 its lo half -- adjacent 16-byte pieces, like two float4)
              mv[i][v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mul_rs, (pix_of(i) * p.ld_mul + colg * NC + 4 * v) * 4, 0, 0));
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // four pixel runs at a time (their fragments: 32 registers)
          bf16x8 xh[4], xl[4];
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const float* q = csm + ((4 * h + a) * TW + m16r) * CLD + ks * 32 + 8 * g;
            xh[a] = *reinterpret_cast<const bf16x8*>(q);
            xl[a] = *reinterpret_cast<const bf16x8*>(q + 4);
          }
#pragma unroll
          for (int j = 0; j < NJ2; ++j) {
            const bf16x8 wh = __builtin_bit_cast(bf16x8, wf[ks][j][0]);
            const bf16x8 wl = __builtin_bit_cast(bf16x8, wf[ks][j][1]);
#pragma unroll
            for (int pr = 0; pr < NP; ++pr)
#pragma unroll
              for (int a = 0; a < 4; ++a) mma(acc2[4 * h + a][j], xh[a], xl[a], wh, wl, pr);
          }
        }
      }
      PRV2_STAMP(6);
#pragma unroll
      for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int v = 0; v < NV; ++v)
          rv[i][v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, (pix_of(i) * p.ld_res + colg * NC + 4 * v) * 4, 0, 0));
      // (bare barriers: __syncthreads() would first drain vmcnt, i.e. wait for the rows just requested)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave has read its rows of the C tile
#pragma unroll
      for (int j = 0; j < NJ2; ++j) {
        const int col = wave * 32 + j * 16 + m16;
#pragma unroll
        for (int a = 0; a < NA2; ++a)
#pragma unroll
          for (int e = 0; e < 4; ++e) csm[(a * TW + grow + 2 * e) * CLD + col] = acc2[a][j][e];
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      PRV2_STAMP(7);
      f32x4 gb[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) gb[v] = gp.gate_bias ? *reinterpret_cast<const f32x4*>(gp.gate_bias + colg * NC + 4 * v) : f32x4{0.f, 0.f, 0.f, 0.f};
      float* const ybase = p.y + (long long)n_img * p.y_bstride + colg * NC;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int rr = tid / CG + i * RPP;
        // ``mul`` counts as hi + lo of its bf16 split in EITHER format (this unit's conv input is the same tensor and sees exactly
        // these 16 bits): the result does not depend on the format the producer chose
        f32x4 mf[NV];
        if constexpr (MX2) {
          typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
          const u32x4v hh = __builtin_bit_cast(u32x4v, mv[i][0]), ll = __builtin_bit_cast(u32x4v, mv[i][NV - 1]);
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            mf[v][0] = __builtin_bit_cast(float, hh[2 * v] << 16) + __builtin_bit_cast(float, ll[2 * v] << 16);
            mf[v][1] = __builtin_bit_cast(float, hh[2 * v] & 0xffff0000u) + __builtin_bit_cast(float, ll[2 * v] & 0xffff0000u);
            mf[v][2] = __builtin_bit_cast(float, hh[2 * v + 1] << 16) + __builtin_bit_cast(float, ll[2 * v + 1] << 16);
            mf[v][3] = __builtin_bit_cast(float, hh[2 * v + 1] & 0xffff0000u) + __builtin_bit_cast(float, ll[2 * v + 1] & 0xffff0000u);
          }
        } else {
          bf16x4 mh, ml;
          split_bf16(mv[i][0], mh, ml);
          mf[0] = __builtin_convertvector(mh, f32x4) + __builtin_convertvector(ml, f32x4);
        }
        const bool inside = y0 + rr / TW < p.H && x0 + (rr & (TW - 1)) < p.W;
        float* dst = ybase + (long long)pix_of(i) * p.ldy;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const f32x4 cv = *reinterpret_cast<const f32x4*>(&csm[rr * CLD + colg * NC + 4 * v]);
          f32x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = (has_mul ? mf[v][e] : 1.0f) * sigmoid_fast(cv[e] + gb[v][e]) + rv[i][v][e];
          if (inside) {
            float* d2 = dst + 4 * v;
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(d2), "v"(ov) : "memory");
          }
        }
      }
    };
    final_stage(std::integral_constant<int, X2IN ? 8 : 4>{});  // (host: mul comes in the format of x -- it IS the first half of x)
    PRV2_STAMP(8);
#ifdef PRV2_GATE_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PRV2_STAMP(9);
    PRV2_CLK_STAMP(1);
#endif
    return;
  }

  if (gp.y_x2) {  // block-uniform (host: no LayerNorm)
    // ---- no gate, pre-split output: y = X2(act(conv + bias) + res) -- GatedConvUnit.conv writing ``out`` into the unit's concat
    // buffer in the operand format of its only consumer (the gate kernel).  32 threads x 8 channels = one 1 KB pixel row
    // ([8 hi | 8 lo] per thread: two adjacent 16-byte stores), 8 rows per thread, residual rows requested up front
    constexpr int C8 = BN / 8, RPP8 = 512 / C8, NR8 = ROWS / RPP8;
    const int col8 = tid % C8;
    auto pix8 = [&](int i) {
      const int rr = tid / C8 + i * RPP8;
      return min(y0 + rr / TW, p.H - 1) * p.W + min(x0 + (rr & (TW - 1)), p.W - 1);
    };
    const __amdgpu_buffer_rsrc_t res_rs8 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.res ? p.res + (long long)n_img * p.H * p.W * p.ld_res : p.x), 0, p.res ? (int)(((unsigned)(p.H * p.W - 1) * p.ld_res + BN) * 4u) : 0,
        0x00020000);
    f32x4 rv8[NR8][2];
#pragma unroll
    for (int i = 0; i < NR8; ++i) {
      rv8[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rs8, (pix8(i) * p.ld_res + col8 * 8) * 4, 0, 0));
      rv8[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rs8, (pix8(i) * p.ld_res + col8 * 8 + 4) * 4, 0, 0));
    }
    float* const ybase8 = p.y + (long long)n_img * p.y_bstride + col8 * 8;
    dispatch_act(p.act, [&](auto act_c) {
#pragma unroll
      for (int i = 0; i < NR8; ++i) {
        const int rr = tid / C8 + i * RPP8;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(&csm[rr * CLD + col8 * 8]), v1 = *reinterpret_cast<const f32x4*>(&csm[rr * CLD + col8 * 8 + 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v0[e] = act_apply_bf(v0[e], decltype(act_c)::value) + rv8[i][0][e];
          v1[e] = act_apply_bf(v1[e], decltype(act_c)::value) + rv8[i][1][e];
        }
        bf16x4 h0, l0, h1, l1;
        split_bf16(v0, h0, l0);
        split_bf16(v1, h1, l1);
        const bf16x8 hv = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 lv = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        if (y0 + rr / TW < p.H && x0 + (rr & (TW - 1)) < p.W) {
          float* dst = ybase8 + (long long)pix8(i) * p.ldy;
          asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16\n\ts_nop 1" ::"v"(dst), "v"(hv), "v"(lv) : "memory");
        }
      }
    });
    return;
  }
  // ---- no gate: y = act([LN](conv + bias)) (+ res); 64 threads x float4 = one 1 KB pixel row, 16 rows per thread; the residual
  // rows are all requested up front (see the gate stage)
  constexpr int C4 = BN / 4, RPP = 512 / C4, NR = ROWS / RPP;
  const int col4 = tid % C4;
  auto pix_of = [&](int i) {
    const int rr = tid / C4 + i * RPP;
    return min(y0 + rr / TW, p.H - 1) * p.W + min(x0 + (rr & (TW - 1)), p.W - 1);
  };
  const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.res ? p.res + (long long)n_img * p.H * p.W * p.ld_res : p.x), 0, p.res ? (int)(((unsigned)(p.H * p.W - 1) * p.ld_res + BN) * 4u) : 0,
      0x00020000);
  f32x4 rv[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) rv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, (pix_of(i) * p.ld_res + col4 * 4) * 4, 0, 0));
  float* const ybase = p.y + (long long)n_img * p.y_bstride + col4 * 4;
  auto store_rows = [&](auto act_c, auto ln_c) {
    constexpr bool LN = decltype(ln_c)::value;
    f32x4 lw = {1.f, 1.f, 1.f, 1.f}, lb = {0.f, 0.f, 0.f, 0.f};
    if constexpr (LN) {
      lw = *reinterpret_cast<const f32x4*>(ln_par + col4 * 4);
      lb = *reinterpret_cast<const f32x4*>(ln_par + BN + col4 * 4);
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int rr = tid / C4 + i * RPP;
      const f32x4 cv = *reinterpret_cast<const f32x4*>(&csm[rr * CLD + col4 * 4]);
      f32x4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = cv[e];
        if constexpr (LN) t = (t - ln_stats[rr]) * ln_stats[ROWS + rr] * lw[e] + lb[e];
        ov[e] = act_apply_bf(t, decltype(act_c)::value) + rv[i][e];
      }
      if (y0 + rr / TW < p.H && x0 + (rr & (TW - 1)) < p.W) {
        float* dst = ybase + (long long)pix_of(i) * p.ldy;
        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(ov) : "memory");
      }
    }
  };
  dispatch_act(p.act, [&](auto act_c) {
    if (has_ln) store_rows(act_c, std::true_type{});
    else store_rows(act_c, std::false_type{});
  });
}

// (named like the other matrix kernels -- name<columns, precision> -- so that prv2_last_kernel(), rocprofv3 and the PMC tables agree)
template <int BN_, int PREC>
__global__ void __launch_bounds__(512, 2) conv3x3_c256_kernel(const GateConvParams gp) {
  static_assert(BN_ == g256::BN, "one tile width");
  __shared__ __attribute__((aligned(16))) float smem[g256::SMEM_FLOATS];
  c256_body<PREC, false>(gp, smem);
}
template <int BN_, int PREC>
__global__ void __launch_bounds__(512, 2) conv3x3_c256_gate_kernel(const GateConvParams gp) {
  static_assert(BN_ == g256::BN, "one tile width");
  __shared__ __attribute__((aligned(16))) float smem[g256::SMEM_FLOATS];
  c256_body<PREC, true>(gp, smem);
}
// the gate kernel on a pre-split (X2) input: the halo loader only copies
template <int BN_, int PREC>
__global__ void __launch_bounds__(512, 2) conv3x3_c256_gate_x2_kernel(const GateConvParams gp) {
  static_assert(BN_ == g256::BN, "one tile width");
  __shared__ __attribute__((aligned(16))) float smem[g256::SMEM_FLOATS];
  c256_body<PREC, true, true>(gp, smem);
}

}  // namespace prv2
#ifdef PRV2_EXPERIMENTS  // (make EXPERIMENTS=1: the round-3 power-wall experiment; not part of the default build)
#include "conv3x3_w4.h"
#endif
namespace prv2 {

// C x C gate weights (PyTorch [cout][cin][1][1]; C = 32, 128, 256) -> the fragment-major image of igemm.h::gate_frag_index
__global__ void __launch_bounds__(256) pack_gate_weight_kernel(const float* __restrict__ w, unsigned* __restrict__ dst, int c) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one (column block, slab, lane) = 8 weights
  const int ks_n = c / 32;
  if (idx >= (c / 16) * ks_n * 64) return;
  const int lane = idx & 63, ks = (idx >> 6) % ks_n, cb = (idx >> 6) / ks_n;
  const int m = lane & 15, g = lane >> 4;
  const float* src = w + (long long)(16 * cb + m) * c + 32 * ks + 8 * g;
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
  bf16x4 h0, l0, h1, l1;
  split_bf16(v0, h0, l0);
  split_bf16(v1, h1, l1);
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 a = __builtin_bit_cast(u32x2, h0), b = __builtin_bit_cast(u32x2, h1), cc = __builtin_bit_cast(u32x2, l0), d = __builtin_bit_cast(u32x2, l1);
  unsigned* o = dst + (gate_frag_index(c, cb, ks, 0) + lane) * 4;
  o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
  o = dst + (gate_frag_index(c, cb, ks, 1) + lane) * 4;
  o[0] = cc.x; o[1] = cc.y; o[2] = d.x; o[3] = d.y;
}

static inline bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

static bool gate_conv_shape_ok(const prv2_conv_desc* d) {
  return d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->convt_k == 0 && !d->same_pad && d->cout == g256::BN && d->cin >= 32 &&
         d->cin % 32 == 0 && d->w >= g256::TW && d->prec != PRV2_PREC_F32 && (long long)d->h * d->w * d->ldx < (1LL << 29);
}

// prv2_conv2d's dispatch: 3x3 convs with 256 output channels whose epilogue is bias, [LayerNorm,] activation, [+ res]
bool conv3x3_c256_eligible(const prv2_conv_desc* d, const float* x, const float* res, const float* y) {
  static const bool off = getenv("PRV2_NO_C256") != nullptr;  // A/B switch
  const long long px = (long long)d->h * d->w;
  return !off && gate_conv_shape_ok(d) && d->ldx % 4 == 0 && aligned16(x) && d->x_bstride % 4 == 0 && d->ldy % 4 == 0 && aligned16(y) &&
         d->y_bstride % 4 == 0 && (!res || (d->ld_res % 4 == 0 && aligned16(res) && px * d->ld_res < (1LL << 29)));
}

}  // namespace prv2

using namespace prv2;

static bool gate_channels_ok(int c) { return c == 32 || c == 128 || c == 256; }

// 32 / 128 channels: the layer runs on the 8 x 32-pixel kernels of conv3x3_m16.hip (their shape contract), gate stage in the epilogue
static bool gate_narrow_shape_ok(const prv2_conv_desc* d) {
  return d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->convt_k == 0 && !d->same_pad && (d->cout == 32 || d->cout == 128) &&
         d->cin >= 32 && d->cin % 32 == 0 && d->w >= 24 && d->h >= 4 && d->prec != PRV2_PREC_F32 && (long long)d->h * d->w * d->ldx < (1LL << 29) &&
         (long long)d->h * d->w * d->ldy < (1LL << 29);
}

extern "C" int prv2_conv3x3_ln_gate_supported(const prv2_conv_desc* d) { return d && (gate_conv_shape_ok(d) || gate_narrow_shape_ok(d)) ? 1 : 0; }

extern "C" int64_t prv2_gate_weight_bytes(int32_t channels) { return gate_channels_ok(channels) ? (int64_t)channels * channels * 4 : 0; }

extern "C" int prv2_pack_gate_weight(const float* w_src, void* w_packed, int32_t cout, int32_t cin, void* stream) {
  PRV2_REQUIRE(w_src && w_packed && aligned16(w_src) && aligned16(w_packed), "pack_gate_weight: null / unaligned pointer");
  PRV2_REQUIRE(cout == cin && gate_channels_ok(cout), "pack_gate_weight: the fused gate is a C -> C 1x1 conv, C = 32, 128 or 256 (got %d -> %d)", cin, cout);
  const int items = (cout / 16) * (cout / 32) * 64;
  hipLaunchKernelGGL(pack_gate_weight_kernel, dim3((unsigned)cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, w_src,
                     reinterpret_cast<unsigned*>(w_packed), (int)cout);
  PRV2_LAUNCH_CHECK("pack_gate_weight");
  return 0;
}

extern "C" int prv2_conv3x3_ln_gate(const prv2_conv_desc* d, const float* x, const void* w_packed, const float* bias, const float* ln_weight,
                                    const float* ln_bias, const void* gate_w_packed, const float* gate_bias, const float* mul, const float* res,
                                    float* y, void* stream) {
  return prv2_conv3x3_ln_gate_pre(d, x, w_packed, bias, nullptr, 0, ln_weight, ln_bias, gate_w_packed, gate_bias, mul, res, y, stream);
}

extern "C" int prv2_conv3x3_ln_gate_pre(const prv2_conv_desc* d, const float* x, const void* w_packed, const float* bias, const float* pre,
                                        int32_t ld_pre, const float* ln_weight, const float* ln_bias, const void* gate_w_packed,
                                        const float* gate_bias, const float* mul, const float* res, float* y, void* stream) {
  PRV2_REQUIRE(d && x && w_packed && y && (ln_weight != nullptr) == (ln_bias != nullptr), "conv3x3_ln_gate: null pointer");
  PRV2_REQUIRE(!pre || (ln_weight && ld_pre >= d->cout && ld_pre % 4 == 0 && aligned16(pre) && (long long)d->h * d->w * ld_pre < (1LL << 29)),
               "conv3x3_ln_gate_pre: the addend goes in front of a LayerNorm; [n, h, w, ld_pre >= cout], ld_pre %% 4 == 0, 16-byte aligned");
  if (gate_w_packed && d->cout != g256::BN) {  // 32 / 128 channels: conv3x3_m16.hip's kernels with the gate stage in their epilogue
    PRV2_REQUIRE(d->fmt == 0, "conv3x3_ln_gate: pre-split (X2) operands are taken at 256 channels only");
    PRV2_REQUIRE(gate_narrow_shape_ok(d) && ln_weight, "conv3x3_ln_gate: 3x3 s1 p1, cout 32 / 128 / 256, cin %% 32 == 0, bf16 modes (got %dx%d %d->%d k%d s%d prec %d)",
                 d->h, d->w, d->cin, d->cout, d->kh, d->stride, d->prec);
    PRV2_REQUIRE(d->act == PRV2_ACT_RELU || d->act == PRV2_ACT_NONE, "conv3x3_ln_gate: ReLU or no activation in front of the gate (act %d)", d->act);
    return conv2d_impl(d, x, w_packed, bias, ln_weight, ln_bias, nullptr, mul, res, nullptr, y, stream, gate_w_packed, gate_bias, nullptr, nullptr, nullptr, 0, 0,
                       pre, ld_pre);
  }
  PRV2_REQUIRE(ln_weight || !gate_w_packed, "conv3x3_ln_gate: the gate stage sits behind the LayerNorm");
  PRV2_REQUIRE(gate_conv_shape_ok(d), "conv3x3_ln_gate: 3x3 s1 p1, cout 256, cin %% 32 == 0, width >= 16, bf16 modes (got %dx%d %d->%d k%d s%d prec %d)",
               d->h, d->w, d->cin, d->cout, d->kh, d->stride, d->prec);
  PRV2_REQUIRE(gate_w_packed || (!mul && !gate_bias), "conv3x3_ln_gate: mul / gate_bias belong to the gate stage");
  PRV2_REQUIRE(!gate_w_packed || d->act == PRV2_ACT_RELU || d->act == PRV2_ACT_NONE, "conv3x3_ln_gate: ReLU or no activation in front of the gate (act %d)", d->act);
  GateConvParams gp;
  memset(&gp, 0, sizeof(gp));
  IgemmParams& p = gp.c;
  p.x = x; p.w = w_packed; p.bias = bias; p.ln_w = ln_weight; p.ln_b = ln_bias; p.ln_eps = d->ln_eps; p.mul = mul; p.res = res; p.y = y;
  p.N = d->n; p.H = d->h; p.W = d->w; p.OH = d->h; p.OW = d->w;
  p.Cin = d->cin; p.Cin_pad = d->cin; p.Cout = d->cout; p.Ncols = d->cout;
  p.KH = 3; p.KW = 3; p.stride = 1; p.pad = 1; p.pad_x = 1;
  p.ldx = d->ldx; p.ldy = d->ldy; p.ld_mul = d->ld_mul; p.ld_res = d->ld_res;
  p.x_bstride = d->x_bstride ? d->x_bstride : (long long)d->h * d->w * d->ldx;
  p.y_bstride = d->y_bstride ? d->y_bstride : (long long)d->h * d->w * d->ldy;
  p.M = (long long)d->n * d->h * d->w;
  p.relu_in = d->relu_in; p.act = d->act;
  PRV2_REQUIRE(d->n > 0 && d->h > 0 && d->ldx >= d->cin && d->ldx % 4 == 0 && aligned16(x) && p.x_bstride % 4 == 0, "conv3x3_ln_gate: x layout");
  PRV2_REQUIRE(d->ldy >= d->cout && d->ldy % 4 == 0 && aligned16(y) && p.y_bstride % 4 == 0, "conv3x3_ln_gate: y layout");
  const long long px = (long long)d->h * d->w;
  PRV2_REQUIRE(!mul || (d->ld_mul >= d->cout && d->ld_mul % 4 == 0 && aligned16(mul) && px * d->ld_mul < (1LL << 29)), "conv3x3_ln_gate: mul layout");
  PRV2_REQUIRE(!res || (d->ld_res >= d->cout && d->ld_res % 4 == 0 && aligned16(res) && px * d->ld_res < (1LL << 29)), "conv3x3_ln_gate: res layout");
  p.vec_ok = 1; p.vec_epi = 1;
  gp.gate_w = gate_w_packed;
  gp.gate_bias = gate_bias;
  gp.pre = pre;
  gp.ld_pre = ld_pre;
  gp.x_x2 = (d->fmt & PRV2_FMT_X_X2) != 0;
  gp.mul_x2 = (d->fmt & PRV2_FMT_MUL_X2) != 0;
  gp.y_x2 = (d->fmt & PRV2_FMT_Y_X2) != 0;
  PRV2_REQUIRE(!(d->fmt & ~(PRV2_FMT_X_X2 | PRV2_FMT_MUL_X2 | PRV2_FMT_Y_X2)), "conv3x3_ln_gate: unknown format bits %d", d->fmt);
  PRV2_REQUIRE(!gp.x_x2 || (gate_w_packed && d->cin % 8 == 0 && !d->relu_in), "conv3x3_ln_gate: a pre-split (X2) input is taken by the gate kernel only (no input ReLU)");
  PRV2_REQUIRE(!gate_w_packed || !mul || gp.mul_x2 == gp.x_x2, "conv3x3_ln_gate: the gate kernel takes mul in the format of x (PRV2_FMT_X_X2 and PRV2_FMT_MUL_X2 go together)");
  PRV2_REQUIRE(!gp.mul_x2 || (gate_w_packed && mul), "conv3x3_ln_gate: PRV2_FMT_MUL_X2 without a gate stage / mul operand");
  PRV2_REQUIRE(!gp.y_x2 || (!gate_w_packed && !ln_weight), "conv3x3_ln_gate: a pre-split (X2) output is written by the plain 256-column conv (no LayerNorm, no gate)");
#ifdef PRV2_GATE_STAMPS
  gp.stamps = getenv("PRV2_STAMP_PTR") ? (long long*)strtoull(getenv("PRV2_STAMP_PTR"), nullptr, 16) : nullptr;
#endif
  const int64_t blocks = (int64_t)d->n * cdiv(d->h, g256::TH) * cdiv(d->w, g256::TW);
  PRV2_REQUIRE(blocks < (1LL << 31), "conv3x3_ln_gate: grid too large");
  hipStream_t s = (hipStream_t)stream;
  const bool x3 = d->prec == PRV2_PREC_BF16X3;
  // PRV2_W4=1: the four-wave kernel of conv3x3_w4.h (two workgroups per CU).  NOT the default: it needs 7 % fewer shader cycles per tile
  // (all-zero operands, where the chip holds 2.39 GHz: 646 vs 607 TF), but on real operands the chip is at its power limit -- the clock
  // falls from 1.94 to 1.81 GHz and the launch takes the same 3.48 ms; inside a frame (two streams) it is 1 % slower
  // (tools/probes/gate_clock.sh, profiles/r03_power_wall.txt)
#ifdef PRV2_EXPERIMENTS
  static const int use_w4 = getenv("PRV2_W4") ? atoi(getenv("PRV2_W4")) : 0;
  if (gate_w_packed && gp.x_x2 && x3 && use_w4 && (!mul || gp.mul_x2)) {
    hipLaunchKernelGGL((conv3x3_w4_gate_kernel<PRV2_PREC_BF16X3>), dim3((unsigned)blocks), dim3(256), 0, s, gp);
    set_kernel("conv3x3_w4_gate_kernel", 256, d->prec);
    PRV2_LAUNCH_CHECK("conv3x3_ln_gate");
    return 0;
  }
#endif
  if (gate_w_packed && gp.x_x2) {
    if (x3) hipLaunchKernelGGL((conv3x3_c256_gate_x2_kernel<256, PRV2_PREC_BF16X3>), dim3((unsigned)blocks), dim3(512), 0, s, gp);
    else hipLaunchKernelGGL((conv3x3_c256_gate_x2_kernel<256, PRV2_PREC_BF16>), dim3((unsigned)blocks), dim3(512), 0, s, gp);
    set_kernel("conv3x3_c256_gate_x2_kernel", 256, d->prec);
    PRV2_LAUNCH_CHECK("conv3x3_ln_gate");
    return 0;
  }
  if (gate_w_packed) {
    if (x3) hipLaunchKernelGGL((conv3x3_c256_gate_kernel<256, PRV2_PREC_BF16X3>), dim3((unsigned)blocks), dim3(512), 0, s, gp);
    else hipLaunchKernelGGL((conv3x3_c256_gate_kernel<256, PRV2_PREC_BF16>), dim3((unsigned)blocks), dim3(512), 0, s, gp);
  } else {
    if (x3) hipLaunchKernelGGL((conv3x3_c256_kernel<256, PRV2_PREC_BF16X3>), dim3((unsigned)blocks), dim3(512), 0, s, gp);
    else hipLaunchKernelGGL((conv3x3_c256_kernel<256, PRV2_PREC_BF16>), dim3((unsigned)blocks), dim3(512), 0, s, gp);
  }
  set_kernel(gate_w_packed ? "conv3x3_c256_gate_kernel" : "conv3x3_c256_kernel", 256, d->prec);
  PRV2_LAUNCH_CHECK("conv3x3_ln_gate");
  return 0;
}
