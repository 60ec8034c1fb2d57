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
#include "conv3x3_gate_epi.h"

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

template <int PREC, bool GATE, bool X2IN = false>
__device__ __forceinline__ void c256_body(const GateConvParams& gp, float* smem) {
  using namespace g256;
  constexpr int NIT = X2IN ? A_IT2 : A_IT;  // halo items per thread per slab
  constexpr int LPI = X2IN ? 2 : 1;         // 16-byte loads per item
  const IgemmParams& p = gp.c;
  char* const As_b = reinterpret_cast<char*>(smem);
  char* const Bs_b = As_b + 2 * A_BYTES;

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

  c256_epilogue<PREC, GATE, X2IN>(gp, smem, n_img, y0, x0, [&](float* ct) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = wn * 64 + j * 16 + m16;
      const float b = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) ct[((4 * wm + a) * TW + 4 * g + e) * CLD + col] = acc[a][j][e] + b;
    }
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
