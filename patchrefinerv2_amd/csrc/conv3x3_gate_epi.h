// Shared between conv3x3_gate.hip (bf16x3 main loop) and conv3x3_f6.hip (fp16 + fp6 main loop): the 8 x 16-pixel x 256-channel tile's constants,
// the kernel parameters and the EPILOGUE of the 256-column kernels (C tile -> LDS, [+ pre], [LayerNorm,] gate GEMM + final stage / stores).
// The text of the epilogue is conv3x3_gate.hip's of rounds 2-4, moved here unchanged in round 5.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "igemm.h"

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
  float f6_x_scale, f6_out_scale;  // conv3x3_c256_gate_f6_kernel (conv3x3_f6.hip): the power-of-two operand scales of the fp16 + fp6 main loop
  unsigned* f6_range;              // ... and its range word (or null)
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

// The epilogue of the 256-column kernels, from the accumulators on: C tile (conv + bias) -> LDS, [+ pre], [LayerNorm statistics,] then the gate
// GEMM + final stage / the pre-split store / the plain store (head of the file).  Shared by the bf16x3 main loop below and the fp16 + fp6 main loop
// of conv3x3_f6.hip (round 5, stage 2): ``write_c_tile(csm)`` stores conv + bias at csm[(tile row * 16 + column) * CLD + channel].  Every buffer
// of the main loop is dead and every wave is behind a barrier when this is entered.
template <int PREC, bool GATE, bool X2IN, class WriteC>
__device__ __forceinline__ void c256_epilogue(const GateConvParams& gp, float* smem, const int n_img, const int y0, const int x0, WriteC&& write_c_tile) {
  using namespace g256;
  const IgemmParams& p = gp.c;
  float* const csm = smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int m16 = lane & 15, g = lane >> 4;
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
  write_c_tile(csm);  // (conv + bias of this workgroup's accumulators: the main loop's own lane layout)
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

}  // namespace prv2
