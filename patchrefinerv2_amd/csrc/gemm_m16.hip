// 1x1 convolution / linear layer as a row-major GEMM  Y[M, N] = X[M, K] W^T, bf16 modes, v_mfma_f32_16x16x32_bf16.
//
// These layers (the 256->256 sigmoid gates of the GatedConvUnits, the 1x1 out_convs, the ViT linears) have 9x
// less arithmetic per activation element than the 3x3 convs, so the activation path is what has to be cheap:
// there is NO LDS staging of X.  Workgroup = 128 rows x 128 columns, 256 threads = 4 waves (two workgroups per CU
// overlap each other's prologue / epilogue and memory stalls: a 1x1 layer with K = 256 is only 8 steps long); a wave owns 32 rows
// (2 runs of 16) x all 128 columns = 2 x 8 accumulators, and loads exactly its own rows straight from global
// memory into MFMA operand layout (lane 16*g + m: row m, channels 8g..8g+7 of the 32-channel slab = two
// buffer_load_dwordx4 -- rows beyond M and channels beyond K come back as hardware zeros), splits them to bf16
// hi / lo in registers, once, for all 128 columns.  Only the weight tile (128 x 32, the same pre-swizzled
// 128-byte row image as the conv kernels) goes through LDS: each wave fetches 2 KB of it into registers and
// stores it (4 rotating buffers).
//
// The global loads of the loop are compiler-visible buffer loads (__builtin_amdgcn_raw_buffer_load_b128), not the
// inline-asm loads of the conv kernels: here raw rows stay in flight ACROSS the loop back-edge, and a value the
// compiler believes to be ready may be copied (phi / live-range split) before its data has arrived -- an asm load
// is only safe when load, counted wait and use sit in one straight-line region, as in conv3x3_m16.hip.  With no
// LDS-DMA in flight hipcc's own s_waitcnt vmcnt(N) are exact and counted (tools/probes/vmcnt_order.hip: loads
// of one wave retire in issue order on gfx950, LDS-DMA included).
//
// Per 32-channel slab (step s): 8 column phases of 2 x 3 MFMAs; weight column j+2 is read while column j
// multiplies; phase 0 issues the loads of slab s+2 (4 row loads + 2 weight loads); the rows of slab s+1
// (loaded a step ago) are split into the second fragment set in phases 2 / 4, its weights are stored to LDS
// in phase 5; the barrier sits before phase 6; phases 6 / 7 read columns 0 / 1 of the next slab (two phases ahead).
// The loop is unrolled x4 so that register-set parity and weight-buffer index are compile-time.
#include <cstdlib>
#include <type_traits>

#include "igemm.h"

namespace prv2 {

// NW = 4: 128-row tiles, two workgroups per CU (short K: prologue / epilogue of one overlap the loop of the other).
// NW = 8: 256-row tiles, one workgroup per CU -- the weight tile is fetched once for 256 rows instead of once per 128:
// per 32-channel step a CU pulls 32 KB of rows + 16 KB of weights instead of 32 + 32 (1.5 GB goes through L2 in 0.16 ms
// on 768->3072 @ 10 k rows).  Pays only for very long K.  Open: on the MFMA-bound shapes the kernel sits at ~47 % MFMA
// utilisation (300-340 TF); removing either operand stream in an ablation gives +38 %, deeper prefetch of rows, weights
// or weight fragments does not -- the per-element hi/lo split (64 VALU per 48 MFMAs, 9x less reuse than in the 3x3
// kernel) was the suspect, but dropping a quarter of it (PRV2_ABL_NORELU) moves nothing either.  PMC (tools/pmc_conv.sh 19):
// MFMA pipe busy 37.6 % at 2.04 GHz (the 3x3 kernel: 76.7 % at 1.80 GHz on the same box), no LDS bank conflicts, waves
// wait on instruction issue 49 % and on s_waitcnt 25 % of their cycles -- next: TCP / TCC counters (per-CU L2 fetch rate:
// a CU pulls 64 KB per 32-channel step here, 3x the 3x3 kernel).
// BN = 64 (with NW = 4): half-width column tiles for grids that do not cover the chip -- a workgroup's time is its serial K
// loop, so twice the workgroups at half the MFMAs per step is what shortens e.g. the ViT-L 4096 -> 1024 projection at
// 1037 tokens (72 -> 144 workgroups on 256 CUs).
template <int PREC, int NW, int BN = 128>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) gemm16_kernel(const IgemmParams p) {
  static_assert(PREC == PRV2_PREC_BF16X3 || PREC == PRV2_PREC_BF16, "bf16 modes only");
  static_assert((NW == 4 && (BN == 128 || BN == 64)) || (NW == 8 && BN == 128), "4 or 8 waves; 64 columns only with 4 waves");
  constexpr int TM = 32 * NW, NA = 2, NJ = BN / 16, ND = BN / (8 * NW), NBUF = 4;
  constexpr int J_SPLIT0 = NJ == 8 ? 2 : 0, J_SPLIT1 = NJ == 8 ? 4 : 1;  // phases that split the next slab's two pixel runs
  constexpr int B_BYTES = BN * 128;
  constexpr int CLD = BN + 4;
  constexpr int SMEM_MAIN = NBUF * B_BYTES / 4;
  constexpr int SMEM_EPI = TM * CLD + 2 * TM;  // C tile + LN row statistics
  __shared__ __attribute__((aligned(16))) float smem[SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI];
  char* const Bs_b = reinterpret_cast<char*>(smem);

  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {  // XCD-aware: consecutive tiles (sharing weights / neighbouring rows) on one XCD's L2
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_n = bid % p.tiles_n;
  const long long row0 = (long long)(bid / p.tiles_n) * TM;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int m16 = lane & 15, g = lane >> 4;

  // ---- X rows through a buffer resource based at this tile's first row ---------------------------------
  const long long rows_here = p.M - row0 < TM ? p.M - row0 : TM;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.x + row0 * p.ldx), 0, (int)(((rows_here - 1) * p.ldx + ((p.Cin + 3) & ~3)) * 4), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const int cin4 = (p.Cin + 3) & ~3;
  unsigned a_off[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    const int r = wave * 32 + a * 16 + m16;
    a_off[a] = r < rows_here ? (unsigned)((r * p.ldx + g * 8) * 4) : OOB;
  }
  f32x4 ra[4][NA][2];  // raw rows of slabs s+1 .. s+3 in flight (set = slab & 3): HBM latency is ~3 steps under load
  auto load_rows = [&](int set, int cc) {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned off = cc * BK + g * 8 + h * 4 < cin4 ? a_off[a] + (unsigned)((cc * BK + h * 4) * 4) : OOB;
        ra[set][a][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0));
      }
  };
  bf16x8 ah[2][NA], al[2][NA];
  const int relu_floor = p.relu_in ? 0 : (int)0x80000000;
  auto split_run4 = [&](int rset, int set, int a) {  // 8 channels of one row: ReLU-in, hi = RNE bf16, lo = RNE bf16 of the rest
    typedef int i32x4v __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 hw, lw;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      i32x4v vi = __builtin_bit_cast(i32x4v, ra[rset][a][h]);
#ifndef PRV2_ABL_NORELU
      vi.x = max(vi.x, relu_floor);
      vi.y = max(vi.y, relu_floor);
      vi.z = max(vi.z, relu_floor);
      vi.w = max(vi.w, relu_floor);
#endif
      const f32x4 v = __builtin_bit_cast(f32x4, vi);
      const bf16x4 hi = __builtin_convertvector(v, bf16x4);
      const u32x2 hp = __builtin_bit_cast(u32x2, hi);
      f32x4 hf;
      hf.x = __builtin_bit_cast(float, hp.x << 16);
      hf.y = __builtin_bit_cast(float, hp.x & 0xffff0000u);
      hf.z = __builtin_bit_cast(float, hp.y << 16);
      hf.w = __builtin_bit_cast(float, hp.y & 0xffff0000u);
      const u32x2 lp = __builtin_bit_cast(u32x2, __builtin_convertvector(v - hf, bf16x4));
      hw[2 * h] = hp.x;
      hw[2 * h + 1] = hp.y;
      lw[2 * h] = lp.x;
      lw[2 * h + 1] = lp.y;
    }
    ah[set][a] = __builtin_bit_cast(bf16x8, hw);
    al[set][a] = __builtin_bit_cast(bf16x8, lw);
  };

  // ---- weight tiles: LDS-DMA, 8 rows x 128 B per instruction ---------------------------------------------
  const int cchunks = p.Cin_pad / BK;
  const long long w_row_stride = p.Cin_pad;  // 1 tap
  const int dma_row = lane >> 3, dma_slot = lane & 7;
  const float* wdma = reinterpret_cast<const float*>(p.w) + ((long long)tile_n * BN + dma_row) * w_row_stride + dma_slot * 4;
  auto dma_src = [&](int s, int i) { return wdma + (long long)s * BK + (long long)((wave * ND + i) * 8) * w_row_stride; };
  auto dma_dst = [&](int bbuf, int i) { return Bs_b + bbuf * B_BYTES + (wave * ND + i) * 1024; };
  auto dma_b = [&](int s, int bbuf, int i) {  // prologue: compiler-visible
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)dma_src(s, i),
                                     (__attribute__((address_space(3))) void*)dma_dst(bbuf, i), 16, 0, 0);
  };
  // main loop: 8 rows x 128 B per instruction into registers (same lane -> byte mapping as the DMA), stored to LDS later
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, 0x7ffffff0, 0x00020000);
  const unsigned wdma_off = (unsigned)((((long long)tile_n * BN + dma_row) * w_row_stride + dma_slot * 4) * 4);
  f32x4 rb[2][ND];  // weight pieces of steps s+1 / s+2 (set = step parity; a third step ahead bought nothing)
  auto load_w = [&](int set, int s) {
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const unsigned off = wdma_off + (unsigned)(((long long)s * BK + (long long)((wave * ND + i) * 8) * w_row_stride) * 4);
      rb[set][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)off, 0, 0));
    }
  };
  auto store_w = [&](int set, int bbuf) {
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      *reinterpret_cast<f32x4*>(dma_dst(bbuf, i) + lane * 16) = rb[set][i];
    }
  };
  const int b_key = (m16 >> 1) & 7;
  const char* const b_lane_hi = Bs_b + m16 * 128 + ((g ^ b_key) << 4);
  const char* const b_lane_lo = Bs_b + m16 * 128 + (((4 + g) ^ b_key) << 4);
  bf16x8 bh[4], bl[4];  // weight column c of a step sits in slot c & 3 (NJ = 8: the next step's column 0 is slot 0 again)
  auto read_b = [&](int slot, int bbuf, int j) {
    bh[slot] = *reinterpret_cast<const bf16x8*>(b_lane_hi + bbuf * B_BYTES + j * 16 * 128);
    if constexpr (PREC == PRV2_PREC_BF16X3) bl[slot] = *reinterpret_cast<const bf16x8*>(b_lane_lo + bbuf * B_BYTES + j * 16 * 128);
  };

  f32x4 acc[NA][NJ];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NP = PREC == PRV2_PREC_BF16X3 ? 3 : 1;
  auto mma = [&](int set, int a, int j, int slot, int pr) {
    if constexpr (PREC == PRV2_PREC_BF16X3) {
      if (pr == 0) acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[set][a], bh[slot], acc[a][j], 0, 0, 0);
      if (pr == 1) acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[set][a], bl[slot], acc[a][j], 0, 0, 0);
      if (pr == 2) acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[set][a], bh[slot], acc[a][j], 0, 0, 0);
    } else {
      acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[set][a], bh[slot], acc[a][j], 0, 0, 0);
    }
  };

  // ---- prologue: slabs 0, 1 (rows) and 0..2 (weights) ------------------------------------------------------
  const int last = cchunks - 1;
  load_rows(0, 0);
  load_rows(1, 1 < cchunks ? 1 : last);
  load_rows(2, 2 < cchunks ? 2 : last);
  load_w(1, 1 < cchunks ? 1 : last);  // stored to buffer 1 during step 0
#pragma unroll
  for (int i = 0; i < ND; ++i) dma_b(0, 0, i);  // (the only LDS-DMA of the kernel: fully drained right below)
#pragma unroll
  for (int a = 0; a < NA; ++a) split_run4(0, 0, a);
  __syncthreads();  // full fence: the weight DMAs have landed
  read_b(0, 0, 0);
  read_b(1, 0, 1);

  // step s: multiplies slab s (fragment set s&1, weight buffer s&3)
  auto step = [&](auto par_c, int s) {
    constexpr int q = decltype(par_c)::value;  // s & 3
    constexpr int set = q & 1;
    const int s2 = s + 2 < cchunks ? s + 2 : last, s3 = s + 3 < cchunks ? s + 3 : last;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      // barrier before phase NJ-2: this wave's LDS store of step s+1's weights (phase NJ-3) and its reads are done; no
      // global-memory condition.  Phases NJ-2 / NJ-1 then read columns 0 / 1 of the NEXT step.
      if (j == NJ - 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // weight columns are read TWO phases ahead: a phase is only 6 MFMAs (96 cycles), less than the LDS latency under load
      if (j + 2 < NJ) read_b((j + 2) & 3, q, j + 2);
      else read_b((j + 2) & 3, (q + 1) & 3, j + 2 - NJ);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) {
        mma(set, 0, j, j & 3, pr);
        mma(set, 1, j, j & 3, pr);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (j == 0) {  // this set's raw registers were consumed during step s-1
#ifndef PRV2_ABL_NOA
        load_rows((q + 3) & 3, s3);  // that set's raw registers were consumed during step s-1
#endif
#ifndef PRV2_ABL_NOB
        load_w(set, s2);
#endif
      }
      // rows of slab s+1 were loaded in step s-2, the weights of step s+1 in step s-1 (hipcc counts the waits itself)
      if (j == J_SPLIT0) {
        split_run4((q + 1) & 3, set ^ 1, 0);
      }
      if (j == J_SPLIT1) {
        split_run4((q + 1) & 3, set ^ 1, 1);
      }
      if (j == NJ - 3) {  // weights of step s+1 -> buffer (s+1)&3, last read in step s-3
#ifndef PRV2_ABL_NOB
        store_w(set ^ 1, (q + 1) & 3);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // Four unconditional steps per iteration: with branches between the steps hipcc's waitcnt pass merges the pending-load
  // state of every path and then drains (vmcnt(3) .. vmcnt(0)) in front of each step's address arithmetic, whose
  // temporaries alias load destinations -- which exposed the full HBM latency on three steps out of four.
  int cc = 0;
  for (; cc + 4 <= cchunks; cc += 4) {
    step(std::integral_constant<int, 0>{}, cc);
    step(std::integral_constant<int, 1>{}, cc + 1);
    step(std::integral_constant<int, 2>{}, cc + 2);
    step(std::integral_constant<int, 3>{}, cc + 3);
  }
  if (cc < cchunks) {  // K % 128 != 0: up to three more slabs
    step(std::integral_constant<int, 0>{}, cc);
    if (cc + 1 < cchunks) {
      step(std::integral_constant<int, 1>{}, cc + 1);
      if (cc + 2 < cchunks) step(std::integral_constant<int, 2>{}, cc + 2);
    }
  }
#ifdef PRV2_ABL_NOEPI
  if (acc[0][0][0] != 12345.f) return;
#endif
  __syncthreads();  // every wave is done with the weight buffers before the C tile overwrites them

  // ---- epilogue through LDS: rows become contiguous 512-byte stores, fused bias / LN / act / gate / residuals ----
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) smem[(wave * 32 + a * 16 + 4 * g + e) * CLD + j * 16 + m16] = acc[a][j][e];
  __syncthreads();
  float* const ln_stats = smem + TM * CLD;
  if (p.ln_w) {  // block-uniform
    ln_row_stats(p, smem, CLD, TM, tid, ln_stats);
    __syncthreads();
  }
  constexpr int C4 = BN / 4;
  constexpr int RPP = NW * 64 / C4;
  const int col4 = tid % C4;
  EpiCols ec;
  if (!epi_cols(p, tile_n * BN + col4 * 4, ec)) return;
  // lean store paths (the general epilogue is ~100 VALU instructions a row, every optional stage a select):
  // bias + act [+ gate] [+ one residual] -- the GatedConvUnit gate sigmoid(.) * out + x and the plain 1x1 / linear layers
  const bool simple = ec.vec && !p.gamma && !p.res2 && !p.ln_w;  // block-uniform
  if (simple) {
    float* const ybase = p.y + row0 * p.ldy + ec.co;
    const float* const mbase = p.mul ? p.mul + row0 * p.ld_mul + ec.co : nullptr;
    const float* const rbase = p.res ? p.res + row0 * p.ld_res + ec.co : nullptr;
    auto lean = [&](auto act_c, auto mul_c, auto res_c) {
      constexpr bool MUL = decltype(mul_c)::value, RES = decltype(res_c)::value;
      for (int rr = tid / C4; rr < TM && rr < rows_here; rr += RPP) {
        const f32x4 cv = *reinterpret_cast<const f32x4*>(&smem[rr * CLD + col4 * 4]);
        f32x4 mv = {1.f, 1.f, 1.f, 1.f}, rv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (MUL) mv = *reinterpret_cast<const f32x4*>(mbase + (unsigned)(rr * p.ld_mul));
        if constexpr (RES) rv = *reinterpret_cast<const f32x4*>(rbase + (unsigned)(rr * p.ld_res));
        f32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = PREC == PRV2_PREC_F32 ? act_apply(cv[e] + ec.bias[e], decltype(act_c)::value) : act_apply_bf(cv[e] + ec.bias[e], decltype(act_c)::value);  // (bf16 modes: gelu_fast, as gemm_ss.hip)
          if constexpr (MUL) t = mv[e] * t;
          if constexpr (RES) t += rv[e];
          ov[e] = e < ec.nvalid ? t : 0.f;  // pad channels behind cout stay zero
        }
        float* dst = ybase + (unsigned)(rr * p.ldy);
        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(ov) : "memory");
      }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    dispatch_act(p.act, [&](auto act_c) {
      if (p.mul && p.res) lean(act_c, T_{}, T_{});
      else if (p.mul) lean(act_c, T_{}, F_{});
      else if (p.res) lean(act_c, F_{}, T_{});
      else lean(act_c, F_{}, F_{});
    });
    return;
  }
  dispatch_act(p.act, [&](auto act_c) {
    for (int rr = tid / C4; rr < TM && rr < rows_here; rr += RPP) {
      const f32x4 cv = *reinterpret_cast<const f32x4*>(&smem[rr * CLD + col4 * 4]);
      const long long m = row0 + rr;
      epi_store<decltype(act_c)::value>(p, ec, cv, m, m * p.ldy + ec.co, ln_stats[rr], ln_stats[TM + rr]);
    }
  });
}

// dense row-major operands only (every 1x1 / linear call of the path): otherwise the generic kernel takes it
bool gemm16_supported(const IgemmParams& p, int prec) {
  return prec != PRV2_PREC_F32 && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.convt_k == 0 &&
         p.x_bstride == (long long)p.H * p.W * p.ldx && p.y_bstride == (long long)p.OH * p.OW * p.ldy &&
         256LL * p.ldx * 4 < (1LL << 31) && p.Ncols > 64;  // (no row-count condition: the choice must not depend on the batch)
}

void launch_gemm16(IgemmParams& p, int prec, hipStream_t s) {
  static const int force_nw = getenv("PRV2_GEMM16_NW") ? atoi(getenv("PRV2_GEMM16_NW")) : 0;      // A/B switches
  static const int force_bn = getenv("PRV2_GEMM16_BN") ? atoi(getenv("PRV2_GEMM16_BN")) : 0;
  const long long tiles128 = cdiv(p.M, 128) * cdiv(p.Ncols, 128);
  // 64-column tiles when 128 x 128 tiles leave more than a third of the CUs without a workgroup
  // (not with a fused LayerNorm: its row statistics need all <= 128 channels of a pixel in one workgroup tile)
  const bool narrow = !p.ln_w && (force_bn ? force_bn == 64 : tiles128 <= 160);
  // 256-row tiles when K is long enough to amortise the lone workgroup's prologue / epilogue and the grid still covers the chip
  const bool big = !narrow && (force_nw ? force_nw == 8 : (p.Cin_pad >= 2048 && cdiv(p.M, 256) * cdiv(p.Ncols, 128) >= 256));  // +6 % at K = 3072, -2 % at 768
  p.tiles_n = (int)cdiv(p.Ncols, narrow ? 64 : 128);
  const int tiles_m = (int)cdiv(p.M, big ? 256 : 128);
  const dim3 grid(tiles_m * p.tiles_n);
  set_kernel("gemm16_kernel", narrow ? 64 : (big ? 256 : 128), prec);
  if (narrow) {
    if (prec == PRV2_PREC_BF16X3) hipLaunchKernelGGL((gemm16_kernel<PRV2_PREC_BF16X3, 4, 64>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm16_kernel<PRV2_PREC_BF16, 4, 64>), grid, dim3(256), 0, s, p);
  } else if (big) {
    if (prec == PRV2_PREC_BF16X3) hipLaunchKernelGGL((gemm16_kernel<PRV2_PREC_BF16X3, 8>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((gemm16_kernel<PRV2_PREC_BF16, 8>), grid, dim3(512), 0, s, p);
  } else {
    if (prec == PRV2_PREC_BF16X3) hipLaunchKernelGGL((gemm16_kernel<PRV2_PREC_BF16X3, 4>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm16_kernel<PRV2_PREC_BF16, 4>), grid, dim3(256), 0, s, p);
  }
}

}  // namespace prv2
