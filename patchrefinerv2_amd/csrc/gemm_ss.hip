// Dense linear layer on PRE-SPLIT activations:  Y[M, N] = epilogue(A[M, K] W^T), bf16x3, v_mfma_f32_16x16x32_bf16.
//
// The ViT blocks (external/depth_anything_v2/dinov2_layers/attention.py:44,46, mlp.py:30,32; MiDaS BEiT blocks) are chains
// LayerNorm -> Linear -> attention -> Linear -> LayerNorm -> Linear -> GELU -> Linear in which every Linear input has exactly
// one consumer.  The producers therefore write their result directly in the operand format of the matrix pipe -- the
// "split-swizzled" row image the packed weights already use (igemm.hip::pack_weight_kernel): per 32 channels one 128-byte
// group [4 x 16 B bf16 hi | 4 x 16 B bf16 lo], 16-byte slot c stored at c ^ ((row >> 1) & 7) -- same bytes per element as fp32,
// no conversion arithmetic left in the GEMM.  Both operands then reach LDS by LDS-DMA (global_load_lds_dwordx4: 8 rows x 128 B
// per wave instruction, linear copy == conflict-free image) and the main loop is ds_read_b128 + MFMA only:
//   workgroup = 256 x 256 outputs, 8 waves as 2 (rows) x 4 (columns), wave = 128 x 64 = 8 x 4 accumulators of 16 x 16;
//   per 32-channel slab: 32 KB of A + 32 KB of B in one of two LDS stages, 96 MFMAs per wave (lo*hi, hi*lo, hi*hi);
//   slab k+1 is in flight (DMA) while slab k multiplies; one s_waitcnt vmcnt(0) + s_barrier per slab orders
//   "DMA landed" -> "everyone may read" and "everyone has read" -> "DMA may overwrite".
// The DMA and its waits are inline asm: while hipcc sees an LDS-DMA in flight it drains vmcnt / lgkmcnt at every LDS
// access (conv3x3_m16.hip has the full story); what asm issues, asm waits for.
// Arithmetic is IDENTICAL to gemm16_kernel (same split, same three products in the same order, same MFMA, same epilogue
// formula): the two kernels give bit-equal results (tests/test_hip_ops.py::test_gemm_ss_bit_equal_to_gemm16), so the host may
// choose between them by problem size without making results depend on the batch.
// GELU (fc1) is gelu_fast (common.h: |error| 4.7e-7 against float64, the exact-erf formula's own fp32 error; ~14 instead of ~45 instructions) in both
// kernels since round 6: the fc1 store loop was VALU-bound (381 -> 362 us at 14 350 x 4096, profiles/r06_experiments.txt #3).
// TM x TN = 256 x 256 (rows >= 4096), 128 x 128 (4 waves as 2 x 2, wave = 64 x 64; two workgroups per CU) or 64 x 64 (grids of one image).
#include <cstdlib>

#include "igemm.h"

namespace prv2 {

struct GemmSSParams {
  const char* a;  // split-swizzled activations: row r, slab k at a + r * lda + k * 128
  long long lda;  // bytes
  const char* w;  // packed weights, one tap: row n, slab k at w + n * ldw + k * 128
  long long ldw;
  long long M;
  int N;       // valid GEMM columns
  int w_rows;  // rows present in w (N rounded up to 128)
  int kslabs;
  const float* bias;
  const float* gamma;
  const float* res;
  int ld_res;
  float* y;  // fp32 output rows (y_ss == nullptr)
  int ldy;
  char* y_ss;  // split-swizzled output rows
  long long ldy_ss;
  int act;
  int tiles_n, tiles_m, blocked;
  int vgrid;         // persistent form: tile ids to walk (ids behind the tiles idle, as the one-tile-per-workgroup grid's do)
  int scale_cols;    // split output: columns [0, scale_cols) are multiplied by ``scale`` behind bias / activation (the q third of a
  float scale;       // qkv Linear: hd^-0.5 log2 e, what qkv_split_kernel used to apply -- attention.hip reads the rows as they are)
};

// NS = LDS stages.  2: slab k + 1 in flight while slab k multiplies (what the LDS allows for 256 x 256 tiles and for two 128 x 128
// workgroups per CU).  4 (clamped issue, counted waits): the 64 x 64 tiles of grids with at most one workgroup per CU.
// DEFER (NS == 2): the MFMAs of a slab's LAST row block are issued behind the next slab's barrier, DMA issue and first fragment
// reads -- their operands are registers by then -- so that the matrix pipe has work while the waves of the workgroup re-converge
// (both waves of a SIMD arrive at the barrier together: without it the pipe idles for the barrier skew + 8 DMA issues + the LDS
// latency of the first fragments, ~15 % of a slab).  Every accumulator still receives its slabs in order: same bits.
template <int WM, int WN, int RI, int RJ, bool OUT_SS, int ACT, int NS = 2, bool DEFER = false>
__global__ void __launch_bounds__(WM * WN * 64, (WM * WN == 8 || NS > 2) ? 1 : 2) gemm_ss_kernel(const GemmSSParams p) {
  constexpr int NW = WM * WN, TM = WM * RI * 16, TN = WN * RJ * 16;
  constexpr int A_BYTES = TM * 128, B_BYTES = TN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_DMA = TM / 8 / NW, B_DMA = TN / 8 / NW;  // 1 KB pieces per wave and slab
  constexpr int STRIP_LD = RJ * 16 + 4, STRIP_BYTES = 16 * STRIP_LD * 4;
  static_assert(NW * STRIP_BYTES <= NS * STAGE && NS * STAGE <= 160 * 1024, "epilogue strips fit in the staging memory");
  __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE];

  int bid = blockIdx.x;
  {  // XCD-aware: consecutive tiles (sharing rows / weights) on one XCD's L2 (blocks b, b + 8, ... share an XCD)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // The ~32 tiles an XCD has in flight (one per CU, consecutive ids) step through K together: a slab of A rows / W columns
  // is fetched into that XCD's L2 once and DMA'd by every tile that shares it.  4 x 8 blocks of tiles share 12 such slabs
  // per step instead of the 18 of a 2 x 16 run (wide layers) -- p.blocked: tiles_n % 8 == 0; ids behind tiles_m are idle
  int tile_m, tile_n;
  if (p.blocked) {
    const int blk = bid >> 5, within = bid & 31, bpr = p.tiles_n >> 3;
    tile_m = (blk / bpr) * 4 + (within >> 3);
    tile_n = (blk % bpr) * 8 + (within & 7);
    if (tile_m >= p.tiles_m) return;
  } else {
    tile_n = bid % p.tiles_n;
    tile_m = bid / p.tiles_n;
  }
  const long long row0 = (long long)tile_m * TM;
  const int col0 = tile_n * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m16 = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;

  // ---- LDS-DMA sources: piece i of this wave = rows (wave * X_DMA + i) * 8 + (lane >> 3), 16-byte slot lane & 7 -------
  const int dr = lane >> 3, dsl = lane & 7;
  const char* a_src[A_DMA];
  const char* b_src[B_DMA];
#pragma unroll
  for (int i = 0; i < A_DMA; ++i) {
    long long r = row0 + (wave * A_DMA + i) * 8 + dr;
    r = r < p.M ? r : p.M - 1;  // rows behind M: any valid row (their outputs are never stored)
    a_src[i] = p.a + r * p.lda + dsl * 16;
  }
#pragma unroll
  for (int i = 0; i < B_DMA; ++i) {
    int n = col0 + (wave * B_DMA + i) * 8 + dr;
    n = n < p.w_rows ? n : p.w_rows - 1;
    b_src[i] = p.w + (long long)n * p.ldw + dsl * 16;
  }
  const unsigned smem_base = (unsigned)(size_t)smem;
  auto dma = [&](const char* src, unsigned dst) {
    const unsigned d = __builtin_amdgcn_readfirstlane(dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(d), "v"(src) : "memory");
  };
  auto issue = [&](int stage, int k) {
    const unsigned sb = smem_base + stage * STAGE;
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) dma(a_src[i] + (long long)k * 128, sb + (wave * A_DMA + i) * 1024);
#pragma unroll
    for (int i = 0; i < B_DMA; ++i) dma(b_src[i] + (long long)k * 128, sb + A_BYTES + (wave * B_DMA + i) * 1024);
  };

  // ---- fragments: lane (m16, g) reads row m16 of a 16-row block, logical slot g (hi) / 4 + g (lo) ---------------------
  const int key = (m16 >> 1) & 7;
  const int a_off_hi = (wm * RI * 16 + m16) * 128 + ((g ^ key) << 4), a_off_lo = (wm * RI * 16 + m16) * 128 + (((4 + g) ^ key) << 4);
  const int b_off_hi = A_BYTES + (wn * RJ * 16 + m16) * 128 + ((g ^ key) << 4);
  const int b_off_lo = A_BYTES + (wn * RJ * 16 + m16) * 128 + (((4 + g) ^ key) << 4);

  f32x4 acc[RI][RJ];
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int j = 0; j < RJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int stage) {
    const char* const sb = smem + stage * STAGE;
    bf16x8 bh[RJ], bl[RJ], ah[2], al[2];
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
      bh[j] = *reinterpret_cast<const bf16x8*>(sb + b_off_hi + j * 2048);
      bl[j] = *reinterpret_cast<const bf16x8*>(sb + b_off_lo + j * 2048);
    }
    ah[0] = *reinterpret_cast<const bf16x8*>(sb + a_off_hi);
    al[0] = *reinterpret_cast<const bf16x8*>(sb + a_off_lo);
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      if (i + 1 < RI) {  // the next row block's fragments travel while this one multiplies
        ah[(i + 1) & 1] = *reinterpret_cast<const bf16x8*>(sb + a_off_hi + (i + 1) * 2048);
        al[(i + 1) & 1] = *reinterpret_cast<const bf16x8*>(sb + a_off_lo + (i + 1) * 2048);
      }
#ifndef PRV2_GSS_NOSCHED
      // (fences: without them hipcc sinks the two reads above to just in front of their first use -- `ds_read; s_waitcnt lgkmcnt(1); v_mfma`
      //  on the value just requested: an LDS round trip of idle matrix pipe per row block, ~1 k of a slab's 1.5 k MFMA cycles per wave)
      __builtin_amdgcn_sched_barrier(0);
#endif
      // smallest terms first (as gemm16_kernel): lo*hi, hi*lo, hi*hi; consecutive MFMAs never share an accumulator
#pragma unroll
      for (int j = 0; j < RJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i & 1], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < RJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i & 1], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < RJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i & 1], bh[j], acc[i][j], 0, 0, 0);
#ifndef PRV2_GSS_NOSCHED
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };

  if constexpr (NS == 2 && DEFER) {
    bf16x8 dbh[RJ], dbl[RJ], dah, dal;  // operands of the deferred row block (RI - 1) of the previous slab
    auto mma_last = [&]() {
#pragma unroll
      for (int j = 0; j < RJ; ++j) acc[RI - 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dal, dbh[j], acc[RI - 1][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < RJ; ++j) acc[RI - 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dah, dbl[j], acc[RI - 1][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < RJ; ++j) acc[RI - 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dah, dbh[j], acc[RI - 1][j], 0, 0, 0);
    };
    issue(0, 0);
    for (int k = 0; k < p.kslabs; ++k) {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      if (k + 1 < p.kslabs) issue((k + 1) & 1, k + 1);
      const char* const sb = smem + (k & 1) * STAGE;
      bf16x8 bh[RJ], bl[RJ], ah[2], al[2];
#pragma unroll
      for (int j = 0; j < RJ; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(sb + b_off_hi + j * 2048);
        bl[j] = *reinterpret_cast<const bf16x8*>(sb + b_off_lo + j * 2048);
      }
      ah[0] = *reinterpret_cast<const bf16x8*>(sb + a_off_hi);
      al[0] = *reinterpret_cast<const bf16x8*>(sb + a_off_lo);
      if (k > 0) mma_last();  // (registers only: runs while the fragments above travel)
#pragma unroll
      for (int i = 0; i < RI - 1; ++i) {
        ah[(i + 1) & 1] = *reinterpret_cast<const bf16x8*>(sb + a_off_hi + (i + 1) * 2048);
        al[(i + 1) & 1] = *reinterpret_cast<const bf16x8*>(sb + a_off_lo + (i + 1) * 2048);
#pragma unroll
        for (int j = 0; j < RJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i & 1], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < RJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i & 1], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < RJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i & 1], bh[j], acc[i][j], 0, 0, 0);
      }
      dah = ah[(RI - 1) & 1];
      dal = al[(RI - 1) & 1];
#pragma unroll
      for (int j = 0; j < RJ; ++j) {
        dbh[j] = bh[j];
        dbl[j] = bl[j];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the deferred operands are in registers before anyone may overwrite the stage)
    }
    mma_last();
    asm volatile("s_barrier" ::: "memory");
  } else if constexpr (NS == 2) {
    issue(0, 0);
    for (int k = 0; k < p.kslabs; ++k) {
      // my DMAs of slab k have landed; behind the barrier everyone's have, and everyone is done reading the other stage
#ifdef PRV2_GSS_NOBAR  // timing ablation (results wrong): no workgroup barrier in the K loop
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#endif
#ifdef PRV2_GSS_NODMA  // timing ablation (results wrong; profiles/r03_experiments.txt): only the first two slabs are ever fetched
      if (k + 1 < p.kslabs && k < 1) issue((k + 1) & 1, k + 1);
#else
      if (k + 1 < p.kslabs) issue((k + 1) & 1, k + 1);
#endif
#ifndef PRV2_GSS_NOMMA  // timing ablation (results wrong): the DMA stream alone
      compute(k & 1);
#endif
    }
    asm volatile("s_barrier" ::: "memory");  // the staging memory becomes the epilogue strips
  } else {
    // NS - 1 slabs in flight; behind the last slab the issue is clamped (re-fetches of the last slab into stages nobody reads any
    // more), so that "slab k has landed" is the same counted wait in every iteration
    const int last = p.kslabs - 1;
#pragma unroll
    for (int k0 = 0; k0 < NS - 1; ++k0) issue(k0, k0 < last ? k0 : last);
    for (int k = 0; k < p.kslabs; ++k) {
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * (A_DMA + B_DMA)) : "memory");
      const int kn = k + NS - 1;
      issue(kn % NS, kn < last ? kn : last);  // the stage of slab k - 1: everyone is behind this iteration's barrier, i.e. done with it
      compute(k % NS);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // nothing may land on the epilogue strips
  }

#ifdef PRV2_GSS_NOEPI  // timing ablation (results wrong): the K loop alone -- one store per lane keeps the accumulators alive
  {
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int j = 0; j < RJ; ++j) sum += acc[i][j];
    if (sum[0] == 12345.678f) p.y[threadIdx.x] = sum[1] + sum[2] + sum[3];
    return;
  }
#endif
  // ---- epilogue: one 16-row block at a time through a wave-private LDS strip -> whole 256-byte row segments ----------
  float* const strip = reinterpret_cast<float*>(smem + wave * STRIP_BYTES);
  const long long wrow0 = row0 + wm * RI * 16;
  const int wcol0 = col0 + wn * RJ * 16;
#pragma unroll
  for (int i = 0; i < RI; ++i) {  // (unrolled: acc is indexed by compile-time constants only; it never enters the lambda below)
#pragma unroll
    for (int j = 0; j < RJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) strip[(4 * g + e) * STRIP_LD + j * 16 + m16] = acc[i][j][e];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    {  // (the activation is a template parameter: with a run-time dispatch around this block hipcc spilled a third of the
       //  accumulators to scratch at the loop exit -- 2.9x the output bytes in WRITE_SIZE / FETCH_SIZE)
      if constexpr (!OUT_SS) {
        constexpr int LPR = RJ * 4, RPI = 64 / LPR;  // lanes per row (16 bytes each), rows per instruction
#pragma unroll
        for (int q = 0; q < 16 / RPI; ++q) {
          const int rl = q * RPI + lane / LPR, c4 = (lane % LPR) * 4;
          const long long grow = wrow0 + i * 16 + rl;
          const int gcol = wcol0 + c4;
          if (grow < p.M && gcol < p.N) {
            const f32x4 cv = *reinterpret_cast<const f32x4*>(&strip[rl * STRIP_LD + c4]);
            f32x4 bv = {0.f, 0.f, 0.f, 0.f}, gv = {1.f, 1.f, 1.f, 1.f}, rv = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + gcol);
            if (p.gamma) gv = *reinterpret_cast<const f32x4*>(p.gamma + gcol);
            if (p.res) rv = *reinterpret_cast<const f32x4*>(p.res + grow * p.ld_res + gcol);
            f32x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = act_apply_bf(cv[e] + bv[e], ACT);
              if (p.gamma) t *= gv[e];
              if (p.res) t += rv[e];
              ov[e] = t;
            }
            float* dst = p.y + grow * p.ldy + gcol;
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(ov) : "memory");
          }
        }
      } else {
        constexpr int GPR = RJ * 2, RPI = 64 / GPR;  // 8-column groups per row, rows per pass
#pragma unroll
        for (int q = 0; q < 16 / RPI; ++q) {
          const int rl = q * RPI + lane / GPR, kg = lane % GPR;
          const long long grow = wrow0 + i * 16 + rl;
          const int gcol = wcol0 + kg * 8;
          if (grow < p.M && gcol < p.N) {
            const float* sp = &strip[rl * STRIP_LD + kg * 8];
            f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
            if (p.bias) {
              const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + gcol), b1 = *reinterpret_cast<const f32x4*>(p.bias + gcol + 4);
              v0 += b0;
              v1 += b1;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v0[e] = act_apply_bf(v0[e], ACT);
              v1[e] = act_apply_bf(v1[e], ACT);
            }
            if (gcol < p.scale_cols) {  // (scale_cols is a multiple of 8: a group is scaled whole)
              v0 *= p.scale;
              v1 *= p.scale;
            }
            bf16x4 h0, l0, h1, l1;
            split_bf16(v0, h0, l0);
            split_bf16(v1, h1, l1);
            const bf16x8 hv = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            const bf16x8 lv = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            char* const rowp = p.y_ss + grow * p.ldy_ss + (gcol >> 5) * 128;
            const int chunk = (gcol & 31) >> 3, k2 = (int)((grow >> 1) & 7);
            *reinterpret_cast<bf16x8*>(rowp + ((chunk ^ k2) << 4)) = hv;
            *reinterpret_cast<bf16x8*>(rowp + (((4 + chunk) ^ k2) << 4)) = lv;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // the strip is rewritten by the next row block
  }
}

// ---- the 256 x 256 tile as a PERSISTENT workgroup (one per CU) -------------------------------------------------------------------
// Same tile, same slab order, same three products per slab in the same order, same epilogue arithmetic as gemm_ss_kernel<2, 4, 8, 4>: bit-equal
// results.  What changes is what the matrix pipe waits for:
//  * a workgroup walks its tiles v = blockIdx.x, + gridDim.x, ... of the same XCD-chunked / 4 x 8-blocked order; the LAST slab of a tile issues
//    the DMA of the NEXT tile's first slab into the stage the epilogue does not use, so the next tile's K loop starts on landed operands (the
//    one-tile kernel pays a cold L2 / HBM round trip per tile: nothing to overlap it with at one workgroup per CU);
//  * the eight 1 KB LDS-DMA pieces a wave issues per slab are spread BEHIND the row blocks' MFMAs (PPB pieces after each of the first 8 / PPB
//    row blocks) instead of all eight in front of the slab's first MFMA: issuing a piece costs a wave 60-185 cycles (MI355X_MICROARCH.md,
//    "LDS-DMA piece issue cost"), and with all eight waves doing it right behind the barrier the pipe idled ~1 k of a slab's ~6 k cycles
//    (profiles/r04_experiments.txt #9: "operand DMA 15 %").  The stage being filled was released by the barrier at the top of the slab, so a
//    piece may go out at any point of it; early enough that the slab's remaining MFMA time covers its L2 round trip.
// The epilogue strips live in the stage of the tile's last slab (free behind one barrier); the other stage is receiving the next tile.
// The tile's store loop.  A lane's columns are the same in every row block (c4 / kg below depend on the lane only): bias and gamma are fetched
// once per tile, the residual rows of a row block in one batch of loads AHEAD of the block's LDS round trip (the one-tile kernel's loop asks for
// bias, gamma and residual again in each of its 32 passes and drains vmcnt(0) in each: a serial chain of L2 round trips, ~10 us per tile).
template <int RI, int RJ, bool OUT_SS, int ACT>
__device__ __forceinline__ void gss_store_rows(const f32x4 (&acc)[RI][RJ], const GemmSSParams& p, float* const strip, const long long wrow0, const int wcol0,
                                               const int lane) {
  constexpr int STRIP_LD = RJ * 16 + 4;
  const int m16 = lane & 15, g = lane >> 4;
  if constexpr (!OUT_SS) {
    constexpr int LPR = RJ * 4, RPI = 64 / LPR, NQ = 16 / RPI;  // lanes per row (16 bytes each), rows per pass, passes per row block
    const int c4 = (lane % LPR) * 4, r_in = lane / LPR;
    const int gcol = wcol0 + c4;
    const bool col_ok = gcol < p.N;
    const bool has_res = p.res != nullptr, has_gamma = p.gamma != nullptr;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, gv = {1.f, 1.f, 1.f, 1.f};
    if (p.bias && col_ok) bv = *reinterpret_cast<const f32x4*>(p.bias + gcol);
    if (has_gamma && col_ok) gv = *reinterpret_cast<const f32x4*>(p.gamma + gcol);
    f32x4 rv[NQ];
    auto fetch_res = [&](int i) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const long long grow = wrow0 + i * 16 + q * RPI + r_in;
        rv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (grow < p.M && col_ok) rv[q] = *reinterpret_cast<const f32x4*>(p.res + grow * p.ld_res + gcol);
      }
    };
    if (has_res) fetch_res(0);
#pragma unroll
    for (int i = 0; i < RI; ++i) {
#pragma unroll
      for (int j = 0; j < RJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) strip[(4 * g + e) * STRIP_LD + j * 16 + m16] = acc[i][j][e];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      f32x4 ov[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const f32x4 cv = *reinterpret_cast<const f32x4*>(&strip[(q * RPI + r_in) * STRIP_LD + c4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = act_apply_bf(cv[e] + bv[e], ACT);
          if (has_gamma) t *= gv[e];
          if (has_res) t += rv[q][e];
          ov[q][e] = t;
        }
      }
      if (has_res && i + 1 < RI) fetch_res(i + 1);  // (travels under the next row block's LDS round trip)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const long long grow = wrow0 + i * 16 + q * RPI + r_in;
        if (grow < p.M && col_ok) {
          // (a store the compiler sees: its counted vmcnt waits for the prefetched residual rows then leave these stores in flight)
          *reinterpret_cast<f32x4*>(p.y + grow * p.ldy + gcol) = ov[q];
        }
      }
      __builtin_amdgcn_wave_barrier();  // the strip is rewritten by the next row block
    }
  } else {
    constexpr int GPR = RJ * 2, RPI = 64 / GPR, NQ = 16 / RPI;  // 8-column groups per row, rows per pass
    const int kg = lane % GPR, r_in = lane / GPR;
    const int gcol = wcol0 + kg * 8;
    const bool col_ok = gcol < p.N;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (p.bias && col_ok) {
      b0 = *reinterpret_cast<const f32x4*>(p.bias + gcol);
      b1 = *reinterpret_cast<const f32x4*>(p.bias + gcol + 4);
    }
    const float cs = gcol < p.scale_cols ? p.scale : 1.0f;  // (scale_cols is a multiple of 8: a group is scaled whole; x 1.0f is exact)
    const int chunk = (gcol & 31) >> 3;
#pragma unroll
    for (int i = 0; i < RI; ++i) {
#pragma unroll
      for (int j = 0; j < RJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) strip[(4 * g + e) * STRIP_LD + j * 16 + m16] = acc[i][j][e];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int rl = q * RPI + r_in;
        const long long grow = wrow0 + i * 16 + rl;
        const float* sp = &strip[rl * STRIP_LD + kg * 8];
        f32x4 v0 = *reinterpret_cast<const f32x4*>(sp) + b0, v1 = *reinterpret_cast<const f32x4*>(sp + 4) + b1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v0[e] = act_apply_bf(v0[e], ACT);
          v1[e] = act_apply_bf(v1[e], ACT);
        }
        v0 *= cs;
        v1 *= cs;
        bf16x4 h0, l0, h1, l1;
        split_bf16(v0, h0, l0);
        split_bf16(v1, h1, l1);
        if (grow < p.M && col_ok) {
          char* const rowp = p.y_ss + grow * p.ldy_ss + (gcol >> 5) * 128;
          const int k2 = (int)((grow >> 1) & 7);
          *reinterpret_cast<bf16x8*>(rowp + ((chunk ^ k2) << 4)) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
          *reinterpret_cast<bf16x8*>(rowp + (((4 + chunk) ^ k2) << 4)) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// PPB: DMA pieces issued behind each of the first 8 / PPB row blocks (8: all in front of the slab, the one-tile kernel's order)
template <bool OUT_SS, int ACT, int PPB>
__global__ void __launch_bounds__(512, 1) gemm_ss_p_kernel(const GemmSSParams p) {
  constexpr int WN = 4, RI = 8, RJ = 4, TM = 256, TN = 256;
  constexpr int A_BYTES = TM * 128, B_BYTES = TN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int STRIP_LD = RJ * 16 + 4, STRIP_BYTES = 16 * STRIP_LD * 4;
  static_assert(8 * STRIP_BYTES <= STAGE, "the epilogue strips fit in one stage");
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int G = p.vgrid, nwg = gridDim.x;

  auto decode = [&](int v, int& tm, int& tn) -> bool {  // virtual id -> tile (gemm_ss_kernel's map over a grid of G workgroups)
    const int q = G >> 3, r = G & 7, xcd = v & 7;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    if (p.blocked) {
      const int blk = bid >> 5, within = bid & 31, bpr = p.tiles_n >> 3;
      tm = (blk / bpr) * 4 + (within >> 3);
      tn = (blk % bpr) * 8 + (within & 7);
      return tm < p.tiles_m;
    }
    tn = bid % p.tiles_n;
    tm = bid / p.tiles_n;
    return true;
  };
  const int dr = lane >> 3, dsl = lane & 7;
  const char* a_src[4];
  const char* b_src[4];
  auto point = [&](int tm, int tn) {  // this wave's four A and four B pieces of a tile: rows (wave * 4 + i) * 8 + (lane >> 3), 16-byte slot lane & 7
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      long long r = (long long)tm * TM + (wave * 4 + i) * 8 + dr;
      r = r < p.M ? r : p.M - 1;
      a_src[i] = p.a + r * p.lda + dsl * 16;
      int n = tn * TN + (wave * 4 + i) * 8 + dr;
      n = n < p.w_rows ? n : p.w_rows - 1;
      b_src[i] = p.w + (long long)n * p.ldw + dsl * 16;
    }
  };
  const unsigned smem_base = (unsigned)(size_t)smem;
  auto dma = [&](const char* src, unsigned dst) {  // (dst is wave-uniform by construction: scalar registers only)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory");
  };
  auto piece = [&](int j, unsigned sb, long long koff) {
    if (j < 4) dma(a_src[j] + koff, sb + (wave * 4 + j) * 1024);
    else dma(b_src[j - 4] + koff, sb + A_BYTES + (wave * 4 + j - 4) * 1024);
  };

  const int key = (m16 >> 1) & 7;
  const int a_off_hi = (wm * RI * 16 + m16) * 128 + ((g ^ key) << 4), a_off_lo = (wm * RI * 16 + m16) * 128 + (((4 + g) ^ key) << 4);
  const int b_off_hi = A_BYTES + (wn * RJ * 16 + m16) * 128 + ((g ^ key) << 4);
  const int b_off_lo = A_BYTES + (wn * RJ * 16 + m16) * 128 + (((4 + g) ^ key) << 4);

  int v = blockIdx.x, tm = 0, tn = 0;
  while (v < G && !decode(v, tm, tn)) v += nwg;
  if (v >= G) return;
  point(tm, tn);
  int st = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) piece(j, smem_base, 0);

  for (;;) {
    const long long wrow0 = (long long)tm * TM + wm * RI * 16;
    const int wcol0 = tn * TN + wn * RJ * 16;
    f32x4 acc[RI][RJ];
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int j = 0; j < RJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int vn = G, tmn = 0, tnn = 0;
    for (int k = 0; k < p.kslabs; ++k) {
      // my pieces of slab k have landed (and my stores of the previous tile are out); behind the barrier everyone's have, and everyone is done
      // with the other stage (slab k - 1, or the previous tile's strips)
#ifdef PRV2_GSS_NOBAR
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
      // (a counted wait that leaves a full tile's 32 store instructions in flight at k == 0 -- vmcnt(32): the counter retires in issue order --
      //  measured nothing: 238.6 / 239.6 vs 238.8 / 240.3 us on the qkv shape, profiles/r06_experiments.txt #2: the store burst of all CUs'
      //  epilogues together is bound by the memory side, 64 MB per round at ~6 TB/s, not by this wave's wait)
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#endif
      long long koff = (long long)(k + 1) * 128;
      bool has_next = true;
      if (k + 1 == p.kslabs) {  // the last slab issues the NEXT tile's first slab (this tile's sources are no longer needed)
        vn = v + nwg;
        while (vn < G && !decode(vn, tmn, tnn)) vn += nwg;
        has_next = vn < G;
        if (has_next) point(tmn, tnn);
        koff = 0;
      }
#ifdef PRV2_GSS_NODMA
      has_next = has_next && k < 1 && v == (int)blockIdx.x;
#endif
      const char* const sb = smem + st * STAGE;
      const unsigned nb = smem_base + (st ^ 1) * STAGE;
      if (PPB == 8 && has_next) {
#pragma unroll
        for (int j = 0; j < 8; ++j) piece(j, nb, koff);
      }
#ifndef PRV2_GSS_NOMMA
      bf16x8 bh[RJ], bl[RJ], ah[2], al[2];
#pragma unroll
      for (int j = 0; j < RJ; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(sb + b_off_hi + j * 2048);
        bl[j] = *reinterpret_cast<const bf16x8*>(sb + b_off_lo + j * 2048);
      }
      ah[0] = *reinterpret_cast<const bf16x8*>(sb + a_off_hi);
      al[0] = *reinterpret_cast<const bf16x8*>(sb + a_off_lo);
#pragma unroll
      for (int i = 0; i < RI; ++i) {
        if (i + 1 < RI) {
          ah[(i + 1) & 1] = *reinterpret_cast<const bf16x8*>(sb + a_off_hi + (i + 1) * 2048);
          al[(i + 1) & 1] = *reinterpret_cast<const bf16x8*>(sb + a_off_lo + (i + 1) * 2048);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < RJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i & 1], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < RJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i & 1], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < RJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i & 1], bh[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (PPB < 8 && i * PPB < 8 && has_next) {
#pragma unroll
          for (int j = 0; j < PPB; ++j) piece(i * PPB + j, nb, koff);
        }
      }
#endif
      st ^= 1;
    }
    // the tile's last slab sits in stage st ^ 1: once everyone has read it, it carries the epilogue strips (stage st is receiving the next tile)
    asm volatile("s_barrier" ::: "memory");
#ifdef PRV2_GSS_NOEPI
    {
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int j = 0; j < RJ; ++j) sum += acc[i][j];
      if (sum[0] == 12345.678f) p.y[threadIdx.x] = sum[1] + sum[2] + sum[3];
    }
#else
    gss_store_rows<RI, RJ, OUT_SS, ACT>(acc, p, reinterpret_cast<float*>(smem + (st ^ 1) * STAGE + wave * STRIP_BYTES), wrow0, wcol0, lane);
#endif
    if (vn >= G) break;
    v = vn;
    tm = tmn;
    tn = tnn;
  }
}

// fp32 rows -> split-swizzled rows (inputs that no kernel of ours produced: tests, the first layer of a chain)
__global__ void __launch_bounds__(256) split_ss_kernel(const float* __restrict__ x, long long rows, int c, int ldx, char* __restrict__ y,
                                                       long long ldy) {
  const int groups = c >> 3;
  const long long total = rows * groups;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long r = t / groups;
    const int gi = (int)(t - r * groups);
    const float* src = x + r * ldx + gi * 8;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
    bf16x4 h0, l0, h1, l1;
    split_bf16(v0, h0, l0);
    split_bf16(v1, h1, l1);
    char* const rowp = y + r * ldy + (gi >> 2) * 128;
    const int chunk = gi & 3, k2 = (int)((r >> 1) & 7);
    *reinterpret_cast<bf16x8*>(rowp + ((chunk ^ k2) << 4)) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    *reinterpret_cast<bf16x8*>(rowp + (((4 + chunk) ^ k2) << 4)) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

}  // namespace prv2

using namespace prv2;

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int prv2_split_ss(const float* x, int64_t rows, int32_t c, int32_t ldx, void* y_ss, void* stream) {
  PRV2_REQUIRE(x && y_ss && rows > 0 && c > 0 && c % 32 == 0 && ldx >= c && ldx % 4 == 0 && al16(x) && al16(y_ss),
               "split_ss: c must be a multiple of 32, rows 16-byte aligned (rows=%lld c=%d ldx=%d)", (long long)rows, c, ldx);
  hipLaunchKernelGGL(split_ss_kernel, dim3(flat_grid(rows * (c >> 3), 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)rows, c, ldx,
                     reinterpret_cast<char*>(y_ss), (long long)c * 4);
  PRV2_LAUNCH_CHECK("split_ss");
  return 0;
}

template <bool OUT_SS, int ACT>
static void launch_gss_p(int ppb, dim3 grid, hipStream_t s, const GemmSSParams& p) {
  switch (ppb) {
    case 8: hipLaunchKernelGGL((gemm_ss_p_kernel<OUT_SS, ACT, 8>), grid, dim3(512), 0, s, p); break;
    case 4: hipLaunchKernelGGL((gemm_ss_p_kernel<OUT_SS, ACT, 4>), grid, dim3(512), 0, s, p); break;
    case 1: hipLaunchKernelGGL((gemm_ss_p_kernel<OUT_SS, ACT, 1>), grid, dim3(512), 0, s, p); break;
    default: hipLaunchKernelGGL((gemm_ss_p_kernel<OUT_SS, ACT, 2>), grid, dim3(512), 0, s, p); break;
  }
}

static int gemm_ss_impl(const void* a_ss, int64_t m, int32_t k, const void* w_packed, int32_t n, const float* bias, const float* gamma, const float* res,
                        int32_t ld_res, int32_t act, float* y, int32_t ldy, void* y_ss, int32_t scale_cols, float scale, void* stream);

extern "C" int prv2_gemm_ss(const void* a_ss, int64_t m, int32_t k, const void* w_packed, int32_t n, const float* bias,
                            const float* gamma, const float* res, int32_t ld_res, int32_t act, float* y, int32_t ldy, void* y_ss,
                            void* stream) {
  return gemm_ss_impl(a_ss, m, k, w_packed, n, bias, gamma, res, ld_res, act, y, ldy, y_ss, 0, 1.0f, stream);
}

extern "C" int prv2_gemm_ss_qkv(const void* a_ss, int64_t m, int32_t k, const void* w_packed, int32_t n, const float* bias, int32_t q_cols, float q_scale,
                                void* qkv_ss, void* stream) {
  PRV2_REQUIRE(q_cols >= 0 && q_cols <= n && q_cols % 32 == 0, "gemm_ss_qkv: q_cols must be a multiple of 32 within n (q_cols=%d n=%d)", q_cols, n);
  return gemm_ss_impl(a_ss, m, k, w_packed, n, bias, nullptr, nullptr, 0, PRV2_ACT_NONE, nullptr, 0, qkv_ss, q_cols, q_scale, stream);
}

static int gemm_ss_impl(const void* a_ss, int64_t m, int32_t k, const void* w_packed, int32_t n, const float* bias, const float* gamma, const float* res,
                        int32_t ld_res, int32_t act, float* y, int32_t ldy, void* y_ss, int32_t scale_cols, float scale, void* stream) {
  PRV2_REQUIRE(a_ss && w_packed && (y || y_ss) && !(y && y_ss), "gemm_ss: null pointer / exactly one of y, y_ss");
  PRV2_REQUIRE(m > 0 && k > 0 && k % 32 == 0 && n > 0 && n % 8 == 0, "gemm_ss: k must be a multiple of 32, n of 8 (m=%lld k=%d n=%d)",
               (long long)m, k, n);
  PRV2_REQUIRE(al16(a_ss) && al16(w_packed) && (!bias || al16(bias)) && (!gamma || al16(gamma)), "gemm_ss: 16-byte alignment");
  PRV2_REQUIRE(!y || (al16(y) && ldy % 4 == 0 && ldy >= n), "gemm_ss: y rows must be 16-byte aligned (ldy=%d)", ldy);
  PRV2_REQUIRE(!res || (y && al16(res) && ld_res % 4 == 0 && ld_res >= n), "gemm_ss: res needs an fp32 output and aligned rows");
  PRV2_REQUIRE(!y_ss || (al16(y_ss) && n % 32 == 0 && !gamma), "gemm_ss: a split output needs n %% 32 == 0 and no gamma");
  GemmSSParams p = {};
  p.a = reinterpret_cast<const char*>(a_ss);
  p.lda = (long long)k * 4;
  p.w = reinterpret_cast<const char*>(w_packed);
  p.ldw = (long long)k * 4;
  p.M = m;
  p.N = n;
  p.w_rows = (int)roundup(n, 128);
  p.kslabs = k / 32;
  p.bias = bias; p.gamma = gamma; p.res = res; p.ld_res = ld_res;
  p.y = y; p.ldy = ldy;
  p.y_ss = reinterpret_cast<char*>(y_ss);
  p.ldy_ss = (long long)n * 4;
  p.act = act;
  p.scale_cols = scale_cols;
  p.scale = scale;
  hipStream_t s = (hipStream_t)stream;
  const char* const fe = getenv("PRV2_GEMM_SS_TILE");  // A/B switch (128 / 256), read per call so that one process can time both
  const int force = fe ? atoi(fe) : 0;
  const long long t256 = cdiv(m, 256) * cdiv(n, 256);
  // 256 x 256 tiles (one workgroup per CU) when they fill whole rounds of the 256 CUs; otherwise 128 x 128 (two per CU).
  // Measured (tools/probes/gemm_ss_bench.py): 14350 x 1024 -> 1024..4096: 375-425 vs 335-385 TF; 4100 rows: 128-tiles win but on 3072 columns
  const bool big = force ? force == 256 : (t256 >= 180 && (double)t256 / (double)(cdiv(t256, 256) * 256) >= 0.75);
  PRV2_REQUIRE(act == PRV2_ACT_NONE || act == PRV2_ACT_GELU, "gemm_ss: activation %d (built: none, GELU -- what the ViT blocks use)", act);
#define PRV2_GSS(WM_, WN_, RI_, RJ_, NT_, NS_)                                                                                         \
  do {                                                                                                                                 \
    if (y_ss) {                                                                                                                        \
      if (act == PRV2_ACT_GELU) hipLaunchKernelGGL((gemm_ss_kernel<WM_, WN_, RI_, RJ_, true, PRV2_ACT_GELU, NS_>), grid, dim3(NT_), 0, s, p);  \
      else hipLaunchKernelGGL((gemm_ss_kernel<WM_, WN_, RI_, RJ_, true, PRV2_ACT_NONE, NS_>), grid, dim3(NT_), 0, s, p);               \
    } else {                                                                                                                           \
      if (act == PRV2_ACT_GELU) hipLaunchKernelGGL((gemm_ss_kernel<WM_, WN_, RI_, RJ_, false, PRV2_ACT_GELU, NS_>), grid, dim3(NT_), 0, s, p); \
      else hipLaunchKernelGGL((gemm_ss_kernel<WM_, WN_, RI_, RJ_, false, PRV2_ACT_NONE, NS_>), grid, dim3(NT_), 0, s, p);              \
    }                                                                                                                                  \
  } while (0)
  const char* const be = getenv("PRV2_GEMM_SS_BLOCKED");  // A/B switch
  if (force == 2563 || force == 1283) {  // experiments (profiles/r03_experiments.txt): 256 x 128 / 128 x 256 tiles, three 48 KB stages
    const int tm = force == 1283 ? 128 : 256, tn = force == 1283 ? 256 : 128;
    p.tiles_n = (int)cdiv(n, tn);
    p.tiles_m = (int)cdiv(m, tm);
    p.blocked = (be ? atoi(be) != 0 : true) && p.tiles_n % 8 == 0 && p.tiles_m >= 8;
    const dim3 grid((unsigned)((p.blocked ? roundup(p.tiles_m, 4) : p.tiles_m) * p.tiles_n));
    if (force == 2563) PRV2_GSS(2, 4, 8, 2, 512, 3);
    else PRV2_GSS(2, 4, 4, 4, 512, 3);
    set_kernel("gemm_ss_kernel", 256, PRV2_PREC_BF16X3);
  } else
  if (big) {
    p.tiles_n = (int)cdiv(n, 256);
    p.tiles_m = (int)cdiv(m, 256);
    p.blocked = (be ? atoi(be) != 0 : true) && p.tiles_n % 8 == 0 && p.tiles_m >= 8;
    const dim3 grid((unsigned)((p.blocked ? roundup(p.tiles_m, 4) : p.tiles_m) * p.tiles_n));
    const char* const pe = getenv("PRV2_GSS_PERSIST");  // A/B switch (0: one tile per workgroup)
    const char* const dfe = getenv("PRV2_GSS_DEFER");  // A/B switch; OFF by default: same time within noise (profiles/r04_experiments.txt)
    if ((pe ? atoi(pe) != 0 : true) && (y_ss || act == PRV2_ACT_NONE)) {  // (fp32 rows + GELU: no layer has it; the one-tile kernel keeps the form)
      // persistent workgroups, one per CU, walking the same tile order (gemm_ss_p_kernel); ids dealt to XCDs as the grid's would be
      p.vgrid = (int)grid.x;
      const char* const ppe = getenv("PRV2_GSS_PPB");  // A/B switch: DMA pieces behind each row block (1, 2, 4; 8 = all in front)
      const int ppb = ppe ? atoi(ppe) : 2;
      const dim3 pgrid((unsigned)(p.vgrid < 256 ? p.vgrid : 256));
      if (y_ss) {
        if (act == PRV2_ACT_GELU) launch_gss_p<true, PRV2_ACT_GELU>(ppb, pgrid, s, p);
        else launch_gss_p<true, PRV2_ACT_NONE>(ppb, pgrid, s, p);
      } else {
        launch_gss_p<false, PRV2_ACT_NONE>(ppb, pgrid, s, p);
      }
    } else if (dfe ? atoi(dfe) != 0 : false) {
      if (y_ss) {
        if (act == PRV2_ACT_GELU) hipLaunchKernelGGL((gemm_ss_kernel<2, 4, 8, 4, true, PRV2_ACT_GELU, 2, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemm_ss_kernel<2, 4, 8, 4, true, PRV2_ACT_NONE, 2, true>), grid, dim3(512), 0, s, p);
      } else {
        if (act == PRV2_ACT_GELU) hipLaunchKernelGGL((gemm_ss_kernel<2, 4, 8, 4, false, PRV2_ACT_GELU, 2, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemm_ss_kernel<2, 4, 8, 4, false, PRV2_ACT_NONE, 2, true>), grid, dim3(512), 0, s, p);
      }
    } else
    PRV2_GSS(2, 4, 8, 4, 512, 2);
    set_kernel("gemm_ss_kernel", 256, PRV2_PREC_BF16X3);
  } else {
    p.tiles_n = (int)cdiv(n, 128);
    p.tiles_m = (int)cdiv(m, 128);
    p.blocked = (be ? atoi(be) != 0 : true) && p.tiles_n % 8 == 0 && p.tiles_m >= 64;  // (hurts small grids: 1037 x 3072: 202 -> 148 TF)
    const dim3 grid((unsigned)((p.blocked ? roundup(p.tiles_m, 4) : p.tiles_m) * p.tiles_n));
    // Grids that leave most of the chip idle (one image: 769 / 1037 token rows -> 56-224 tiles of 128 x 128) take 64 x 64 tiles: four
    // times the workgroups; a workgroup's time is its serial K loop (a slab of a 128-tile is 768 MFMA cycles in ~1.25 k: compute,
    // not latency -- four LDS stages instead of two made it 13 % SLOWER, tools/probes/gemm_ss_small_bench.py), so only more of
    // them in parallel helps: 4096 -> 1024 at 769 rows 87 -> 61 us.  Same arithmetic per output element: bit-identical.
    const char* const se = getenv("PRV2_GEMM_SS_SMALL");  // A/B switch
    const bool small = (se ? atoi(se) != 0 : true) && !force && (long long)p.tiles_m * p.tiles_n < 256;
    if (small) {
      p.tiles_n = (int)cdiv(n, 64);
      p.tiles_m = (int)cdiv(m, 64);
      p.blocked = 0;
      const dim3 grid64((unsigned)(p.tiles_m * p.tiles_n));
      // <= 256 of them (one per CU): four LDS stages -- three slabs in flight hide the L2 round trip that a 192-MFMA-cycle slab
      // cannot (4096 -> 1024: 61 -> 41 us); more than 256: two stages, so that several workgroups share a CU (four stages: -20 %)
      const char* const de = getenv("PRV2_GEMM_SS_DEEP");  // A/B switch: 2 / 4 forces the stage count
      const dim3 grid = grid64;
      const bool deep = de ? atoi(de) == 4 : (long long)p.tiles_m * p.tiles_n <= 256;
      if (deep) PRV2_GSS(2, 2, 2, 2, 256, 4);
      else PRV2_GSS(2, 2, 2, 2, 256, 2);
      set_kernel("gemm_ss_kernel", 64, PRV2_PREC_BF16X3);
    } else {
      PRV2_GSS(2, 2, 4, 4, 256, 2);
      set_kernel("gemm_ss_kernel", 128, PRV2_PREC_BF16X3);
    }
  }
#undef PRV2_GSS
  PRV2_LAUNCH_CHECK("gemm_ss");
  return 0;
}
