// A 5x5 convolution of a bilinear(align_corners=True) x2 UPSAMPLE, computed at the LOW resolution -- the composite of C2FModule's
// output_conv1 (3x3, 256 -> 128, with refinenet1.out_conv folded in) and output_conv2[0] (3x3, 128 -> 32), which the reference applies
// back to back with nothing non-linear in between (bi_directional_fusion_model.py:139-146 interpolate + out_conv, :169-173 definitions,
// :201-203 use):
//     conv3x3(conv3x3(up(u); W1) + b1; W2) + b2  ==  conv5x5(up(u); Weff) + bias_map - ring_fix,     Weff[d] = sum_{d1 + d2 = d} W2[d2] W1[d1]
// (bias_map: data independent, 5 x 5 position classes; ring_fix: non-zero on the one-pixel border ring only -- the outer conv's zero padding
//  hides inner-conv outputs that the 5x5 form over the zero-padded map includes; tools/studies/composite5x5_ring.py has the float64 algebra.)
// 25 * 256 * 32 = 205 k MACs per output pixel instead of 9 * 256 * 128 + 9 * 128 * 32 = 332 k, and in csrc/upconv.hip's form -- the conv is
// linear and its input an interpolation, so the channel contraction commutes with the interpolation:
//     conv5x5(up(u); Weff)(p) = sum_tap [p + d_tap inside the image] Bil(G_tap; s(p + d_tap)),     G_tap = Weff[:, :, tap] . u   (1x1 GEMMs)
// -- 25 taps x 32 channels = 800 GEMM columns per source pixel instead of 1152 and 800 gathered tap values per output pixel instead of 1152,
// while the 128-channel full-resolution intermediate (4 GB written + read per 41-tile batch) and the 128 -> 32 layer disappear.
//
// This file: the MAIN term (+ bias map, + activation off the ring) as one kernel, and the ring fix as a second, tiny one.
// upconv5x5_kernel is csrc/upconv.hip's kernel re-cut for five-tap kernel rows:
//   workgroup = 14 x 24 output pixels (8 waves); source footprint <= 11 x 17 low-resolution pixels ((14 + 4) / 2 + 2 rows, (24 + 4) / 2 + 2
//   columns), linearised and padded to 192 = 12 MFMA row runs;
//   a PASS is one KERNEL ROW ky: GEMM [192 x cin] x [cin x 160] (five taps x 32 channels = ten 16-column blocks; 8 waves = 4 row slots of
//   3 runs x 2 column slots of 5 blocks: 60 accumulator registers), main loop exactly as upconv.hip's (footprint slab fp32 -> bf16 hi / lo ->
//   LDS once per slab, the slab's 160 x 128 B of packed weights by LDS-DMA, two stages, one barrier per slab, weights as the MFMA's A operand);
//   the gather of a kernel row goes through the G tile in TWO rounds -- taps kx 0..2, then 3..4 -- so that the tile keeps upconv.hip's size
//   (192 x 96 floats) and the next pass's first slab still lands beside it; a round walks its taps' shared source columns together
//   (wave-uniform steps, table entries in VGPR lanes); the OUTPUT accumulators (6 pixels x 4 channels per thread) live across all five
//   passes and leave once, with the position's bias class added and the activation applied -- except on the ring, whose pixels are
//   written raw: upconv5x5_ring_kernel subtracts their fix and applies the activation.
// upconv5x5_ring_kernel: per edge the fix is a 1-D five-tap conv (zero padded) of the border row / column of up(u) -- an interpolation of
//   u's border row / column, so once more tap GEMMs at the source resolution (a plain prv2_conv2d over the four border lines of u, done by
//   the caller) and a 1-D two-corner gather here; at the four corner pixels the outside corner position is counted by both edges and added
//   back once (its own 32 GEMM columns).
// Weights: the ordinary packed image of prv2_pack_conv_weight(cout, cin, 5, 5) of the composite Weff (host side, float64: fusion.py).
// Arithmetic: the split products and fp32 accumulation of the other bf16x3 kernels; fp32-grade, not bit-identical to the two-conv sequence.
#include <cstdlib>
#include <type_traits>

#include "igemm.h"

namespace prv2 {

namespace upc5 {
constexpr int TH = 14, TW = 24, SEG = TW / 4;  // output tile; pixels per thread in the gather
constexpr int LR = 11, LC = 17;               // source footprint of a tile (rows x columns)
constexpr int RUNS = 12, MPX = RUNS * 16;     // 192 >= LR * LC = 187
constexpr int CP = 32;                        // output channels (all of them: one channel group)
constexpr int NKY = 5, NKX = 5;
constexpr int AROW = 160;                     // bytes per footprint pixel in LDS: [32 bf16 hi | 32 bf16 lo | 32 B pad]
constexpr int A_BYTES = MPX * AROW;           // 30 720
constexpr int W_BYTES = NKX * CP * 128;       // 20 480
constexpr int STAGE = A_BYTES + W_BYTES;      // 51 200
constexpr int GT = 3;                         // taps per gather round (rounds: kx 0..2, kx 3..4)
constexpr int CLD = GT * CP + 4;              // G tile pixel pitch (floats)
constexpr int GRP = LC * CLD - 4;             // G tile row pitch: 32 banks mod 64 (see upconv.hip)
static_assert(GRP % 64 == 32, "G tile row pitch");
constexpr int C_BYTES = MPX * CLD * 4;        // 76 800
constexpr int S1_OFF = C_BYTES;               // [ G tile | stage 1 | column table ]; stage 0 aliases the G tile
constexpr int TAB_OFF = S1_OFF + STAGE;
constexpr int NTAB = TW + 4;                  // output columns x0 - 2 .. x0 + TW + 1
constexpr int SMEM_BYTES = TAB_OFF + NTAB * 16;
constexpr int NPIECE = NKX * 4, NDMA = 3;     // weight pieces (tap, 8-row group) per slab; per wave 3 or 2
static_assert(LR * LC <= MPX && STAGE <= C_BYTES && SMEM_BYTES <= 160 * 1024, "LDS layout");
}  // namespace upc5

struct Upconv5Params {
  const float* xu;  // low-resolution source, NHWC
  int uH, uW, ldxu, C;
  long long xu_bstride;
  const void* w;          // packed [cout rows][25 taps][C] image of the composite weights
  const float* bias_map;  // [5][5][Cout]: row class (0, 1, interior, H - 2, H - 1) x column class
  int act;
  float* y;
  int H, W, ldy, Cout;
  long long y_bstride;
  float usy, usx;
  int tiles_x, tiles_y;
};

__device__ __forceinline__ f32x4 u5_fma4(float s, const f32x4 a, const f32x4 c) {
  f32x4 r;
  r.x = __builtin_fmaf(s, a.x, c.x);
  r.y = __builtin_fmaf(s, a.y, c.y);
  r.z = __builtin_fmaf(s, a.z, c.z);
  r.w = __builtin_fmaf(s, a.w, c.w);
  return r;
}

__device__ __forceinline__ int u5_class(int v, int n) { return v < 2 ? v : (v >= n - 2 ? v - (n - 5) : 2); }  // 0, 1, 2 (interior), 3, 4

template <int PREC>
__global__ void __launch_bounds__(512) upconv5x5_kernel(const Upconv5Params p) {
  using namespace upc5;
  __shared__ __attribute__((aligned(1024))) char smem[SMEM_BYTES];
  float* const csm = reinterpret_cast<float*>(smem);
  f32x4* const tab = reinterpret_cast<f32x4*>(smem + TAB_OFF);

  // ---- XCD-aware block -> tile ----
  int t = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = t & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
  }
  const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, n_img = t / (p.tiles_x * p.tiles_y);
  const int y0 = ty * TH, x0 = tx * TW;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15, g = lane >> 4;
  // MFMA roles: wave = (row slot: three of the twelve footprint runs) x (column slot: five of the pass's ten 16-column blocks)
  const int nslot = wave >> 2, mslot = wave & 3;
  const int run0 = 3 * mslot;
  const int seg = wave & 3;  // gather roles: wave = (6-pixel segment, half of the tile's rows); lane = (row, channel quad)

  // ---- source footprint origin and the column table (output columns x0 - 2 .. x0 + TW + 1) ----
  const int rbase = ac_tap(max(y0 - 2, 0), p.usy, p.uH).i0;
  const int cbase = ac_tap(max(x0 - 2, 0), p.usx, p.uW).i0;
  if (tid < NTAB) {
    const int xx = x0 - 2 + tid;
    const bool valid = (unsigned)xx < (unsigned)p.W;
    const AxisTap a = ac_tap(min(max(xx, 0), p.W - 1), p.usx, p.uW);
    float w0 = a.w0, w1 = a.w1;
    if (a.i1 == a.i0) {  // clamped at the last source column: both corners are that column
      w0 += w1;
      w1 = 0.f;
    }
    f32x4 e;
    e.x = __builtin_bit_cast(float, a.i0 - cbase);
    e.y = valid ? w0 : 0.f;
    e.z = valid ? w1 : 0.f;
    e.w = 0.f;
    tab[tid] = e;
  }
  __syncthreads();
  // lane l (< SEG + 4) of every wave keeps table entry SEG seg + l: the walk fetches what a step needs with v_readlane
  int vci, vw0, vw1;
  {
    const volatile int* const te = reinterpret_cast<const volatile int*>(tab + SEG * seg + min(lane, SEG + 3));
    vci = te[0];
    vw0 = te[1];
    vw1 = te[2];
  }

  // ---- footprint loader: item = (footprint pixel prow + 64 it, 4 channels `chunk`), fp32 -> bf16 hi / lo on the way to LDS ----
  const int chunk = tid & 7, prow_lin = tid >> 3;
  const int prow = (prow_lin & ~3) | ((prow_lin & 1) << 1) | ((prow_lin >> 1) & 1);
  constexpr int NIT = 3;
  unsigned hoff[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int px = prow + 64 * it;
    const int r = px / LC, c = px - r * LC;
    const int gr = min(rbase + r, p.uH - 1), gc = min(cbase + c, p.uW - 1);
    hoff[it] = (unsigned)(((gr * p.uW + gc) * p.ldxu + chunk * 4) * 4);
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.xu + (long long)n_img * p.xu_bstride), 0, (int)((((long long)p.uH * p.uW - 1) * p.ldxu + p.C) * 4), 0x00020000);
  f32x4 ra[NIT];
  auto load_a_async = [&](int cc, int it) {
    ra[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, hoff[it], cc * BK * 4, 0));
  };
  auto store_a = [&](int stage, int it) {
    const int px = prow + 64 * it;
    const f32x4 v = ra[it];
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const bf16x4 hi = __builtin_convertvector(v, bf16x4);
    const u32x2 hw = __builtin_bit_cast(u32x2, hi);
    f32x4 hf;
    hf.x = __builtin_bit_cast(float, hw.x << 16);
    hf.y = __builtin_bit_cast(float, hw.x & 0xffff0000u);
    hf.z = __builtin_bit_cast(float, hw.y << 16);
    hf.w = __builtin_bit_cast(float, hw.y & 0xffff0000u);
    const bf16x4 lo = __builtin_convertvector(v - hf, bf16x4);
    const unsigned addr = (unsigned)(size_t)(smem + stage * S1_OFF + px * AROW) + chunk * 8;
    const unsigned long long h = __builtin_bit_cast(unsigned long long, hi), l = __builtin_bit_cast(unsigned long long, lo);
    if constexpr (PREC == PRV2_PREC_BF16X3) asm volatile("ds_write2_b64 %0, %1, %2 offset1:8" ::"v"(addr), "v"(h), "v"(l) : "memory");
    else asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(h) : "memory");
  };

  // ---- weight DMA: piece q = (tap kx = q >> 2, 8-row group q & 3) of kernel row ky; wave w moves pieces w, w + 8, w + 16 ----
  const int dr = lane >> 3, dsl = lane & 7;
  const long long w_row_bytes = (long long)NKY * NKX * p.C * 4;
  const unsigned wlane = (unsigned)(dr * (int)w_row_bytes + dsl * 16);
  auto dma_w = [&](int ky, int cc, int stage) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int q = wave + 8 * i;
      if (q < NPIECE) {  // wave-uniform
        const int tap = q >> 2, rg = q & 3;
        const char* src = reinterpret_cast<const char*>(p.w) + (long long)(rg * 8) * w_row_bytes + ((long long)(ky * NKX + tap) * p.C + cc * BK) * 4;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(smem + stage * S1_OFF + A_BYTES + q * 1024));
        // (s_nop 4: see upconv.hip -- a VMEM instruction reading an SGPR a VALU instruction wrote needs 5 wait states, and the hazard
        //  recognizer does not look into inline asm)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(wlane), "s"(src) : "memory");
      }
    }
  };

  // ---- fragments ----
  const int key = (m16 >> 1) & 7;
  const int a_off = (run0 * 16 + m16) * AROW + g * 16;
  const int b_off_hi = A_BYTES + (nslot * 5 * 16 + m16) * 128 + ((g ^ key) << 4);
  const int b_off_lo = A_BYTES + (nslot * 5 * 16 + m16) * 128 + (((4 + g) ^ key) << 4);
  auto mma = [&](f32x4& c, const bf16x8& xh, const bf16x8& xl, const bf16x8& wh, const bf16x8& wl) {
    if constexpr (PREC == PRV2_PREC_BF16X3) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, c, 0, 0, 0);
    }
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, c, 0, 0, 0);
  };

  const int cslabs = p.C / BK;
  const long long img_y = (long long)n_img * p.y_bstride;

  // ---- output accumulators: live across the five kernel-row passes ----
  f32x4 o[SEG];
#pragma unroll
  for (int xi = 0; xi < SEG; ++xi) o[xi] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- first pass: slab 0 -> stage 1 ----
  dma_w(0, 0, 1);
#pragma unroll
  for (int it = 0; it < NIT; ++it) load_a_async(0, it);
#pragma unroll
  for (int it = 0; it < NIT; ++it) store_a(1, it);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the weight DMA: asm issued it, asm waits for it)

#pragma unroll 1
  for (int ky = 0; ky < NKY; ++ky) {
    f32x4 acc[3][5];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifdef UPC5_ABL_NOMAIN  // (timing ablations: results wrong)
    for (int cc = 0; cc < 0; ++cc) {
#else
    for (int cc = 0; cc < cslabs; ++cc) {
#endif
      const int st = (cc + 1) & 1;  // slab 0 of a pass sits in stage 1
      if (cc == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const bool more = cc + 1 < cslabs;  // block-uniform
      const char* const sb = smem + st * S1_OFF;
      bf16x8 ah[3], al[3], bh[2], bl[2];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        ah[a] = *reinterpret_cast<const bf16x8*>(sb + a_off + a * 16 * AROW);
        if constexpr (PREC == PRV2_PREC_BF16X3) al[a] = *reinterpret_cast<const bf16x8*>(sb + a_off + a * 16 * AROW + 64);
      }
      auto read_b = [&](int slot, int j) {
        bh[slot] = *reinterpret_cast<const bf16x8*>(sb + b_off_hi + j * 2048);
        if constexpr (PREC == PRV2_PREC_BF16X3) bl[slot] = *reinterpret_cast<const bf16x8*>(sb + b_off_lo + j * 2048);
      };
      read_b(0, 0);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        if (j + 1 < 5) read_b((j + 1) & 1, j + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 3; ++a) mma(acc[a][j], ah[a], al[a], bh[j & 1], bl[j & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (j == 0 && more) {  // the next slab's requests go out behind the first MFMA block
          dma_w(ky, cc + 1, st ^ 1);
#pragma unroll
          for (int it = 0; it < NIT; ++it) load_a_async(cc + 1, it);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (j == 3 && more) {
#pragma unroll
          for (int it = 0; it < NIT; ++it) store_a(st ^ 1, it);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // stage 0 becomes the G tile; stage 1 is free
    const bool has_next = ky + 1 < NKY;   // block-uniform
    if (has_next) dma_w(ky + 1, 0, 1);    // the next pass's weights fly during the whole gather

    // ---- gather of kernel row ky, two rounds through the G tile: taps kx 0..2, then kx 3..4 ----
    int lane_g = lane;
    asm volatile("" : "+v"(lane_g));
    const int seg4 = (lane_g >> 2) & 7;
    const int quad = ((seg4 & 2) << 1) + (lane_g & 3), prow_t = ((0xD728 >> (2 * seg4)) & 3) + 4 * (lane_g >> 5) + 8 * (wave >> 2);
    const int m16g = lane_g & 15, gg = lane_g >> 4;
    int gpx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int px = (run0 + a) * 16 + m16g;
      gpx[a] = px * CLD - 4 * (px / LC);
    }
    const int yy = y0 + prow_t + ky - 2;
    const bool vy = (unsigned)yy < (unsigned)p.H;
    const AxisTap ay = ac_tap(min(max(yy, 0), p.H - 1), p.usy, p.uH);
    const float wy0 = vy ? ay.w0 : 0.f, wy1 = vy ? ay.w1 : 0.f;
    const unsigned g0 = (unsigned)(size_t)(csm + (ay.i0 - rbase) * GRP + 4 * quad);
    const unsigned g1 = (unsigned)(size_t)(csm + (ay.i1 - rbase) * GRP + 4 * quad);
    typedef const __attribute__((address_space(3))) f32x4* lds_f32x4;

    auto round = [&](auto kx0_c, auto nt_c, bool first, bool last) {
      constexpr int KX0 = decltype(kx0_c)::value, NT = decltype(nt_c)::value;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int b = nslot * 5 + j, tp = b >> 1;
        if (tp >= KX0 && tp < KX0 + NT) {  // wave-uniform
          const int col = (tp - KX0) * CP + (b & 1) * 16 + 4 * gg;
#pragma unroll
          for (int a = 0; a < 3; ++a) *reinterpret_cast<f32x4*>(&csm[gpx[a] + col]) = acc[a][j];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (first && has_next) {  // the next pass's first footprint slab: requested in the FIRST round (both walks to arrive under), stored in the last
#pragma unroll
        for (int it = 0; it < NIT; ++it) load_a_async(0, it);
      }
      auto issue = [&](int c, f32x4 (&v0)[NT], f32x4 (&v1)[NT]) {  // source column c of the round's taps (tap = a constant offset)
        const unsigned off = (unsigned)(c * CLD * 4);
        unsigned a0 = g0 + off, a1 = g1 + off;
        asm volatile("" : "+v"(a0), "+v"(a1));  // (pins each read to the branch that needs it: see upconv.hip)
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          v0[k] = *(lds_f32x4)(size_t)(a0 + k * CP * 4);
          v1[k] = *(lds_f32x4)(size_t)(a1 + k * CP * 4);
        }
      };
#ifndef UPC5_ABL_NOWALK
      f32x4 lc[NT], ln[NT], q0[NT], q1[NT];
      int c = __builtin_amdgcn_readlane(vci, KX0);
      {
        f32x4 a0[NT], a1[NT];
        issue(c, a0, a1);
        issue(c + 1, q0, q1);
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          lc[k] = u5_fma4(wy0, a0[k], wy1 * a1[k]);
          ln[k] = u5_fma4(wy0, q0[k], wy1 * q1[k]);
        }
        issue(c + 2, q0, q1);  // one column ahead of the walk
      }
#pragma unroll
      for (int i = 0; i < SEG + NT - 1; ++i) {
        const int ci = __builtin_amdgcn_readlane(vci, KX0 + i);
        if (ci != c) {  // wave-uniform: the walk enters the next source column (scale <= 1/2: one step at most)
          c = ci;
#pragma unroll
          for (int k = 0; k < NT; ++k) {
            lc[k] = ln[k];
            ln[k] = u5_fma4(wy0, q0[k], wy1 * q1[k]);
          }
          issue(c + 2, q0, q1);
        }
        const float w0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vw0, KX0 + i)), w1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vw1, KX0 + i));
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          const int xi = i - k;
          if (xi >= 0 && xi < SEG) {
            o[xi] = u5_fma4(w0, lc[k], u5_fma4(w1, ln[k], o[xi]));
            asm volatile("" : "+v"(o[xi]));  // (pins the update here)
          }
        }
      }
#endif
      if (last && has_next) {  // the next pass's first footprint slab -> stage 1 (its weights: issued before the gather)
#pragma unroll
        for (int it = 0; it < NIT; ++it) store_a(1, it);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the G tile is rewritten by the next round / the next pass's slab 1
    };
    round(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, true, false);
    round(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{}, false, true);
  }

  // ---- output: 6 pixels x 4 channels per thread; + the position's bias class; activation everywhere but on the ring ----
  {
    int lane_g = lane;
    asm volatile("" : "+v"(lane_g));
    const int seg4 = (lane_g >> 2) & 7;
    const int quad = ((seg4 & 2) << 1) + (lane_g & 3), prow_t = ((0xD728 >> (2 * seg4)) & 3) + 4 * (lane_g >> 5) + 8 * (wave >> 2);
    const int ch0 = 4 * quad;
    const int nvalid = min(max(p.Cout - ch0, 0), 4);
    const int oy = y0 + prow_t;
    const bool rowok = prow_t < TH && oy < p.H;
    const int cy = u5_class(min(oy, p.H - 1), p.H);
    const bool ring_y = oy == 0 || oy == p.H - 1;
    dispatch_act(p.act, [&](auto act_c) {
#pragma unroll
      for (int xi = 0; xi < SEG; ++xi) {
        const int ox = x0 + SEG * seg + xi;
        const int cx = u5_class(min(ox, p.W - 1), p.W);
        const float* bm = p.bias_map + (cy * 5 + cx) * p.Cout + ch0;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < nvalid) bv[e] = bm[e];
        const bool ring = ring_y || ox == 0 || ox == p.W - 1;
        // (the arithmetic stays OUTSIDE the exec-masked branch: see upconv.hip)
        f32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float raw = o[xi][e] + bv[e];
          ov[e] = ring ? raw : act_apply_bf(raw, decltype(act_c)::value);
        }
        if (rowok && ox < p.W && nvalid > 0) {
          float* dst = p.y + img_y + ((long long)oy * p.W + ox) * p.ldy + ch0;
          if (nvalid == 4) {
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(ov) : "memory");
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (e < nvalid) dst[e] = ov[e];
          }
        }
      }
    });
  }
}

// The ring fix.  ge: [n][2 uw + 2 uh][ldg] = the border lines of u (top row, bottom row, left column, right column: positions 0 .. uw - 1,
// uw .. 2 uw - 1, 2 uw .. 2 uw + uh - 1, ...) through ALL edges' tap weights: columns (edge e * 7 + j) * cout + c, j < 5 the taps -2 .. 2 along the
// edge, j = 5 / 6 the corner term at the line's first / last position.  Thread = (ring pixel, 4 channels): pixel r of an image's ring =
// top row x = r (r < W), bottom row (r < 2 W), left column y = 1 .. H - 2, right column; the corner pixels belong to the rows and take the
// columns' contribution too.
__global__ void __launch_bounds__(256) upconv5x5_ring_kernel(float* __restrict__ y, int ldy, long long y_bstride, int N, int H, int W, int cout,
                                                              const float* __restrict__ ge, int ldg, int uH, int uW, float usy, float usx, int act) {
  const int cq = cout / 4;
  const int ring = 2 * W + 2 * (H - 2);
  const long long tidx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tidx >= (long long)N * ring * cq) return;
  const int c4 = (int)(tidx % cq) * 4;
  const int r = (int)((tidx / cq) % ring), n = (int)(tidx / ((long long)cq * ring));
  int oy, ox;
  if (r < W) { oy = 0; ox = r; }
  else if (r < 2 * W) { oy = H - 1; ox = r - W; }
  else if (r < 2 * W + H - 2) { oy = r - 2 * W + 1; ox = 0; }
  else { oy = r - 2 * W - (H - 2) + 1; ox = W - 1; }
  const float* gn = ge + (long long)n * (2 * uW + 2 * uH) * ldg;
  float4 fix = make_float4(0.f, 0.f, 0.f, 0.f);
  auto acc = [&](float w, const float* src) {
    const float4 v = *reinterpret_cast<const float4*>(src);
    fix.x = fmaf(w, v.x, fix.x);
    fix.y = fmaf(w, v.y, fix.y);
    fix.z = fmaf(w, v.z, fix.z);
    fix.w = fmaf(w, v.w, fix.w);
  };
  // an edge's contribution at position ``pos`` of ``len`` along it: sum_e [0 <= pos + e < len] Lerp(G_e; s(pos + e))
  auto edge = [&](int e, int line0, int pos, int len, float us, int ulen) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int q = pos + j - 2;
      if (q >= 0 && q < len) {
        const AxisTap a = ac_tap(q, us, ulen);
        const float* col = gn + (e * 7 + j) * cout + c4;
        acc(a.w0, col + (long long)(line0 + a.i0) * ldg);
        acc(a.w1, col + (long long)(line0 + a.i1) * ldg);
      }
    }
  };
  const bool top = oy == 0, bot = oy == H - 1, lef = ox == 0, rig = ox == W - 1;
  if (top) edge(0, 0, ox, W, usx, uW);
  if (bot) edge(1, uW, ox, W, usx, uW);
  if (lef) edge(2, 2 * uW, oy, H, usy, uH);
  if (rig) edge(3, 2 * uW + uH, oy, H, usy, uH);
  // corners: the outside corner position is in both edges' sums -- its term (columns 5 / 6 of the ROW edge's group, sampled like the row edge
  // itself at the pixel's own column) goes back in once
  auto corner = [&](int e, int j, int line0, int q) {
    const AxisTap a = ac_tap(q, usx, uW);
    const float* col = gn + (e * 7 + j) * cout + c4;
    acc(-a.w0, col + (long long)(line0 + a.i0) * ldg);
    acc(-a.w1, col + (long long)(line0 + a.i1) * ldg);
  };
  if (top && lef) corner(0, 5, 0, 0);
  if (top && rig) corner(0, 6, 0, W - 1);
  if (bot && lef) corner(1, 5, uW, 0);
  if (bot && rig) corner(1, 6, uW, W - 1);
  float* dst = y + (long long)n * y_bstride + ((long long)oy * W + ox) * ldy + c4;
  float4 v = *reinterpret_cast<float4*>(dst);
  v.x = act_apply(v.x - fix.x, act);
  v.y = act_apply(v.y - fix.y, act);
  v.z = act_apply(v.z - fix.z, act);
  v.w = act_apply(v.w - fix.w, act);
  *reinterpret_cast<float4*>(dst) = v;
}

// The four border lines of up(u) along which the ring fix runs, at u's resolution: lines[n][pos][c], pos = top row (uW), bottom row (uW), left
// column (uH), right column (uH).  The border row / column of up(u) is the align_corners sample of u at output row 0 / H - 1 (column 0 / W - 1):
// row 0 is u's row 0 exactly, row H - 1 the two-row blend PyTorch's float32 source index gives (scale * (H - 1) may fall an ulp short of uH - 1).
__global__ void __launch_bounds__(256) upconv5x5_lines_kernel(const float* __restrict__ xu, long long xu_bstride, int uH, int uW, int ldxu, int C, int N, int H,
                                                               int W, float usy, float usx, float* __restrict__ lines) {
  const int cq = C / 4, npos = 2 * uW + 2 * uH;
  const long long tidx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tidx >= (long long)N * npos * cq) return;
  const int c4 = (int)(tidx % cq) * 4, pos = (int)((tidx / cq) % npos), n = (int)(tidx / ((long long)cq * npos));
  const float* img = xu + (long long)n * xu_bstride;
  const float4* p0;
  const float4* p1;
  float w0, w1;
  if (pos < 2 * uW) {  // rows: output row 0 / H - 1
    const AxisTap a = ac_tap(pos < uW ? 0 : H - 1, usy, uH);
    const int col = pos < uW ? pos : pos - uW;
    p0 = reinterpret_cast<const float4*>(img + ((long long)a.i0 * uW + col) * ldxu + c4);
    p1 = reinterpret_cast<const float4*>(img + ((long long)a.i1 * uW + col) * ldxu + c4);
    w0 = a.w0; w1 = a.w1;
  } else {             // columns: output column 0 / W - 1
    const int q = pos - 2 * uW;
    const AxisTap a = ac_tap(q < uH ? 0 : W - 1, usx, uW);
    const int row = q < uH ? q : q - uH;
    p0 = reinterpret_cast<const float4*>(img + ((long long)row * uW + a.i0) * ldxu + c4);
    p1 = reinterpret_cast<const float4*>(img + ((long long)row * uW + a.i1) * ldxu + c4);
    w0 = a.w0; w1 = a.w1;
  }
  const float4 a = *p0, b = *p1;
  float4 r;
  r.x = w0 * a.x + w1 * b.x;
  r.y = w0 * a.y + w1 * b.y;
  r.z = w0 * a.z + w1 * b.z;
  r.w = w0 * a.w + w1 * b.w;
  *reinterpret_cast<float4*>(lines + ((long long)n * npos + pos) * C + c4) = r;
}

}  // namespace prv2

using namespace prv2;

static inline bool u5_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static bool upconv5_shape_ok(const prv2_ups_src* u, int32_t n, int32_t h, int32_t w, int32_t cout, int32_t prec) {
  return u && u->x && n > 0 && h >= 5 && w >= 5 && cout > 0 && cout <= upc5::CP && cout % 4 == 0 && (prec == PRV2_PREC_BF16X3 || prec == PRV2_PREC_BF16) &&
         u->channels >= 32 && u->channels % 32 == 0 && u->ld % 4 == 0 && u->ld >= u->channels && u->h >= 2 && u->w >= 2 && ac_scale(u->h, h) <= 0.5f &&
         ac_scale(u->w, w) <= 0.5f && (long long)u->h * u->w * u->ld < (1LL << 29) && (long long)128 * 25 * u->channels < (1LL << 29);
}

extern "C" int prv2_upconv5x5_supported(const prv2_ups_src* u, int32_t n, int32_t h, int32_t w, int32_t cout, int32_t prec) {
  return upconv5_shape_ok(u, n, h, w, cout, prec) ? 1 : 0;
}

extern "C" int prv2_upconv5x5(const prv2_ups_src* u, const void* w_packed, const float* bias_map, int32_t n, int32_t h, int32_t w, int32_t cout,
                              int32_t act, int32_t prec, float* y, int32_t ldy, int64_t y_bstride, void* stream) {
  PRV2_REQUIRE(upconv5_shape_ok(u, n, h, w, cout, prec),
               "upconv5x5: layer not covered (bf16 modes, cout <= 32 and %% 4 == 0, channels %% 32 == 0, source step (h_in - 1) / (h_out - 1) <= 1/2 both ways, h, w >= 5)");
  PRV2_REQUIRE(w_packed && bias_map && y && u5_al16(u->x) && u5_al16(w_packed) && u5_al16(y) && ldy % 4 == 0 && ldy >= cout && u->bstride % 4 == 0 && y_bstride % 4 == 0,
               "upconv5x5: 16-byte aligned NHWC rows (ldy=%d)", ldy);
  PRV2_REQUIRE((long long)h * w * ldy < (1LL << 31), "upconv5x5: image too large");
  Upconv5Params p = {};
  p.xu = u->x; p.uH = u->h; p.uW = u->w; p.ldxu = u->ld; p.C = u->channels;
  p.xu_bstride = u->bstride ? u->bstride : (long long)u->h * u->w * u->ld;
  p.w = w_packed; p.bias_map = bias_map; p.act = act;
  p.y = y; p.H = h; p.W = w; p.ldy = ldy; p.Cout = cout;
  p.y_bstride = y_bstride ? y_bstride : (long long)h * w * ldy;
  p.usy = ac_scale(u->h, h); p.usx = ac_scale(u->w, w);
  p.tiles_x = (int)cdiv(w, upc5::TW); p.tiles_y = (int)cdiv(h, upc5::TH);
  const long long tiles = (long long)n * p.tiles_x * p.tiles_y;
  PRV2_REQUIRE(tiles < (1LL << 31), "upconv5x5: grid too large");
  hipStream_t s = (hipStream_t)stream;
  if (prec == PRV2_PREC_BF16X3) hipLaunchKernelGGL((upconv5x5_kernel<PRV2_PREC_BF16X3>), dim3((unsigned)tiles), dim3(512), 0, s, p);
  else hipLaunchKernelGGL((upconv5x5_kernel<PRV2_PREC_BF16>), dim3((unsigned)tiles), dim3(512), 0, s, p);
  set_kernel("upconv5x5_kernel", 32, prec);
  PRV2_LAUNCH_CHECK("upconv5x5");
  return 0;
}

extern "C" int prv2_upconv5x5_ring(float* y, int32_t ldy, int64_t y_bstride, int32_t n, int32_t h, int32_t w, int32_t cout, const float* g_edges,
                                   int32_t ldg, int32_t uh, int32_t uw, int32_t act, void* stream) {
  PRV2_REQUIRE(y && g_edges && n > 0 && h >= 5 && w >= 5 && uh >= 2 && uw >= 2 && cout > 0 && cout % 4 == 0 && ldy % 4 == 0 && ldy >= cout && ldg % 4 == 0 &&
                   ldg >= 28 * cout && u5_al16(y) && u5_al16(g_edges) && y_bstride % 4 == 0,
               "upconv5x5_ring: cout %% 4 == 0, ldg >= 28 cout, 16-byte aligned rows (cout=%d ldy=%d ldg=%d)", cout, ldy, ldg);
  const long long work = (long long)n * (2 * w + 2 * (h - 2)) * (cout / 4);
  hipLaunchKernelGGL(upconv5x5_ring_kernel, dim3((unsigned)cdiv(work, 256)), dim3(256), 0, (hipStream_t)stream, y, ldy,
                     y_bstride ? y_bstride : (long long)h * w * ldy, n, h, w, cout, g_edges, ldg, uh, uw, ac_scale(uh, h), ac_scale(uw, w), act);
  PRV2_LAUNCH_CHECK("upconv5x5_ring");
  return 0;
}

extern "C" int prv2_upconv5x5_lines(const prv2_ups_src* u, int32_t n, int32_t h, int32_t w, float* lines, void* stream) {
  PRV2_REQUIRE(u && u->x && lines && n > 0 && h >= 5 && w >= 5 && u->h >= 2 && u->w >= 2 && u->channels % 4 == 0 && u->ld % 4 == 0 && u->ld >= u->channels &&
                   u5_al16(u->x) && u5_al16(lines), "upconv5x5_lines: channels %% 4 == 0, 16-byte aligned NHWC rows");
  const long long work = (long long)n * (2 * u->w + 2 * u->h) * (u->channels / 4);
  hipLaunchKernelGGL(upconv5x5_lines_kernel, dim3((unsigned)cdiv(work, 256)), dim3(256), 0, (hipStream_t)stream, u->x,
                     u->bstride ? u->bstride : (long long)u->h * u->w * u->ld, u->h, u->w, u->ld, u->channels, n, h, w, ac_scale(u->h, h), ac_scale(u->w, w), lines);
  PRV2_LAUNCH_CHECK("upconv5x5_lines");
  return 0;
}
