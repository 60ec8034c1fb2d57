// 3x3 convolution of a bilinear(align_corners=True) UPSAMPLE (x2; up to a source step of 3/5 pixel), computed at the LOW resolution:
//
//   C2FModule output_conv1       bi_directional_fusion_model.py:139-142,201   conv3x3(interpolate(path_1, scale 2, align_corners=True))
//   UpSample.forward_hardcode    fusion_model.py:15-24                        the x1 part of DoubleConv.0(cat[interpolate(x1), x2, pred1, pred2])
//
// The conv is linear and its input is an interpolation of the low-resolution tensor u, so the channel contraction commutes with the
// interpolation:
//     conv3x3(up(u); W)(p) = sum_tap [p + d_tap inside the image] Bil(G_tap; s(p + d_tap)),     G_tap = W[:, :, tap] . u   (a 1x1 GEMM)
// with s() = PyTorch's align_corners source position and Bil = its two-by-two interpolation.  The nine G_tap live on the LOW-resolution
// grid: the MFMA work is 9 * cin * cout MACs per low-resolution pixel of a tile's footprint -- 192 footprint rows for 448 outputs,
// 2.3x fewer matrix operations than the direct conv over the upsampled image (conv3x3_m16.hip UPS), which these power-limited layers
// turn into time.  What is added is VALU / LDS work: 36 FMAs per output element (four corners x nine taps), organised below so that a
// thread reads each G value it needs once.
//
// Workgroup = one 16 x 28 tile of output pixels (8 waves; 14 x 24 for a source step in (1/2, 3/5]).  Its source footprint is at most
// 11 x 17 low-resolution pixels (17 rows x 1/2 span <= 9 source rows + 2, 29 columns <= 15 + 2), linearised and padded to 192 = 12 MFMA
// row runs.  A PASS produces 32 output channels for all nine taps: GEMM [192 x cin] x [cin x 288] (18 column blocks = 9 taps x 2; 8 waves
// = 4 row slots of 3 runs x 2 column slots of 9 blocks = 108 accumulator registers), so that the output accumulators of a pass (7 pixels
// x 4 channels per thread) never coexist with more than one pass of MFMA accumulators.  The register file (256 per wave at 8 waves)
// shapes all of this: 3-tap passes over all channels would need 192 + 128 accumulators, and the first shape tried (16 x 32 tiles, 224
// rows, 4 / 3 runs per wave) needed ~270 registers -- accumulators spilled inside the slab loop and the kernel lost to the direct one.
//   main loop, per 32 input channels: the footprint slab is loaded as fp32, split to bf16 hi / lo and staged once (A fragments are read
//   ONCE per slab: they do not depend on the tap), the slab's 288 x 128 B of packed weights arrive by LDS-DMA (36 pieces of 8 rows);
//   two stages, one barrier per slab, 81 MFMAs per wave between barriers; weights are the MFMA's A operand, so that an accumulator holds
//   four consecutive CHANNELS of one footprint pixel: one 16-byte write into the G tile, in the layout the gather reads.
//   gather, per kernel row ky: the three taps' G tiles (192 x 96 floats) go through LDS; thread = (output row, 7-pixel segment, channel
//   quad).  The three taps of a kernel row walk the SAME source columns (tap kx reaches column-table entry i at its pixel i - kx), so
//   they walk together: one wave-uniform step per table entry (every lane of a wave has the same segment), the row interpolation of a
//   source column formed when the walk first needs it, one column prefetched: each G value is read once and each output costs 2 FMAs
//   per tap and channel on top.
//   LDS [G tile | stage 1 | column table]: stage 0 aliases the G tile; stage 1 is never touched by the gather, so the NEXT pass's first
//   slab lands there while this pass is gathered.
// Weights: the ordinary packed 3x3 image of prv2_pack_conv_weight (row = cout, tap-major 128-byte slabs), read slab by slab.
// Arithmetic: same split products (lo*hi, hi*lo, hi*hi) and fp32 accumulation as the other bf16x3 kernels; the sum over taps and
// corners is ordered differently from upsample -> conv (not bit-identical: tests compare against the float64 reference of the pair).
#include <cstdlib>
#include <type_traits>

#include "igemm.h"

namespace prv2 {

namespace upc {
constexpr int TH0 = 16, TW0 = 28;            // output tile for a source step <= 1/2 pixel (x2 upsamples)
constexpr int TH1 = 14, TW1 = 24;            // ... for a step in (1/2, 3/5]: the same 11 x 17 footprint bound (15 rows x 0.6 = 9, 25 columns x 0.6 = 15)
constexpr int LR = 11, LC = 17;              // source footprint of a tile (rows x columns), see the head of the file
constexpr int RUNS = 12, MPX = RUNS * 16;    // 192 >= LR * LC = 187: three runs per wave (with 16 x 32 tiles, 224 footprint pixels and 4 / 3
                                             // runs per wave the kernel needed ~270 registers: accumulators in scratch inside the slab loop)
constexpr int CP = 32;                       // output channels per pass: 18 column blocks of 16 (9 taps x 2)
// bytes per footprint pixel in LDS: [32 bf16 hi | 32 bf16 lo | 32 B pad].  160, as in the conv kernels: ds_read_b128 serves the NON-contiguous
// lane groups {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS): with 144-byte rows every fragment read was a 2-way bank conflict
// (SQ_LDS_BANK_CONFLICT, profiles/r04_bf16x3_pmc_sq_upconv.txt), and the loader's pixel swizzle below is the one made for 160
constexpr int AROW = 160;
constexpr int A_BYTES = MPX * AROW;          // 30 720
constexpr int W_BYTES = 9 * CP * 128;        // 36 864
constexpr int STAGE = A_BYTES + W_BYTES;     // 67 584
constexpr int CLD = 3 * CP + 4;              // G tile pixel pitch (floats): three taps x 32 channels + pad
// G tile ROW pitch (floats): the last pixel's 4 pad floats are overlapped by the next row, which makes the pitch 32 banks mod 64 -- two
// adjacent source rows x 8 channel quads then fill the 64 banks exactly (the gather's lane roles put two output rows into each lane group)
constexpr int GRP = LC * CLD - 4;
static_assert(GRP % 64 == 32, "G tile row pitch");
constexpr int C_BYTES = MPX * CLD * 4;       // 76 800
// LDS: [ G tile | stage 1 | column table ]; stage 0 aliases the G tile (main loop and gather alternate), stage 1 is never touched by the
// gather: the NEXT pass's first slab lands there while this pass is gathered (a pass therefore starts on stage 1)
constexpr int S1_OFF = C_BYTES;
constexpr int TAB_OFF = S1_OFF + STAGE;      // 144 384
constexpr int NTAB = TW0 + 2;                // output columns x0 - 1 .. x0 + TW (the wider tile's count)
constexpr int SMEM_BYTES = TAB_OFF + NTAB * 16;
constexpr int NDMA = 5;                      // weight pieces per wave and slab (36 over 8 waves: 5 or 4)
static_assert(LR * LC <= MPX && STAGE <= C_BYTES && SMEM_BYTES <= 160 * 1024, "LDS layout");
}  // namespace upc

struct UpconvParams {
  const float* xu;  // low-resolution source, NHWC
  int uH, uW, ldxu, C;
  long long xu_bstride;
  const void* w;    // packed [cout rows][9 taps][C] image
  const float* bias;
  const float* add;  // pre-activation addend [n, H, W, ld_add >= Cout] (may alias y: every thread reads what it writes), or null
  int ld_add;
  int act;
  float* y;
  int H, W, ldy, Cout;
  long long y_bstride;
  float usy, usx;
  int tiles_x, tiles_y, npass, pass_groups;
};

// s * a + c as FMAs (the library is built with -ffp-contract=off; this kernel has no bit-identical twin, and the gather is VALU-bound)
__device__ __forceinline__ f32x4 fma4(float s, const f32x4 a, const f32x4 c) {
  f32x4 r;
  r.x = __builtin_fmaf(s, a.x, c.x);
  r.y = __builtin_fmaf(s, a.y, c.y);
  r.z = __builtin_fmaf(s, a.z, c.z);
  r.w = __builtin_fmaf(s, a.w, c.w);
  return r;
}

template <int PREC, int TH, int TW>
__global__ void __launch_bounds__(512) upconv3x3_kernel(const UpconvParams p) {
  using namespace upc;
  constexpr int SEG = TW / 4;  // output pixels per thread in the gather
  __shared__ __attribute__((aligned(1024))) char smem[SMEM_BYTES];
  float* const csm = reinterpret_cast<float*>(smem);
  f32x4* const tab = reinterpret_cast<f32x4*>(smem + TAB_OFF);

  // ---- XCD-aware block -> (tile, pass group); the pass groups of a tile are neighbours (they share the footprint rows in L2) ----
  int t = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = t & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
  }
  const int pgrp = t % p.pass_groups;
  t /= p.pass_groups;
  const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, n_img = t / (p.tiles_x * p.tiles_y);
  const int y0 = ty * TH, x0 = tx * TW;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: the role branches below are s_cbranch)
  const int m16 = lane & 15, g = lane >> 4;
  // MFMA roles: wave = (row slot: three of the twelve footprint runs) x (column slot: nine of the pass's eighteen 16-column blocks)
  const int nslot = wave >> 2, mslot = wave & 3;
  const int run0 = 3 * mslot;
  // gather roles: wave = (8-pixel segment, half of the tile's rows); lane = (row, channel quad)
  const int seg = wave & 3;

  // ---- source footprint origin and the column table -----------------------------------------------------------------------
  const int rbase = ac_tap(max(y0 - 1, 0), p.usy, p.uH).i0;
  const int cbase = ac_tap(max(x0 - 1, 0), p.usx, p.uW).i0;
  if (tid < NTAB) {
    const int xx = x0 - 1 + tid;
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
  // a wave's gather walks ONE 8-pixel segment under the three kx shifts: ten table entries, wave-uniform and fixed for the kernel.  Lane l
  // (< 10) of every wave keeps entry 8 seg + l; the walk fetches what a step needs with v_readlane (read from LDS inside the walk every
  // step waited ~100 cycles for its entry: 7.4 k cycles per tap; as thirty scalars they overflowed the SGPR file into scratch)
  int vci, vw0, vw1;
  {
    // (three scalar loads through a volatile pointer: as one f32x4 load + readlane of its elements hipcc 7.2 read ONLY element x and
    //  used it for all three)
    const volatile int* const te = reinterpret_cast<const volatile int*>(tab + SEG * seg + min(lane, SEG + 1));
    vci = te[0];
    vw0 = te[1];
    vw1 = te[2];
  }

  // ---- footprint loader: item = (footprint pixel prow + 64 it, 4 channels `chunk`), fp32 -> bf16 hi / lo on the way to LDS -------------
  const int chunk = tid & 7, prow_lin = tid >> 3;
  const int prow = (prow_lin & ~3) | ((prow_lin & 1) << 1) | ((prow_lin >> 1) & 1);  // (bank spread of the ds_write pairs)
  constexpr int NIT = 3;  // 192 footprint pixels x 8 chunks / 512 threads
  unsigned hoff[NIT];
  {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int px = prow + 64 * it;
      const int r = px / LC, c = px - r * LC;
      const int gr = min(rbase + r, p.uH - 1), gc = min(cbase + c, p.uW - 1);  // (behind the footprint / the image: any valid pixel)
      hoff[it] = (unsigned)(((gr * p.uW + gc) * p.ldxu + chunk * 4) * 4);
    }
  }
  // (compiler-managed loads: an inline-asm load whose destination is waited for by a LATER asm statement is only safe while the
  //  compiler never copies the value in between -- it may, the asm "defined" it -- see the gather below)
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

  // ---- weight DMA: piece q = (tap q >> 2, 8-row group q & 3) of the pass's 32 rows; wave w moves pieces w, w + 8, ... ------------------
  // (scalar base + ONE per-lane 32-bit offset: the piece / slab / pass arithmetic stays on the scalar unit)
  const int dr = lane >> 3, dsl = lane & 7;
  const long long w_row_bytes = 9LL * p.C * 4;
  const unsigned wlane = (unsigned)(dr * (int)w_row_bytes + dsl * 16);
  auto dma_w = [&](int pass, int cc, int stage) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int q = wave + 8 * i;
      if (q < 36) {  // wave-uniform
        const int tap = q >> 2, rg = q & 3;
        const char* src = reinterpret_cast<const char*>(p.w) + (long long)(pass * CP + rg * 8) * w_row_bytes + ((long long)tap * p.C + cc * BK) * 4;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(smem + stage * S1_OFF + A_BYTES + q * 1024));
        // s_nop 4: the scalar operands may have been restored from a spill lane (v_readlane) right in front of this statement; a VMEM
        // instruction reading an SGPR that a VALU instruction wrote needs 5 wait states, and the hazard recognizer does not look into
        // inline asm (seen: the bf16 instantiation fetched some weight pieces from a stale address)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(wlane), "s"(src) : "memory");
      }
    }
  };

  // ---- fragments ---------------------------------------------------------------------------------------------------------
  const int key = (m16 >> 1) & 7;
  const int a_off = (run0 * 16 + m16) * AROW + g * 16;
  const int b_off_hi = A_BYTES + (nslot * 9 * 16 + m16) * 128 + ((g ^ key) << 4);
  const int b_off_lo = A_BYTES + (nslot * 9 * 16 + m16) * 128 + (((4 + g) ^ key) << 4);
  auto mma = [&](f32x4& c, const bf16x8& xh, const bf16x8& xl, const bf16x8& wh, const bf16x8& wl) {
    // weights as the A operand: D = [16 channels x 16 footprint pixels], lane (pixel m16, g) holds channels 4g .. 4g + 3 of its pixel --
    // one 16-byte LDS write per accumulator into the G tile, in the float4-per-channel-quad layout the gather reads
    if constexpr (PREC == PRV2_PREC_BF16X3) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, c, 0, 0, 0);
    }
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, c, 0, 0, 0);
  };

  const int cslabs = p.C / BK;
  const long long img_y = (long long)n_img * p.y_bstride;

  // ---- first pass: slab 0 -> stage 1 ---------------------------------------------------------------------------------------------------
  if (pgrp < p.npass) {
    dma_w(pgrp, 0, 1);
#pragma unroll
    for (int it = 0; it < NIT; ++it) load_a_async(0, it);
#pragma unroll
    for (int it = 0; it < NIT; ++it) store_a(1, it);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the weight DMA: asm issued it, asm waits for it)
  }

  for (int pass = pgrp; pass < p.npass; pass += p.pass_groups) {
    f32x4 acc[3][9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int j = 0; j < 9; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifdef UPC_ABL_NOMAIN  // (timing ablations: results wrong)
    for (int cc = 0; cc < 0; ++cc) {
#else
    for (int cc = 0; cc < cslabs; ++cc) {
#endif
      const int st = (cc + 1) & 1;  // slab 0 of a pass sits in stage 1
      // my DMAs / stores of slab cc are done (slab 0: waited for where they were issued); behind the barrier everyone's are, and everyone is
      // done reading the other stage.  No vmcnt wait in front of slab 0: the previous pass's output stores are still being acknowledged
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
      for (int j = 0; j < 9; ++j) {
        if (j + 1 < 9) read_b((j + 1) & 1, j + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 3; ++a) mma(acc[a][j], ah[a], al[a], bh[j & 1], bl[j & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (j == 0 && more) {  // the next slab's requests go out BEHIND the first MFMA block: the fragment reads above are what the matrix
                               // pipe waits for after the barrier (both waves of a SIMD arrive together), not these
          dma_w(pass, cc + 1, st ^ 1);
#pragma unroll
          for (int it = 0; it < NIT; ++it) load_a_async(cc + 1, it);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (j == 5 && more) {  // the next slab's footprint rows have had ~2/3 of a slab to arrive
#pragma unroll
          for (int it = 0; it < NIT; ++it) store_a(st ^ 1, it);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // stage 0 becomes the G tile; stage 1 is free
    const int next = pass + p.pass_groups;
    const bool has_next = next < p.npass;  // block-uniform
    if (has_next) dma_w(next, 0, 1);       // the next pass's weights fly during the whole gather

    // ---- gather: kernel row ky = taps 3 ky .. 3 ky + 2 through LDS -----------------------------------------------------------------
    // (its per-lane roles are recomputed HERE from an opaque copy of the lane id: hoisted out of the pass loop they only sit in
    //  registers -- then in scratch -- during the main loop)
    int lane_g = lane;
    asm volatile("" : "+v"(lane_g));
    // (output row, channel quad) of a lane: every ds_read_b128 lane group -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, the same + 32 -- holds
    // TWO adjacent output rows x all 8 quads: their source rows are the same (one address per quad: broadcast) or adjacent (GRP: the other
    // 32 banks).  With quad = lane & 7, row = lane >> 3 every read of the walk was a 2-way conflict.
    const int seg4 = (lane_g >> 2) & 7;
    const int quad = ((seg4 & 2) << 1) + (lane_g & 3), prow_t = ((0xD728 >> (2 * seg4)) & 3) + 4 * (lane_g >> 5) + 8 * (wave >> 2);
    const int m16g = lane_g & 15, gg = lane_g >> 4;
    int gpx[3];  // G tile offset (floats) of this lane's footprint pixel in each of the wave's three runs
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int px = (run0 + a) * 16 + m16g;
      gpx[a] = px * CLD - 4 * (px / LC);
    }
    f32x4 o[SEG];
#pragma unroll
    for (int xi = 0; xi < SEG; ++xi) o[xi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ch0 = pass * CP + 4 * quad;
    const int nvalid = min(max(p.Cout - ch0, 0), 4);
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    f32x4 av[SEG];
#pragma unroll
    for (int xi = 0; xi < SEG; ++xi) av[xi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const int b = nslot * 9 + j, tp = b >> 1;
        if (tp / 3 == ky) {  // wave-uniform
          const int col = (tp - 3 * ky) * CP + (b & 1) * 16 + 4 * gg;
#pragma unroll
          for (int a = 0; a < 3; ++a) *reinterpret_cast<f32x4*>(&csm[gpx[a] + col]) = acc[a][j];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (ky == 1 && has_next) {  // the next pass's first footprint slab: requested one kernel row ahead of its store (round 5: a whole walk to arrive under;
                                  // requested in the last row its wait sat exposed in front of the output stores)
#pragma unroll
        for (int it = 0; it < NIT; ++it) load_a_async(0, it);
      }
      if (ky == 2) {  // (the accumulators of the first two kernel rows are dead: registers for what the output stage needs)
        if (p.bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nvalid) bv[e] = p.bias[ch0 + e];
        }
        if (p.add) {  // the rest of the conv's concat input, convolved by the direct kernel: requested here, added in the output stage
          const int oy_ = y0 + prow_t;
#pragma unroll
          for (int xi = 0; xi < SEG; ++xi) {
            const int ox = x0 + SEG * seg + xi;
            if ((TH >= 16 || prow_t < TH) && oy_ < p.H && ox < p.W) {
              const float* src = p.add + (long long)n_img * p.H * p.W * p.ld_add + ((long long)oy_ * p.W + ox) * p.ld_add + ch0;
              if (nvalid == 4) av[xi] = *reinterpret_cast<const f32x4*>(src);
              else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (e < nvalid) av[xi][e] = src[e];
              }
            }
          }
        }
      }
      const int yy = y0 + prow_t + ky - 1;
      const bool vy = (unsigned)yy < (unsigned)p.H;
      const AxisTap ay = ac_tap(min(max(yy, 0), p.H - 1), p.usy, p.uH);
      const float wy0 = vy ? ay.w0 : 0.f, wy1 = vy ? ay.w1 : 0.f;
      const unsigned g0 = (unsigned)(size_t)(csm + (ay.i0 - rbase) * GRP + 4 * quad);
      const unsigned g1 = (unsigned)(size_t)(csm + (ay.i1 - rbase) * GRP + 4 * quad);
      // The G reads are plain LDS loads whose ADDRESS passes through an empty asm inside the branch that needs them.  As ordinary loads
      // hipcc if-converts the walk (LDS is always dereferenceable), issues every read of the unrolled walk up front and spills ~800
      // registers; the opaque address pins each read to its branch.  NOT inline-asm reads with a separate asm wait (the pattern of the
      // conv kernels' halo loads): between the two statements the compiler may copy the destination registers -- the issuing asm
      // "defined" them -- and did (v_mov of registers whose data had not landed).  Here the compiler's own counted waits stand in front
      // of the uses.
      typedef const __attribute__((address_space(3))) f32x4* lds_f32x4;
      auto issue3 = [&](int c, f32x4 (&v0)[3], f32x4 (&v1)[3]) {  // source column c of the round's three taps (tap = a constant offset)
        const unsigned off = (unsigned)(c * CLD * 4);  // (scalar)
        unsigned a0 = g0 + off, a1 = g1 + off;
        asm volatile("" : "+v"(a0), "+v"(a1));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#ifdef UPC_ABL_NOREAD  // (timing ablation: the walk without its LDS reads)
          v0[k] = f32x4{(float)a0, 0.f, 0.f, 0.f};
          v1[k] = f32x4{(float)a1, 0.f, 0.f, 0.f};
#else
          v0[k] = *(lds_f32x4)(size_t)(a0 + k * CP * 4);
          v1[k] = *(lds_f32x4)(size_t)(a1 + k * CP * 4);
#endif
        }
      };
#ifndef UPC_ABL_NOWALK
      // The three taps of the kernel row walk the SAME source columns (tap kx reaches table entry i at its pixel i - kx), so they walk
      // together: one wave-uniform step per table entry, three independent chains per step, one wait per column change.
      f32x4 lc[3], ln[3], q0[3], q1[3];
      int c = __builtin_amdgcn_readlane(vci, 0);
      {
        f32x4 a0[3], a1[3];
        issue3(c, a0, a1);
        issue3(c + 1, q0, q1);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          lc[k] = fma4(wy0, a0[k], wy1 * a1[k]);
          ln[k] = fma4(wy0, q0[k], wy1 * q1[k]);
        }
        issue3(c + 2, q0, q1);  // one column ahead of the walk
      }
#pragma unroll
      for (int i = 0; i < SEG + 2; ++i) {
        const int ci = __builtin_amdgcn_readlane(vci, i);
        if (ci != c) {  // wave-uniform: the walk enters the next source column (scale <= 1/2: one step at most)
          c = ci;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            lc[k] = ln[k];
            ln[k] = fma4(wy0, q0[k], wy1 * q1[k]);
          }
          issue3(c + 2, q0, q1);
        }
        const float w0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vw0, i)), w1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vw1, i));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int xi = i - k;
          if (xi >= 0 && xi < SEG) {
            o[xi] = fma4(w0, lc[k], fma4(w1, ln[k], o[xi]));
            asm volatile("" : "+v"(o[xi]));  // (pins the update here: hipcc otherwise sinks all of them behind the walk and keeps every step's operands alive)
          }
        }
      }
#endif
      if (ky == 2 && has_next) {  // the next pass's first footprint slab -> stage 1 (its weights: issued before the gather); all of it is
                                  // waited for HERE, in front of the output stores, so that the next pass starts without a vmcnt wait
#pragma unroll
        for (int it = 0; it < NIT; ++it) store_a(1, it);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the DMA of the next pass's weights)
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the G tile is rewritten by the next kernel row / the next pass's slab 1
    }

    // ---- output: 8 pixels x 4 channels per thread; the 8 quads of a pixel = one 128-byte row segment ------------------------------------
    const int oy = y0 + prow_t;
    dispatch_act(p.act, [&](auto act_c) {
#pragma unroll
      for (int xi = 0; xi < SEG; ++xi) {
        const int ox = x0 + SEG * seg + xi;
        // (the arithmetic stays OUTSIDE the exec-masked branch: hipcc's packed-fp32 code under exec masks gave intermittently wrong
        //  values in tap_gather_kernel, profiles/r04_experiments.txt #1 -- only the store is predicated)
        f32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = act_apply_bf(o[xi][e] + bv[e] + av[xi][e], decltype(act_c)::value);
        if ((TH >= 16 || prow_t < TH) && oy < p.H && ox < p.W && nvalid > 0) {
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

}  // namespace prv2

using namespace prv2;

static inline bool al16u(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static bool upconv_shape_ok(const prv2_ups_src* u, int32_t n, int32_t h, int32_t w, int32_t cout, int32_t prec) {
  return u && u->x && n > 0 && h >= 2 && w >= 2 && cout > 0 && (prec == PRV2_PREC_BF16X3 || prec == PRV2_PREC_BF16) && u->channels >= 32 &&
         u->channels % 32 == 0 && u->ld % 4 == 0 && u->ld >= u->channels && u->h >= 1 && u->w >= 1 && ac_scale(u->h, h) <= 0.6f &&
         ac_scale(u->w, w) <= 0.6f && (long long)u->h * u->w * u->ld < (1LL << 29) && (long long)roundup(cout, 128) * 9 * u->channels < (1LL << 29);
}

extern "C" int prv2_upconv3x3_supported(const prv2_ups_src* u, int32_t n, int32_t h, int32_t w, int32_t cout, int32_t prec) {
  return upconv_shape_ok(u, n, h, w, cout, prec) ? 1 : 0;
}

extern "C" int prv2_upconv3x3(const prv2_ups_src* u, const void* w_packed, const float* bias, const float* add, int32_t ld_add, int32_t n, int32_t h,
                              int32_t w, int32_t cout, int32_t act, int32_t prec, float* y, int32_t ldy, int64_t y_bstride, void* stream) {
  PRV2_REQUIRE(upconv_shape_ok(u, n, h, w, cout, prec),
               "upconv3x3: layer not covered (bf16 modes, channels %% 32 == 0, source step (h_in - 1) / (h_out - 1) <= 0.6 in both directions)");
  PRV2_REQUIRE(w_packed && y && al16u(u->x) && al16u(w_packed) && al16u(y) && ldy % 4 == 0 && ldy >= cout && u->bstride % 4 == 0 && y_bstride % 4 == 0,
               "upconv3x3: 16-byte aligned NHWC rows (ldy=%d)", ldy);
  PRV2_REQUIRE((long long)h * w * ldy < (1LL << 31), "upconv3x3: image too large");
  PRV2_REQUIRE(!add || (ld_add >= cout && ld_add % 4 == 0 && al16u(add)), "upconv3x3: addend rows must be 16-byte aligned (ld_add=%d)", ld_add);
  UpconvParams p = {};
  p.xu = u->x; p.uH = u->h; p.uW = u->w; p.ldxu = u->ld; p.C = u->channels;
  p.xu_bstride = u->bstride ? u->bstride : (long long)u->h * u->w * u->ld;
  p.w = w_packed; p.bias = bias; p.act = act; p.add = add; p.ld_add = ld_add;
  p.y = y; p.H = h; p.W = w; p.ldy = ldy; p.Cout = cout;
  p.y_bstride = y_bstride ? y_bstride : (long long)h * w * ldy;
  p.usy = ac_scale(u->h, h); p.usx = ac_scale(u->w, w);
  // tile shape by the source step: the footprint bound (11 x 17 source pixels) holds for 16 x 28 outputs up to 1/2 pixel per output pixel
  // (x2 upsamples) and for 14 x 24 outputs up to 3/5 (DepthAnything's 256 -> 448 head resolution under a V1 fusion decoder)
  const bool wide = p.usy <= 0.5f && p.usx <= 0.5f;
  const int th = wide ? upc::TH0 : upc::TH1, tw = wide ? upc::TW0 : upc::TW1;
  p.tiles_x = (int)cdiv(w, tw); p.tiles_y = (int)cdiv(h, th);
  p.npass = (int)cdiv(cout, upc::CP);
  const long long tiles = (long long)n * p.tiles_x * p.tiles_y;
  // few tiles (the low pyramid levels): the passes of a tile are spread over workgroups until the chip has two rounds of them
  const char* const ge = getenv("PRV2_UPCONV_GROUPS");  // A/B switch
  long long groups = ge ? atoll(ge) : cdiv(512, tiles);
  groups = groups < 1 ? 1 : (groups > p.npass ? p.npass : groups);
  p.pass_groups = (int)groups;
  const dim3 grid((unsigned)(tiles * groups));
  hipStream_t s = (hipStream_t)stream;
  if (prec == PRV2_PREC_BF16X3) {
    if (wide) hipLaunchKernelGGL((upconv3x3_kernel<PRV2_PREC_BF16X3, upc::TH0, upc::TW0>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((upconv3x3_kernel<PRV2_PREC_BF16X3, upc::TH1, upc::TW1>), grid, dim3(512), 0, s, p);
  } else {
    if (wide) hipLaunchKernelGGL((upconv3x3_kernel<PRV2_PREC_BF16, upc::TH0, upc::TW0>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((upconv3x3_kernel<PRV2_PREC_BF16, upc::TH1, upc::TW1>), grid, dim3(512), 0, s, p);
  }
  set_kernel("upconv3x3_kernel", 32, prec);
  PRV2_LAUNCH_CHECK("upconv3x3");
  return 0;
}
