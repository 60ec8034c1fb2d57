// GatedConvUnit tail, second design (round 3): FOUR-wave workgroups, TWO of them per CU.   (included by conv3x3_gate.hip)
// OPT-IN (PRV2_W4=1).  Measured (tools/probes/gate_clock.sh, profiles/r03_power_wall.txt): 285.6 k shader cycles per tile against the
// 8-wave kernel's 308.3 k (-7.4 %), 646 against 607 TFLOP/s on all-zero operands at 2.39 GHz -- and the SAME 3.48 ms per launch on
// random operands, where the chip lowers its clock from 1.94 to 1.81 GHz: the dominant kernel sits at the board's power limit
// (~1.48 PFLOP/s of raw bf16 MFMA work), not at an issue limit.  Kept as the evidence for that, and for operands sparser than the
// synthetic benchmark's.
//
// conv3x3_c256_gate[_x2]_kernel keeps one 8-wave workgroup per CU (153 KB of LDS): its epilogue -- LayerNorm statistics, normalise,
// gate GEMM, store loop, 16 % of a tile's cycles -- and every barrier / DMA wait of its main loop run with the MFMA pipe idle
// (74.7 % busy over the kernel).  Here the same tile (8 x 16 pixels x all 256 output channels) belongs to 4 waves with 76.5 KB of LDS,
// so that a CU holds two workgroups in different phases: one's epilogue and stalls are the other's MFMA time.
//
//   * wave w = image rows 2w, 2w + 1 of the tile (2 pixel runs of 16) x ALL 256 output channels: 2 x 16 accumulators of
//     v_mfma_f32_16x16x32_bf16, computed TRANSPOSED (A operand = weights, B operand = pixels): lane (px = lane & 15, g = lane >> 4)
//     holds 4 output channels of pixel px per accumulator -- 64 channels of ONE pixel per lane, four lanes per pixel.  The
//     LayerNorm statistics are therefore lane sums + two cross-lane adds, the normalised values are split to bf16 hi / lo in
//     registers and ARE the B operand of the gate GEMM (its k index = the channel, by the output-channel order chosen below): no C
//     tile in LDS, no barrier in the epilogue except the weight pipeline's.
//   * a step = (32-channel slab, tap, half of the output channels): weight tile 128 rows x 128 B = 16 KB by LDS-DMA into three
//     rotating buffers (the DMA of step s + 2 is issued behind the barrier of step s; one step of lookahead cost 3 %); 48 MFMAs per
//     wave and step, 18 steps per slab.  The gate GEMM is 16 more steps of the same pipeline (k = 256 normalised channels from registers, weight tiles gathered
//     by the DMA from the fragment-major image of prv2_pack_gate_weight).
//   * the halo slab (10 x 18 pixels x 32 channels, pre-split "X2" input: see the head of conv3x3_gate.hip) goes global -> LDS by
//     `buffer_load_dwordx4 ... lds`: per-lane gather addresses, hardware zero fill outside the image, no registers, no VALU; ONE
//     buffer: the last tap's fragments are read at the end of step 15, the next slab's DMA is issued behind the barrier of step 16
//     (its per-lane offsets wait in LDS: 6 registers the loop does not have).  LDS rows are 128 B, the 16-byte slot q of halo
//     pixel hp at q ^ (hp & 7) (slots 0-3: bf16 hi of channels 8q..8q+7, 4-7: lo): conflict free for ds_read_b128 at every tap.
//   * output channel order: accumulator j (0..15), row m = 4g + e of the MFMA result is channel 32 (j >> 1) + 8 g + 4 (j & 1) + e,
//     i.e. a lane's accumulators 2s, 2s + 1 are the 8 CONSECUTIVE channels 32 s + 8 g .. + 7: one k-group of the gate GEMM's slab s,
//     one X2 group of ``mul``, 32 contiguous bytes of res / y (128 B per pixel over the four lanes).  The packed conv weights are the
//     ordinary image of prv2_pack_conv_weight: the DMA lanes pick the rows (and undo / redo the row-keyed slot swizzle).
//
// Arithmetic = the 8-wave kernel's up to the order of the LayerNorm partial sums: same split products in the same order per
// accumulator (slab, tap; lo*hi, hi*lo, hi*hi), two-pass statistics over 4 partial sums per pixel (other channel sets: results agree
// to ~1e-6 relative, tests/test_hip_ops.py::test_w4_gate_kernel_matches_the_eight_wave_kernel), same final stage.
#pragma once

namespace prv2 {

namespace w4 {
constexpr int BN = 256, TH = 8, TW = 16, HWP = TW + 2;
constexpr int HALO = (TH + 2) * HWP;   // 180 halo pixels
constexpr int A_BYTES = HALO * 128;     // ONE halo buffer (22.5 KB): the last tap's fragments are read two steps early, the next slab's DMA follows
constexpr int B_BYTES = 128 * 128;      // weight tile: 128 output channels x (32 k: 64 B hi | 64 B lo)
constexpr int NBUF = 3;                 // weight tiles in LDS: the DMA runs two steps ahead
constexpr int SMEM_BYTES = NBUF * B_BYTES + A_BYTES + 6 * 1024;  // 76.5 KB (the last 6 KB: the halo DMA's per-lane offsets)
constexpr int NHD = 6, NWD = 4;         // DMA instructions per wave: halo slab (wave 3: 5) / weight tile
static_assert(SMEM_BYTES * 2 <= 160 * 1024, "two workgroups per CU");
}  // namespace w4

template <int PREC>
__global__ void __launch_bounds__(256, 2) conv3x3_w4_gate_kernel(const GateConvParams gp) {
  using namespace w4;
  static_assert(PREC == PRV2_PREC_BF16X3, "bf16x3");
  __shared__ __attribute__((aligned(1024))) char smem[SMEM_BYTES];
  const IgemmParams& p = gp.c;
  char* const Bs_b = smem;
  char* const As_b = smem + NBUF * B_BYTES;

  // ---- XCD-aware block -> pixel tile (as conv3x3_c256_gate_kernel) ------------------------------------------------
  const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
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
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15, g = lane >> 4;

  // ---- halo DMA: instruction I = 4 i + wave writes LDS bytes [1024 I, 1024 I + 1024) of the slab buffer; lane = (pixel 8 I + lane / 8,
  // slot lane & 7) fetches the logical piece q = slot ^ (pixel & 7) of that pixel's 128 slab bytes in global memory
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  constexpr unsigned OOB = 0x80000000u;
  i32x4 rsrc;
  unsigned* const hoff_lds = reinterpret_cast<unsigned*>(smem + NBUF * B_BYTES + A_BYTES) + tid;  // [NHD][256]: registers are scarce in the loop
  {
    const unsigned long long img_base = (unsigned long long)(size_t)(p.x + (long long)n_img * p.x_bstride);
    rsrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)img_base);
    rsrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((img_base >> 32) & 0xffffu));
    rsrc.z = __builtin_amdgcn_readfirstlane((int)(unsigned)((((long long)p.H * p.W - 1) * p.ldx + p.Cin) * 4));
    rsrc.w = 0x00020000;
#pragma unroll
    for (int i = 0; i < NHD; ++i) {
      const int hp = (4 * i + wave) * 8 + (lane >> 3);
      const int q = (lane & 7) ^ (hp & 7);
      const int hy = hp / HWP, hx = hp - hy * HWP;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = hp < HALO && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      // X2 source: per 8 channels [16 B hi | 16 B lo]; piece q < 4: hi of group q, else lo of group q - 4
      hoff_lds[i * 256] = ok ? (unsigned)((iy * p.W + ix) * p.ldx * 4 + (q & 3) * 32 + (q >> 2) * 16) : OOB;
    }
  }
  const bool halo_tail_ok = (4 * (NHD - 1) + wave) * 8 + (lane >> 3) < HALO;  // last instruction: wave 2 half, wave 3 nothing
  auto dma_halo = [&](int cc) {
#pragma unroll
    for (int i = 0; i < NHD; ++i) {
      const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(As_b + (4 * i + wave) * 1024));
      const unsigned voff = hoff_lds[i * 256] + (unsigned)(cc * 128);  // (2^31 + cc * 128 stays out of range)
      if (i < NHD - 1 || halo_tail_ok)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(voff), "s"(rsrc) : "memory");
    }
  };

  // ---- weight DMA: instruction I = 4 wave + i writes rows 8 I .. 8 I + 7 of the 128-row tile; LDS row R = 16 jj + m holds the output
  // channel of accumulator j = 8 h + jj, MFMA row m (see the head of this file)
  const long long w_row_stride = 9LL * p.Cin_pad;  // floats per packed row
  // per-lane 32-bit byte offsets from a wave-uniform base (SGPR pair): conv rows / gate fragments
  unsigned wsrc[NWD];
  auto tile_row = [&](int i, int& c, int& slot, int& key_lds) {
    const int R = (4 * wave + i) * 8 + (lane >> 3);
    const int jj = R >> 4, m = R & 15;
    slot = lane & 7;
    c = 32 * (jj >> 1) + 8 * (m >> 2) + 4 * (jj & 1) + (m & 3);  // (half h: + 128)
    key_lds = (m >> 1) & 7;
  };
#pragma unroll
  for (int i = 0; i < NWD; ++i) {
    int c, slot, key_lds;
    tile_row(i, c, slot, key_lds);
    // conv weights: packed row c, its slots stored swizzled by (c >> 1) & 7 (pack_weight_kernel)
    wsrc[i] = (unsigned)(((long long)c * w_row_stride) * 4 + ((slot ^ key_lds ^ ((c >> 1) & 7)) << 4));
  }
  auto gate_src = [&](int i) {  // gate weights: fragment-major (gate_frag_index): 16-column block c / 16, lane (c & 15, k-group), hi / lo planes
    int c, slot, key_lds;
    tile_row(i, c, slot, key_lds);
    const int q = slot ^ key_lds;
    return (unsigned)((gate_frag_index(BN, c >> 4, 0, q >> 2) + (q & 3) * 16 + (c & 15)) * 16);
  };
  auto dma_issue = [&](const char* base, unsigned voff, int bbuf, int i) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(Bs_b + bbuf * B_BYTES + (4 * wave + i) * 1024));
    const unsigned long long b = (unsigned long long)(size_t)base;
    const unsigned long long sb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(sb) : "memory");
  };
  auto dma_conv_tile = [&](long long byte_off, int bbuf) {
#pragma unroll
    for (int i = 0; i < NWD; ++i) dma_issue(reinterpret_cast<const char*>(p.w) + byte_off, wsrc[i], bbuf, i);
  };
  auto dma_gate_tile = [&](long long byte_off, int bbuf) {
#pragma unroll
    for (int i = 0; i < NWD; ++i) dma_issue(reinterpret_cast<const char*>(gp.gate_w) + byte_off, gate_src(i), bbuf, i);
  };
  const int cslabs = p.Cin_pad / 32;
  auto conv_tile_off = [&](int cc, int tap, int h) { return ((long long)tap * p.Cin_pad + cc * 32 + (long long)h * 128 * w_row_stride) * 4; };
  auto gate_tile_off = [&](int h2, int ks) { return gate_frag_index(BN, 8 * h2, ks, 0) * 16; };

  // ---- fragment addressing ---------------------------------------------------------------------------------------
  const int w_off = m16 * 128 + ((g ^ ((m16 >> 1) & 7)) << 4);  // bf16 hi of the lane's fragment; + 2048 jj; lo: ^ 64
  const int hp0 = 2 * wave * HWP + m16;  // halo pixel of (row 2 wave, tap (0, 0)); run f, tap (dy, dx): + (f + dy) * 18 + dx
  bf16x8 xh[2], xl[2], wh[3], wl[3];
  int hp0v = hp0;  // (made opaque once per slab: 12 loop-invariant fragment addresses x hi / lo would otherwise be hoisted -- and spilled)
  auto read_x_to = [&](bf16x8 (&dh)[2], bf16x8 (&dl)[2], int tap) {
    const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int hp = hp0v + (f + dy) * HWP + dx;
      const int a = NBUF * B_BYTES + hp * 128 + ((g ^ (hp & 7)) << 4);
      dh[f] = *reinterpret_cast<const bf16x8*>(smem + a);
      dl[f] = *reinterpret_cast<const bf16x8*>(smem + (a ^ 64));
    }
  };
  auto read_w = [&](int slot, int bbuf, int jj) {
    wh[slot] = *reinterpret_cast<const bf16x8*>(smem + w_off + bbuf * B_BYTES + jj * 2048);
    wl[slot] = *reinterpret_cast<const bf16x8*>(smem + (w_off ^ 64) + bbuf * B_BYTES + jj * 2048);
  };
  // (same products, same order as the 8-wave kernel's mma(): x_lo * w_hi, x_hi * w_lo, x_hi * w_hi)
  // (the two pixel runs alternate: no MFMA waits for the one before it)
  auto mma6 = [&](f32x4& c0, f32x4& c1, const bf16x8& wh_, const bf16x8& wl_, const bf16x8 (&xh_)[2], const bf16x8 (&xl_)[2]) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh_, xl_[0], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh_, xl_[1], c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl_, xh_[0], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl_, xh_[1], c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh_, xh_[0], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh_, xh_[1], c1, 0, 0, 0);
  };

  f32x4 acc[2][16];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[f][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  PRV2_CLK_STAMP(0);
  // ---- prologue: slab 0 and the first two weight tiles -------------------------------------------------------------------
  dma_halo(0);
  dma_conv_tile(conv_tile_off(0, 0, 0), 0);
  dma_conv_tile(conv_tile_off(0, 0, 1), 1);

  // ---- main loop: step u of a slab = (tap u / 2, output-channel half u & 1), weight tile in buffer u % 3 ----------------------
  for (int cc = 0; cc < cslabs; ++cc) {
    const int ccn = cc + 1 < cslabs ? cc + 1 : cc;  // behind the last slab: the same slab again, nobody reads it (uniform DMA counts)
    const bool last = cc + 1 == cslabs;
    hp0v = hp0;
    asm volatile("" : "+v"(hp0v));
    auto step = [&](auto u_c) {
      constexpr int u = decltype(u_c)::value, tap = u >> 1, h = u & 1, bb = u % 3;
      // VMEM instructions younger than this step's weight tile (issued two steps ago): the next tile's 4 DMAs -- and at step 17 the halo
      // DMAs issued behind them at step 16 (5 on wave 3, 6 elsewhere).  At step 0 this also covers the halo slab (issued before tile 0).
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(u == 17 ? NWD + NHD - 1 : NWD) : "memory");
      if constexpr (h == 0 && tap < 8) read_x_to(xh, xl, tap);  // (tap 8: at the end of step 15)
      read_w(0, bb, 0);
      read_w(1, bb, 1);
      // the tile of step u + 2 -> the buffer step u - 1 has just left
      if constexpr (u + 2 < 18) {
        dma_conv_tile(conv_tile_off(cc, (u + 2) >> 1, (u + 2) & 1), (u + 2) % 3);
      } else {  // next slab's first tiles, or the gate GEMM's (same buffer rotation: 18 = 0 mod 3)
        if (last) dma_gate_tile(gate_tile_off(0, u + 2 - 18), (u + 2) % 3);
        else dma_conv_tile(conv_tile_off(cc + 1, 0, u + 2 - 18), (u + 2) % 3);
      }
      if constexpr (u == 16) dma_halo(ccn);  // every wave is past its last read of the slab (step 15)
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        if (jj + 2 < 8) read_w((jj + 2) % 3, bb, jj + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma6(acc[0][8 * h + jj], acc[1][8 * h + jj], wh[jj % 3], wl[jj % 3], xh, xl);
        __builtin_amdgcn_sched_barrier(0);
      }
      // the last tap's fragments leave the halo buffer one step early: behind the barrier of step 16 the buffer is free for the next slab
      if constexpr (u == 15) read_x_to(xh, xl, 8);
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
    step(std::integral_constant<int, 9>{});
    step(std::integral_constant<int, 10>{});
    step(std::integral_constant<int, 11>{});
    step(std::integral_constant<int, 12>{});
    step(std::integral_constant<int, 13>{});
    step(std::integral_constant<int, 14>{});
    step(std::integral_constant<int, 15>{});
    step(std::integral_constant<int, 16>{});
    step(std::integral_constant<int, 17>{});
  }

  // ---- epilogue, wave-local: bias, LayerNorm over the 256 channels of a pixel (4 lanes x 64), ReLU, bf16 split -------------------
  // lane's channels of accumulator j: ch0(j) + e,  ch0(j) = 32 (j >> 1) + 8 g + 4 (j & 1)
  auto ch0 = [&](int j) { return 32 * (j >> 1) + 8 * g + 4 * (j & 1); };
  if (gp.pre) {  // block-uniform: the conv's coarse half (prv2_conv3x3_ln_gate_pre; coarse_taps.hip) joins in front of the LayerNorm
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int oy = y0 + 2 * wave + f, ox = x0 + m16;
      const float* pp = gp.pre + ((long long)n_img * p.H * p.W + min(oy, p.H - 1) * p.W + min(ox, p.W - 1)) * gp.ld_pre;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        acc[f][j] += *reinterpret_cast<const f32x4*>(pp + ch0(j));
        if ((j & 3) == 3) asm volatile("" ::: "memory");  // (four loads in flight, not sixteen: registers)
      }
    }
  }
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + ch0(j));
      acc[0][j] += b;
      acc[1][j] += b;
      if ((j & 3) == 3) asm volatile("" ::: "memory");  // (four loads in flight, not sixteen: registers)
    }
  }
  float mean[2], rstd[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {  // two passes like convs.py:25-27; partial sums over the lane's channels, then over the pixel's 4 lanes
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += (acc[f][j][0] + acc[f][j][1]) + (acc[f][j][2] + acc[f][j][3]);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    mean[f] = s / (float)BN;
    float d2 = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float dx = acc[f][j][0] - mean[f], dy = acc[f][j][1] - mean[f], dz = acc[f][j][2] - mean[f], dw = acc[f][j][3] - mean[f];
      d2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    d2 += __shfl_xor(d2, 16);
    d2 += __shfl_xor(d2, 32);
    rstd[f] = 1.0f / sqrtf(d2 / (float)BN + p.ln_eps);
  }
  // normalise + activate + split: accumulators 2s, 2s + 1 -> the B fragment (k = channels 32 s + 8 g .. + 7) of the gate GEMM's slab s
  bf16x8 gh[8][2], gl[8][2];
  {
    const float act_floor = p.act == PRV2_ACT_RELU ? 0.f : -__builtin_inff();  // (host: ReLU or none in front of the gate)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const f32x4 lw0 = *reinterpret_cast<const f32x4*>(p.ln_w + ch0(2 * s)), lw1 = *reinterpret_cast<const f32x4*>(p.ln_w + ch0(2 * s) + 4);
      const f32x4 lb0 = *reinterpret_cast<const f32x4*>(p.ln_b + ch0(2 * s)), lb1 = *reinterpret_cast<const f32x4*>(p.ln_b + ch0(2 * s) + 4);
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        f32x4 v0 = (acc[f][2 * s] - mean[f]) * rstd[f] * lw0 + lb0;
        f32x4 v1 = (acc[f][2 * s + 1] - mean[f]) * rstd[f] * lw1 + lb1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v0[e] = fmaxf(v0[e], act_floor);
          v1[e] = fmaxf(v1[e], act_floor);
        }
        bf16x4 h0, l0, h1, l1;
        split_bf16(v0, h0, l0);
        split_bf16(v1, h1, l1);
        gh[s][f] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        gl[s][f] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
      asm volatile("" ::: "memory");  // (one slab's LayerNorm parameters in flight at a time: registers)
    }
  }

  // ---- gate GEMM: 16 more steps (half h2 of the gate columns, slab ks) + the final stage of each half -------------------------
  const long long img_m = (long long)n_img * p.H * p.W;
  const bool has_mul = p.mul != nullptr;  // block-uniform
  int pix[2];
  bool inside[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int oy = y0 + 2 * wave + f, ox = x0 + m16;
    inside[f] = oy < p.H && ox < p.W;
    pix[f] = min(oy, p.H - 1) * p.W + min(ox, p.W - 1);  // (outside: a clamped address, never stored)
  }
  auto gate_half = [&](auto h2_c) {
    constexpr int h2 = decltype(h2_c)::value;
    f32x4 acc2[2][8];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) acc2[f][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      constexpr int G0 = 8 * h2;  // gate step G = G0 + ks, tile buffer G % 3; tiles G, G + 1 are in flight or landed
      // (VMEM instructions younger than tile G: the 4 DMAs of tile G + 1, except at the very last step; the loads of the LayerNorm
      // phase / the first half's final stage have been consumed, its stores make the wait of step 8 stricter than needed)
      if (G0 + ks == 15) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NWD) : "memory");
      read_w(0, (G0 + ks) % 3, 0);
      read_w(1, (G0 + ks) % 3, 1);
      if (G0 + ks + 2 < 16) dma_gate_tile(gate_tile_off((G0 + ks + 2) >> 3, (G0 + ks + 2) & 7), (G0 + ks + 2) % 3);
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        if (jj + 2 < 8) read_w((jj + 2) % 3, (G0 + ks) % 3, jj + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma6(acc2[0][jj], acc2[1][jj], wh[jj % 3], wl[jj % 3], gh[ks], gl[ks]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // final stage of this half: y = mul * sigmoid(gate + bias) + res; a lane's accumulators 2t, 2t + 1 = 8 consecutive channels
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int tq = 0; tq < 4; ++tq) {
      const int cb = 128 * h2 + 32 * tq + 8 * g;
      f32x4 gb0 = {0.f, 0.f, 0.f, 0.f}, gb1 = {0.f, 0.f, 0.f, 0.f};
      if (gp.gate_bias) {
        gb0 = *reinterpret_cast<const f32x4*>(gp.gate_bias + cb);
        gb1 = *reinterpret_cast<const f32x4*>(gp.gate_bias + cb + 4);
      }
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        f32x4 m0 = {1.f, 1.f, 1.f, 1.f}, m1 = {1.f, 1.f, 1.f, 1.f}, r0 = {0.f, 0.f, 0.f, 0.f}, r1 = {0.f, 0.f, 0.f, 0.f};
        if (has_mul) {  // X2 group: [8 hi | 8 lo]; the multiplicand is hi + lo (as in the 8-wave kernel)
          const float* mp = p.mul + (img_m + pix[f]) * p.ld_mul + cb;
          const u32x4v hh = *reinterpret_cast<const u32x4v*>(mp), ll = *reinterpret_cast<const u32x4v*>(mp + 4);
          m0[0] = __builtin_bit_cast(float, hh[0] << 16) + __builtin_bit_cast(float, ll[0] << 16);
          m0[1] = __builtin_bit_cast(float, hh[0] & 0xffff0000u) + __builtin_bit_cast(float, ll[0] & 0xffff0000u);
          m0[2] = __builtin_bit_cast(float, hh[1] << 16) + __builtin_bit_cast(float, ll[1] << 16);
          m0[3] = __builtin_bit_cast(float, hh[1] & 0xffff0000u) + __builtin_bit_cast(float, ll[1] & 0xffff0000u);
          m1[0] = __builtin_bit_cast(float, hh[2] << 16) + __builtin_bit_cast(float, ll[2] << 16);
          m1[1] = __builtin_bit_cast(float, hh[2] & 0xffff0000u) + __builtin_bit_cast(float, ll[2] & 0xffff0000u);
          m1[2] = __builtin_bit_cast(float, hh[3] << 16) + __builtin_bit_cast(float, ll[3] << 16);
          m1[3] = __builtin_bit_cast(float, hh[3] & 0xffff0000u) + __builtin_bit_cast(float, ll[3] & 0xffff0000u);
        }
        if (p.res) {
          const float* rp = p.res + (img_m + pix[f]) * p.ld_res + cb;
          r0 = *reinterpret_cast<const f32x4*>(rp);
          r1 = *reinterpret_cast<const f32x4*>(rp + 4);
        }
        f32x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o0[e] = m0[e] * sigmoid_fast(acc2[f][2 * tq][e] + gb0[e]) + r0[e];
          o1[e] = m1[e] * sigmoid_fast(acc2[f][2 * tq + 1][e] + gb1[e]) + r1[e];
        }
        if (inside[f]) {
          float* dst = p.y + (long long)n_img * p.y_bstride + (long long)pix[f] * p.ldy + cb;
          *reinterpret_cast<f32x4*>(dst) = o0;
          *reinterpret_cast<f32x4*>(dst + 4) = o1;
        }
      }
    }
  };
  gate_half(std::integral_constant<int, 0>{});
  gate_half(std::integral_constant<int, 1>{});
#ifdef PRV2_GATE_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PRV2_CLK_STAMP(1);
#endif
}

}  // namespace prv2
