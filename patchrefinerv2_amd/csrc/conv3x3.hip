// 3x3 / stride 1 / pad 1 convolution with an LDS-staged halo tile (the dominant FLOPs of the path:
// the BiDirectionalFusion / FusionUnet / DPT-head convs).
//
// Workgroup = 8 x 32 output pixels x BN output channels, 512 threads = 8 waves (4 along pixels x 2
// along channels; a wave owns 2 spatial rows x 32 columns x BN/2 channels = 2 x NJ MFMA tiles).
// Per 32-channel slab the (8+2) x (32+2) input halo is staged ONCE (global -> registers, fused
// zero padding / ReLU / hi-lo bf16 split -> LDS) and reused by all nine taps: tap (ky, kx) is just
// a different LDS base address for the A fragments.  Compared with the generic kernel this
// removes 9x of the activation traffic and of the split arithmetic; what is left per step is the
// 16 KB (BN=128) weight tile of that tap.
//
// Halo rows in LDS are 144 bytes (32 fp32, or 32 bf16 hi | 32 bf16 lo, + 16 B pad).  A lane's A row is the
// halo pixel under its output pixel: the lanes of a ds_read_b128 group walk consecutive halo pixels, so
// every group touches 16 distinct 16-byte bank slots -- conflict free for every tap shift, and a tap is
// an immediate offset on one address register.  Weight rows are unpadded 128 bytes whose 16-byte slots
// are XOR-swizzled (slot q of row r at q ^ ((r >> 1) & 7)), the image a linear LDS-DMA can produce.
// 2 halo buffers + 4 weight buffers = 159.6 KB of the 160 KB.
//
// Sync: one barrier per (slab, tap) step, and every dependency is at least one whole step old when it is
// waited for: global -> LDS traffic is issued 2-3 steps ahead (s_waitcnt vmcnt(N) counts only the current
// step's loads as outstanding), fragment reads one k-step ahead, also across the barrier.
#include <stdlib.h>

#include <type_traits>

#include "igemm.h"

namespace prv2 {

constexpr int TH = 8, TW = 32;              // output tile (pixels)
constexpr int HW_ = TW + 2, HH_ = TH + 2;   // halo
constexpr int HALO = HH_ * HW_;             // 340 pixels
constexpr int A_IT = (HALO * 8 + 511) / 512;  // float4 loads per thread per slab (6)

template <int BN, int PREC>
__global__ void __launch_bounds__(512, 2) conv3x3_halo_kernel(const IgemmParams p) {
  // wave grid: BN >= 64: 4 (pixel rows pairs) x 2 (channel halves), a wave owns 2 rows x BN/2 channels;
  //            BN == 32 (cout <= 32 layers): 8 x 1, a wave owns 1 row x 32 channels -- no MFMA work on padding columns
  constexpr int WN = BN >= 64 ? 2 : 1;
  constexpr int NI = BN >= 64 ? 2 : 1;             // 32-pixel rows per wave
  constexpr int NJ = BN / (32 * WN);               // 32-channel column tiles per wave
  constexpr int ND = BN >= 64 ? BN / 64 : 1;       // LDS-DMA pieces (8 rows x 128 B) per wave per weight tile
  constexpr int A_STAGE = HALO * LDS_LD;           // floats
  constexpr int A_BYTES = A_STAGE * 4;
  constexpr int B_STAGE = BN * 32;                 // floats: unpadded 128-byte rows, XOR-swizzled (LDS-DMA image)
  constexpr int CLD = BN + 4;
  constexpr int NBUF = 4;                          // weight tiles of steps s (read), s+1 (read ahead), s+2, s+3 (landing)
  constexpr int SMEM_MAIN = 2 * A_STAGE + NBUF * B_STAGE;
  constexpr int SMEM_EPI = TH * TW * CLD + 2 * TH * TW;  // C tile + LN row statistics
  __shared__ __attribute__((aligned(16))) float smem[SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI];
  float* const As = smem;
  float* const Bs = smem + 2 * A_STAGE;

  // ---- XCD-aware block -> (pixel tile, channel tile) ----------------------------------------
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_n = bid % p.tiles_n;
  int tm = bid / p.tiles_n;
  const int tiles_x = p.tiles_x, tiles_y = (p.H + TH - 1) / TH;  // (tiles_x may exclude a remainder strip: igemm.hip)
  const int tx = tm % tiles_x;
  tm /= tiles_x;
  const int ty = tm % tiles_y;
  const int n_img = tm / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
  const int r32 = lane & 31, half = lane >> 5;
  const int chunk = tid & 7, prow = tid >> 3;  // loader role: float4 `chunk` of rows prow + 64*i

  // ---- halo loader addressing (constant over the K loop) ---------------------------------------
  const float* img = p.x + (long long)n_img * p.x_bstride + chunk * 4;
  int a_off[A_IT];      // element offset of halo pixel (prow + 64*it), or -1 when it is zero padding / unused
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int hp = prow + 64 * it;
    const int hy = hp / HW_, hx = hp - hy * HW_;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool ok = hp < HALO && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    a_off[it] = ok ? (iy * p.W + ix) * p.ldx : -1;
  }
  const long long w_row_stride = 9LL * p.Cin_pad;
  const int cchunks = p.Cin_pad / BK;
  const int nsteps = 9 * cchunks;
  const int cin4 = (p.Cin + 3) & ~3;

  f32x4 ra[A_IT];
  const float x_floor = p.relu_in ? 0.f : -INFINITY;  // fused input ReLU without a branch next to the loads
  char* const As_b = reinterpret_cast<char*>(As);
  char* const Bs_b = reinterpret_cast<char*>(Bs);

  // halo item `it` (one float4 per thread) of the next slab: issued at tap `it`, stored two taps later.
  // Loads are unconditional (padding lanes re-read the image's first float4); zeroing + the fused
  // input ReLU happen at store time so that nothing forces an early s_waitcnt next to the load.
  auto load_a = [&](int cc, int it) {
    const bool ok = cc * BK + chunk * 4 < cin4 && a_off[it] >= 0;
    ra[it] = *reinterpret_cast<const f32x4*>(ok ? img + a_off[it] + cc * BK : img);
  };
  // Same load as inline asm for the main loop, where hipcc's own bookkeeping cannot be used: with LDS-DMAs in
  // flight it treats vmcnt as unordered and waits vmcnt(0) in front of the first use -- draining the weight
  // DMAs issued since.  The register is handed back by wait_a() (the "+v" ties the value to the wait).
  auto load_a_async = [&](int cc, int it) {
    const bool ok = cc * BK + chunk * 4 < cin4 && a_off[it] >= 0;
    const float* src = ok ? img + a_off[it] + cc * BK : img;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[it]) : "v"(src) : "memory");
  };
  auto store_a = [&](int cc, int abuf, int it) {
    const int hp = prow + 64 * it;
    const bool ok = cc * BK + chunk * 4 < cin4 && a_off[it] >= 0;
    if (hp >= HALO) return;
    const f32x4 v = floor4(zero_unless(ra[it], ok), x_floor);
    // The LDS stores are inline asm on purpose: for a compiler-visible ds_write hipcc drains ALL in-flight
    // LDS-DMAs first (s_waitcnt vmcnt(0): it cannot tell the weight buffers from the halo buffers), which
    // would put the DMA latency this pipeline hides right back.  lgkmcnt is settled at the step barrier.
    const unsigned addr = (unsigned)(size_t)(As_b + abuf * A_BYTES + hp * (LDS_LD * 4)) + chunk * (PREC == PRV2_PREC_F32 ? 16 : 8);
    if constexpr (PREC == PRV2_PREC_F32) {
      asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
    } else {
      bf16x4 hi, lo;
      split_bf16(v, hi, lo);
      const unsigned long long h = __builtin_bit_cast(unsigned long long, hi), l = __builtin_bit_cast(unsigned long long, lo);
      if constexpr (PREC == PRV2_PREC_BF16X3) asm volatile("ds_write2_b64 %0, %1, %2 offset1:8" ::"v"(addr), "v"(h), "v"(l) : "memory");
      else asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(h) : "memory");
    }
  };
  // Weight tile of step s via LDS-DMA (global_load_lds_dwordx4): 64 lanes x 16 B = 8 rows x 128 B land
  // contiguously in LDS, no VGPRs and no ds_write.  The packed weights are pre-swizzled in HBM
  // (prv2_pack_conv_weight) so that the linear copy IS the conflict-free XOR image.  Every wave issues
  // ND DMAs per step (BN = 32: waves 4-7 repeat pieces 0-3 -- same bytes, same place) so that the vmcnt
  // bookkeeping below is the same number in all waves.
  const int dma_row = lane >> 3, dma_slot = lane & 7;
  const float* wdma = reinterpret_cast<const float*>(p.w) + ((long long)tile_n * BN + dma_row) * w_row_stride + dma_slot * 4;
  auto dma_b = [&](int s, int bbuf, int i) {  // piece i (of ND) of this wave
    const int cc = s / 9, tap = s - cc * 9;
    const float* wsrc = wdma + (long long)tap * p.Cin_pad + cc * BK;
    const int piece = (wave * ND + i) % (BN / 8);  // 8 rows each
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(wsrc + (long long)(piece * 8) * w_row_stride),
        (__attribute__((address_space(3))) void*)(Bs + bbuf * B_STAGE + piece * 256), 16, 0, 0);
  };

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // lane's halo pixel for its output rows at tap (0,0): row NI*wm + i, column r32
  const int a_pix0 = (NI * wm) * HW_ + r32;
  int b_key[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) b_key[j] = (((wn * (BN / WN) + j * 32 + r32) >> 1) & 7) ^ half;
  // Fragments of k-step `ks` of (halo buffer, weight buffer, tap); tap (ky, kx) is only an LDS base address.
  // Lane (r32, half) wants slot 2*ks + half (bf16 modes: + 4 for the lo plane) of its row.
  auto read_step = [&](Frags<NJ, PREC, NI>& f, int abuf, int bbuf, int tap, int ks) {
    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const char* row = As_b + abuf * A_BYTES + (a_pix0 + (ky + i) * HW_ + kx) * (LDS_LD * 4) + half * 16;
      if constexpr (PREC == PRV2_PREC_F32) {
        f.a[i] = *reinterpret_cast<const f32x4*>(row + ks * 32);
      } else {
        f.ah[i] = *reinterpret_cast<const bf16x8*>(row + ks * 32);
        if constexpr (PREC == PRV2_PREC_BF16X3) f.al[i] = *reinterpret_cast<const bf16x8*>(row + 64 + ks * 32);
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const char* row = Bs_b + bbuf * (B_STAGE * 4) + (wn * (BN / WN) + j * 32 + r32) * 128;
      if constexpr (PREC == PRV2_PREC_F32) {
        f.b[j] = *reinterpret_cast<const f32x4*>(row + (((2 * ks) ^ b_key[j]) << 4));
      } else {
        f.bh[j] = *reinterpret_cast<const bf16x8*>(row + (((2 * ks) ^ b_key[j]) << 4));
        if constexpr (PREC == PRV2_PREC_BF16X3) f.bl[j] = *reinterpret_cast<const bf16x8*>(row + (((2 * ks + 4) ^ b_key[j]) << 4));
      }
    }
  };

  // ---- pipeline ----------------------------------------------------------------------------------------
  // step s = (slab cc, tap).  In step s a wave issues: halo item `tap` of slab cc+1 (taps 0..5) and the
  // weight DMA of step s+3 (into the buffer step s-1 read); it consumes: the halo item issued two taps ago
  // (-> LDS, second halo buffer) and, at the closing barrier, the DMA issued in step s-1, i.e. only this
  // step's own loads may still be in flight: s_waitcnt vmcnt(#loads of this step).  That DMA (step s+2's
  // weights) is first read at the end of step s+1, when the fragments of step s+2 / k-step 0 are pulled
  // one k-step ahead of their MFMAs.  Before this (loads waited for within their own step, fragments read
  // right after the barrier) the same loop ran at 380-420 TF on the big layers; the history of what did
  // NOT pay: weights through registers + ds_write instead of LDS-DMA (equal), s_setprio around MFMAs (0).
  constexpr int KS = ksteps<PREC>();
  Frags<NJ, PREC, NI> fr[2];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) load_a(0, it);
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    dma_b(0, 0, i);
    dma_b(1, 1, i);
    dma_b(2, 2, i);
  }
#pragma unroll
  for (int it = 0; it < A_IT; ++it) store_a(0, 0, it);
  __syncthreads();  // full fence: drains the LDS-DMAs too (vmcnt(0))
  read_step(fr[0], 0, 0, 0, 0);
  for (int cc = 0; cc < cchunks; ++cc) {
    // No uniform branches around the loads (hipcc would drain vmcnt at each one): the last slab / last steps
    // simply re-load clamped (already cached) data into buffers nobody reads afterwards.
    const int ccn = cc + 1 < cchunks ? cc + 1 : cc;
    auto step = [&](auto tap_c) {
      constexpr int tap = decltype(tap_c)::value;
      const int s = cc * 9 + tap;
      const int s3 = s + 3 < nsteps ? s + 3 : nsteps - 1;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        // Between the three MFMA portions of a k-step go (a) the fragment reads of the NEXT k-step -- after the
        // first MFMAs, so the wait in front of those sees only reads a whole k-step old, and 2/3 of a k-step
        // before their use -- and (b) one global / LDS-store instruction group at a time: all 8 waves leave the
        // barrier together, and a burst of 3 VMEM + 2 DS-store instructions per wave in one place backs up the
        // address pipes with both waves of a SIMD queued behind it, MFMA pipe idle.  The sched_barriers pin
        // the order (hipcc would sink loads and reads to their uses).
        mma_frags<NJ, PREC, NI, 1>(acc, fr[ks & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < KS) read_step(fr[(ks + 1) & 1], cc & 1, s & 3, tap, ks + 1);
        else read_step(fr[0], tap == 8 ? (cc + 1) & 1 : cc & 1, (s + 1) & 3, (tap + 1) % 9, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma_frags<NJ, PREC, NI, 3>(acc, fr[ks & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) dma_b(s3, (s + 3) & 3, 0);
        if (ks == KS - 1 && ND > 1) dma_b(s3, (s + 3) & 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_frags<NJ, PREC, NI, 4>(acc, fr[ks & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (tap < A_IT) if (ks == 0) load_a_async(ccn, tap);
        // the other halo buffer was last read in slab cc-1: free since this slab's first barrier
        if constexpr (tap >= 2 && tap - 2 < A_IT) if (ks == KS - 1) {
          // VMEM instructions issued since load_a_async(tap - 2): the rest of its step, steps tap-1 and tap
          constexpr int newer = (ND - 1) + ND + (tap - 1 < A_IT ? 1 : 0) + ND + (tap < A_IT ? 1 : 0);
          asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ra[tap - 2]) : "n"(newer) : "memory");
          store_a(ccn, (cc + 1) & 1, tap - 2);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // LDS writes/reads of this step done; of the global loads only this step's may still fly
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((tap < A_IT ? 1 : 0) + ND) : "memory");
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
  // clamped tail DMAs must have landed before the C tile overwrites the buffers (explicit: see conv3x3_m16.hip)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // ---- epilogue through LDS (see igemm.hip) ------------------------------------------------------
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int col = (reg & 3) + 8 * (reg >> 2) + 4 * half;  // pixel column inside the 32-wide row
        const int row = (NI * wm + i) * TW + col;
        smem[row * CLD + wn * (BN / WN) + j * 32 + r32] = acc[i][j][reg];
      }
  __syncthreads();
  float* const ln_stats = smem + TH * TW * CLD;
  if (p.ln_w) {  // block-uniform
    ln_row_stats(p, smem, CLD, TH * TW, tid, ln_stats);
    __syncthreads();
  }

  constexpr int C4 = BN / 4;
  constexpr int RPP = 512 / C4;
  const int col4 = tid % C4;
  EpiCols ec;
  if (!epi_cols(p, tile_n * BN + col4 * 4, ec)) return;
  for (int rr = tid / C4; rr < TH * TW; rr += RPP) {
    const int py = rr / TW, px = rr - py * TW;
    const int oy = y0 + py, ox = x0 + px;
    if (oy >= p.H || ox >= p.W) continue;
    const f32x4 cv = *reinterpret_cast<const f32x4*>(&smem[rr * CLD + col4 * 4]);
    const long long pix = (long long)oy * p.W + ox;
    const long long m = (long long)n_img * p.H * p.W + pix;
    const long long o = (long long)n_img * p.y_bstride + pix * p.ldy + ec.co;
    epi_store(p, ec, cv, m, o, ln_stats[rr], ln_stats[TH * TW + rr]);
  }
}

bool conv3x3_halo_supported(const IgemmParams& p) {
  return p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.convt_k == 0 && p.W >= 24 && p.H >= 4 &&
         (long long)p.H * p.W * p.ldx < (1LL << 31);
}

void launch_conv3x3_halo16(IgemmParams& p, int prec, hipStream_t s);  // conv3x3_m16.hip

// bf16 modes run on the 16x16x32 MFMA shape (conv3x3_m16.hip; its buffer-addressed halo loads need the image extent
// below 2^31 bytes); PRV2_HALO_MFMA32=1 keeps them on this file's 32x32x16 kernel (A/B knob for tools/ab_conv.sh)
bool conv3x3_halo16_usable(const IgemmParams& p, int prec) {
  static const bool force32 = [] { const char* e = getenv("PRV2_HALO_MFMA32"); return e && e[0] == '1'; }();
  return prec != PRV2_PREC_F32 && !force32 && (long long)p.H * p.W * p.ldx < (1LL << 29);
}

void launch_conv3x3_halo(IgemmParams& p, int prec, hipStream_t s) {
  if (conv3x3_halo16_usable(p, prec)) return launch_conv3x3_halo16(p, prec, s);  // (with its strip, if any)
  const int tiles = p.N * ((p.H + TH - 1) / TH) * p.tiles_x;
#define PRV2_LAUNCH_HALO(BN_, PREC_) \
  hipLaunchKernelGGL((conv3x3_halo_kernel<BN_, PREC_>), dim3(tiles * p.tiles_n), dim3(512), 0, s, p)
  set_kernel("conv3x3_halo_kernel", p.Ncols > 64 ? 128 : (p.Ncols > 32 ? 64 : 32), prec);
  if (p.Ncols > 64) {
    p.tiles_n = (int)cdiv(p.Ncols, 128);
    if (prec == PRV2_PREC_F32) PRV2_LAUNCH_HALO(128, PRV2_PREC_F32);
    else if (prec == PRV2_PREC_BF16X3) PRV2_LAUNCH_HALO(128, PRV2_PREC_BF16X3);
    else PRV2_LAUNCH_HALO(128, PRV2_PREC_BF16);
#ifdef PRV2_NO_BN32
  } else if (true) {
#else
  } else if (p.Ncols > 32) {
#endif
    p.tiles_n = 1;
    if (prec == PRV2_PREC_F32) PRV2_LAUNCH_HALO(64, PRV2_PREC_F32);
    else if (prec == PRV2_PREC_BF16X3) PRV2_LAUNCH_HALO(64, PRV2_PREC_BF16X3);
    else PRV2_LAUNCH_HALO(64, PRV2_PREC_BF16);
  } else {
    p.tiles_n = 1;
    if (prec == PRV2_PREC_F32) PRV2_LAUNCH_HALO(32, PRV2_PREC_F32);
    else if (prec == PRV2_PREC_BF16X3) PRV2_LAUNCH_HALO(32, PRV2_PREC_BF16X3);
    else PRV2_LAUNCH_HALO(32, PRV2_PREC_BF16);
  }
#undef PRV2_LAUNCH_HALO
}

}  // namespace prv2
