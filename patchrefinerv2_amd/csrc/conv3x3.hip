// 3x3 / stride 1 / pad 1 convolution with an LDS-staged halo tile (the dominant FLOPs of the path:
// the BiDirectionalFusion / FusionUnet / DPT-head convs).
//
// Workgroup = 8 x 32 output pixels x BN output channels, 512 threads = 8 waves (4 along pixels x 2
// along channels; a wave owns 2 spatial rows x 32 columns x BN/2 channels = 2 x NJ MFMA tiles).
// Per 32-channel slab the (8+2) x (32+2) input halo is staged ONCE (global -> registers, fused
// zero padding / ReLU / hi-lo bf16 split -> LDS) and reused by all nine taps: tap (ky, kx) is just
// a different LDS base address for the A fragments.  Compared with the generic kernel this
// removes 9x of the activation traffic and of the split arithmetic; what is left per step is the
// 16 KB (BN=128) weight tile of that tap, double buffered like the halo.
//
// LDS rows are 144 bytes (32 fp32, or 32 bf16 hi | 32 bf16 lo, + 16 B pad).  A lane's A row is the
// halo pixel under its output pixel: the 32 lanes of a wave half walk 32 consecutive halo pixels,
// so every 16-lane ds_read_b128 group touches 16 distinct 16-byte bank slots (rows distinct
// mod 16) -- conflict free for every tap shift.
//
// Sync: one barrier per (slab, tap) step.  Weights of step s+1 and (at tap 0) the halo of the
// next slab are in flight in registers under the MFMAs of step s.
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
  constexpr int B_STAGE = BN * 32;                 // floats: unpadded 128-byte rows, XOR-swizzled (LDS-DMA image)
  constexpr int CLD = BN + 4;
  constexpr int NBUF = 2;                          // weight tile of this step + the one landing for the next
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
  const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
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

  // halo item `it` (one float4 per thread) of slab cc: issued at tap `it`, stored after that tap's MFMAs.
  // Loads are unconditional (padding lanes re-read the image's first float4); zeroing + the fused
  // input ReLU happen at store time so that nothing forces an early s_waitcnt next to the load.
  auto load_a = [&](int cc, int it) {
    const bool ok = cc * BK + chunk * 4 < cin4 && a_off[it] >= 0;
    ra[it] = *reinterpret_cast<const f32x4*>(ok ? img + a_off[it] + cc * BK : img);
  };
  auto store_a = [&](int cc, int abuf, int it) {
    const int hp = prow + 64 * it;
    const bool ok = cc * BK + chunk * 4 < cin4 && a_off[it] >= 0;
    if (hp < HALO) stage_a<PREC>(As + abuf * A_STAGE + hp * LDS_LD, chunk, floor4(zero_unless(ra[it], ok), x_floor));
  };
  // Weight tile of step s via LDS-DMA (global_load_lds_dwordx4): 64 lanes x 16 B = 8 rows x 128 B land
  // contiguously in LDS, no VGPRs and no ds_write.  The packed weights are pre-swizzled in HBM
  // (prv2_pack_conv_weight) so that the linear copy IS the conflict-free XOR image.
  const int dma_row = lane >> 3, dma_slot = lane & 7;
  const float* wdma = reinterpret_cast<const float*>(p.w) + ((long long)tile_n * BN + dma_row) * w_row_stride + dma_slot * 4;
  auto dma_b = [&](int s, int bbuf) {
    const int cc = s / 9, tap = s - cc * 9;
    const float* wsrc = wdma + (long long)tap * p.Cin_pad + cc * BK;
    if (BN == 32 && wave >= 4) return;  // 32 rows = 4 pieces
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int piece = wave * ND + i;  // 8 rows each
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(wsrc + (long long)(piece * 8) * w_row_stride),
          (__attribute__((address_space(3))) void*)(Bs + bbuf * B_STAGE + piece * 256), 16, 0, 0);
    }
  };

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // lane's halo pixel for its two output rows at tap (0,0): row 2*wm + i, column r32
  const int a_pix0 = (NI * wm) * HW_ + r32;
  auto compute = [&](int abuf, int bbuf, int tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const char* Ab = reinterpret_cast<const char*>(As + abuf * A_STAGE + (a_pix0 + ky * HW_ + kx) * LDS_LD) + half * 16;
    const char* Bb = reinterpret_cast<const char*>(Bs + bbuf * B_STAGE + (wn * (BN / WN) + r32) * 32);
    const char* a_row[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) a_row[i] = Ab + i * HW_ * LDS_LD * 4;
    const char* b_row[NJ];
    int b_swz[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      b_row[j] = Bb + j * 32 * 128;
      b_swz[j] = ((wn * (BN / WN) + j * 32 + r32) >> 1) & 7;
    }
    mma_slab<NJ, PREC, true, NI>(acc, a_row, b_row, b_swz, half * 16);
  };

  // ---- pipeline: one barrier per (slab, tap) step -------------------------------------------------
  // Measured alternatives that did NOT pay on MI355X (tools/ab_conv.sh, same box, 512->256 @224^2 x27):
  //   weights through registers + ds_write instead of LDS-DMA: equal within 1 %;
  //   3 weight buffers, DMA two steps ahead, hand-counted vmcnt + raw s_barrier: +1 %;
  //   fragment reads issued before the VMEM instructions: -1.5 %;  s_setprio around the MFMAs: 0.
  // Ablation: without any global->LDS traffic the same loop reaches 607 TF (73 % of the bf16x3 peak).
#pragma unroll
  for (int it = 0; it < A_IT; ++it) load_a(0, it);
  dma_b(0, 0);
#pragma unroll
  for (int it = 0; it < A_IT; ++it) store_a(0, 0, it);
  __syncthreads();  // (also drains the LDS-DMA: with a DMA in flight the barrier's fence waits vmcnt(0))
  for (int cc = 0; cc < cchunks; ++cc) {
    // No uniform branches around the loads (hipcc would drain vmcnt at each one): the last slab / last step
    // simply re-load clamped (already cached) data into buffers nobody reads afterwards.
    const int ccn = cc + 1 < cchunks ? cc + 1 : cc;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int s = cc * 9 + tap;
      dma_b(s + 1 < nsteps ? s + 1 : s, (s + 1) & 1);  // lands during the MFMAs below
      if (tap < A_IT) load_a(ccn, tap);
      __builtin_amdgcn_sched_barrier(0);  // keep the loads above the MFMAs (hipcc sinks them to their use)
      compute(cc & 1, s & 1, tap);
      __builtin_amdgcn_sched_barrier(0);
      // the other halo buffer was last read in slab cc-1: free since this slab's first barrier
      if (tap < A_IT) store_a(ccn, (cc + 1) & 1, tap);
      __syncthreads();
    }
  }

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

void launch_conv3x3_halo(IgemmParams& p, int prec, hipStream_t s) {
  const int tiles = p.N * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
#define PRV2_LAUNCH_HALO(BN_, PREC_) \
  hipLaunchKernelGGL((conv3x3_halo_kernel<BN_, PREC_>), dim3(tiles * p.tiles_n), dim3(512), 0, s, p)
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
