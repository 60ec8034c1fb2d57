// Implicit-GEMM convolution / linear for gfx950 (MI355X), NHWC activations, fused epilogue.
//
//   C[m, n] = sum_{tap, c} A[pix(m) + tap, c] * W[n, tap, c]        m = output pixel, n = out channel
//
// Tile: 128 pixels x BN channels x 32 input channels per step, 256 threads = 4 waves (2x2),
// each wave owns 64 x BN/2 as 32x32 MFMA accumulators.  Operands are staged
// global -> registers -> LDS (the register hop applies the zero padding, the fused input ReLU
// and, for the split-bf16 mode, the hi/lo split), double buffered: the global loads of step
// s+1 are issued before the MFMAs of step s and written to the other LDS buffer after them,
// one barrier per step.
//
// LDS rows are 32 floats padded to 36: the MFMA fragment reads are ds_read_b128 at
// [row][8*ks + 4*half] -- with a 144-byte row stride the 16 lanes of every ds_read_b128 lane
// group land on 16 distinct 16-byte bank slots (rows distinct mod 16), i.e. conflict free;
// the staging ds_write_b128 of 8 consecutive lanes covers one contiguous 128-byte row.
//
// PRV2_PREC_F32   : v_mfma_f32_32x32x2_f32 -- exact fp32 products (k-ordered fmaf chain), the
//                   parity path.  One float4 per operand feeds 4 MFMAs: lane half h takes
//                   channels 8ks+4h..+3 so the k index of MFMA e is channel 8ks+4h+e on both
//                   operands (the k order inside a step is free as long as A and B agree).
// PRV2_PREC_BF16X3: fp32 operands split into bf16 hi + bf16 lo while staging; three
//                   v_mfma_f32_32x32x16_bf16 per k-step (hi*hi, hi*lo, lo*hi), fp32 accumulate.
//                   ~2^-17 relative per product, 5.3x the fp32-MFMA rate.
//
// Workgroup -> tile mapping is XCD aware: the 8 XCDs each get a contiguous run of tiles, and
// within a run the BN-tiles of one pixel tile are adjacent, so the A tile and its 3x3 halo
// re-reads stay inside one XCD's L2.
#include "igemm.h"

namespace prv2 {

template <int BN, int PREC>
__global__ void __launch_bounds__(256, 2) igemm_kernel(const IgemmParams p) {
  constexpr int NJ = BN / 64;  // 32-wide column sub-tiles per wave
  constexpr int NB = BN / 32;  // B rows loaded per thread per step (float4 each)
  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDS_LD + 2 * BM];  // + LN row statistics
  constexpr int STAGE = (BM + BN) * LDS_LD;  // A rows then B rows

  // ---- XCD-aware, bijective block -> tile map -------------------------------------------
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_n = bid % p.tiles_n;
  const int tile_m = bid / p.tiles_n;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, half = lane >> 5;

  // ---- per-thread A rows (pixels) ---------------------------------------------------------
  const int chunk = tid & 7;   // which float4 of the 32-channel slab
  const int row0 = tid >> 3;   // 0..31, rows row0 + 32*i
  // rows = output pixels of the column window [rx0, rx0 + rw) of every image row (rw = OW unless a strip was split off)
  const int rw = p.rw > 0 ? p.rw : p.OW;
  const long long ohw = (long long)p.OH * rw;
  const float* a_ptr[4];  // &x[n, iy0, ix0, chunk*4]; may point outside the image -- only dereferenced when in range
  int a_iy0[4], a_ix0[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    long long m = (long long)tile_m * BM + row0 + 32 * i;
    if (m < p.M) {
      int n = (int)(m / ohw);
      int rem = (int)(m - (long long)n * ohw);
      int oy = rem / rw, ox = rem - oy * rw + p.rx0;
      a_iy0[i] = oy * p.stride - p.pad;
      a_ix0[i] = ox * p.stride - p.pad_x;
      a_ptr[i] = p.x + (long long)n * p.x_bstride + ((long long)a_iy0[i] * p.W + a_ix0[i]) * p.ldx + chunk * 4;
    } else {
      a_iy0[i] = -(1 << 28);  // never in range
      a_ix0[i] = 0;
      a_ptr[i] = p.x;
    }
  }
  const long long w_row_stride = (long long)p.KH * p.KW * p.Cin_pad;
  const float* wbase = reinterpret_cast<const float*>(p.w) + ((long long)tile_n * BN + row0) * w_row_stride + chunk * 4;

  const int cchunks = p.Cin_pad / BK;
  const int nsteps = p.KH * p.KW * cchunks;
  const int cin4 = (p.Cin + 3) & ~3;  // channels [Cin, cin4) are read (finite, host guaranteed) and hit zero weights

  f32x4 ra[4], rb[NB];
  const float x_floor = p.relu_in ? 0.f : -INFINITY;  // fused input ReLU without a branch next to the loads

  unsigned okmask = 0;
  auto load_step = [&](int s) {
    const int tap = s / cchunks;
    const int cb = (s - tap * cchunks) * BK;  // channel base of this step (wave uniform)
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int tap_off = (ky * p.W + kx) * p.ldx + cb;
    const bool c_ok = cb + chunk * 4 < cin4;
    okmask = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
      const bool ok = c_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      // branch-free: select the ADDRESS (a padding lane re-reads x[0..3]); the data is zeroed at store time.
      // A branch (or any use of the data) next to the load makes hipcc wait vmcnt(0) per load:
      // 4 serialized L2 round trips per step.
      ra[i] = *reinterpret_cast<const f32x4*>(ok ? a_ptr[i] + tap_off : p.x);
      okmask |= (ok ? 1u : 0u) << i;
    }
    const float* wsrc = wbase + (long long)tap * p.Cin_pad + cb;
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(wsrc + (long long)(32 * i) * w_row_stride);
  };

  auto store_step = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      stage_a<PREC>(&smem[buf * STAGE + (row0 + 32 * i) * LDS_LD], chunk,
                    floor4(zero_unless(ra[i], (okmask >> i) & 1u), x_floor));
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      // packed weights are the LDS row image with XOR-swizzled 16-byte slots (for the LDS-DMA of conv3x3.hip):
      // physical slot `chunk` of row r holds logical slot chunk ^ ((r >> 1) & 7)
      const int r = row0 + 32 * i;
      *reinterpret_cast<f32x4*>(&smem[buf * STAGE + (BM + r) * LDS_LD + ((chunk ^ ((r >> 1) & 7)) << 2)]) = rb[i];
    }
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto compute_step = [&](int buf) {
    const char* Ab = reinterpret_cast<const char*>(smem + buf * STAGE + (wm * 64 + r32) * LDS_LD) + half * 16;
    const char* Bb = reinterpret_cast<const char*>(smem + buf * STAGE + (BM + wn * (BN / 2) + r32) * LDS_LD) + half * 16;
    const char* a_row[2] = {Ab, Ab + 32 * LDS_LD * 4};
    const char* b_row[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) b_row[j] = Bb + j * 32 * LDS_LD * 4;
    mma_slab<NJ, PREC>(acc, a_row, b_row);
  };

  load_step(0);
  store_step(0);
  __syncthreads();
  for (int s = 0; s + 1 < nsteps; ++s) {
    const int buf = s & 1;
    load_step(s + 1);      // global -> registers, in flight under the MFMAs below
    __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks the loads down to their use
    compute_step(buf);
    __builtin_amdgcn_sched_barrier(0);
    store_step(buf ^ 1);   // registers -> the other LDS buffer
    __syncthreads();
  }
  compute_step((nsteps - 1) & 1);

  // ---- epilogue: accumulators -> LDS tile -> 16-byte row-wise stores ------------------------
  // acc[i][j][reg]: row = (reg&3) + 8*(reg>>2) + 4*half, col = r32 (cdna_hip_programming.md section 3).
  // Going through LDS turns the 2-rows-x-128-B store shape of the accumulator layout into whole
  // 512-byte output rows and lets bias / residual / gate operands be read as float4.
  constexpr int CLD = BN + 4;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = wm * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
        smem[row * CLD + wn * (BN / 2) + j * 32 + r32] = acc[i][j][reg];
      }
  __syncthreads();
  float* const ln_stats = smem + 2 * (BM + BN) * LDS_LD;
  if (p.ln_w) {  // block-uniform
    ln_row_stats(p, smem, CLD, BM, tid, ln_stats);
    __syncthreads();
  }

  constexpr int C4 = BN / 4;          // float4 per tile row
  constexpr int RPP = 256 / C4;       // rows per pass
  const int col4 = tid % C4;
  EpiCols ec;
  if (!epi_cols(p, tile_n * BN + col4 * 4, ec)) return;
  const int kk = p.convt_k > 0 ? p.convt_k : 1;
  dispatch_act(p.act, [&](auto act_c) {
    for (int rr = tid / C4; rr < BM; rr += RPP) {
      const long long m = (long long)tile_m * BM + rr;
      if (m >= p.M) break;
      const f32x4 cv = *reinterpret_cast<const f32x4*>(&smem[rr * CLD + col4 * 4]);
      const int n_img = (int)(m / ohw);
      const int rem = (int)(m - (long long)n_img * ohw);
      const int oy = rem / rw, ox = rem - oy * rw + p.rx0;
      const long long pix = (long long)oy * p.OW + ox;  // dense pixel index of the image (== rem without a window)
      long long o;
      if (p.convt_k > 0) {
        o = (long long)n_img * p.y_bstride + ((long long)(oy * kk + ec.sub_y) * (p.OW * kk) + ox * kk + ec.sub_x) * p.ldy + ec.co;
      } else {
        o = (long long)n_img * p.y_bstride + pix * p.ldy + ec.co;
      }
      epi_store<decltype(act_c)::value>(p, ec, cv, (long long)n_img * p.OH * p.OW + pix, o, ln_stats[rr], ln_stats[BM + rr]);
    }
  });
}

// ---- weight packing: [cout][cin][kh][kw] (or ConvT [cin][cout][k][k]) -> [cout_pad][taps][cin_pad] ----
__global__ void __launch_bounds__(256) pack_weight_kernel(const float* __restrict__ src, const float* __restrict__ scale,
                                                          float* __restrict__ dst, int Cout, int Cin, int KH, int KW,
                                                          int convt_k, int rows, int rows_pad, int cin_pad, int prec) {
  const int taps = convt_k > 0 ? 1 : KH * KW;
  long long total = (long long)rows_pad * taps * cin_pad;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int c = (int)(idx % cin_pad);
    int t = (int)((idx / cin_pad) % taps);
    int r = (int)(idx / ((long long)cin_pad * taps));
    float v = 0.f;
    if (r < rows && c < Cin) {
      if (convt_k > 0) {
        int sub = r / Cout, co = r - sub * Cout;  // row = (ky*k + kx)*Cout + co
        v = src[((long long)c * Cout + co) * convt_k * convt_k + sub];
        if (scale) v *= scale[co];
      } else {
        v = src[((long long)r * Cin + c) * taps + t];
        if (scale) v *= scale[r];
      }
    }
    // 128-byte row image of this (row, tap, 32-channel chunk); its eight 16-byte slots are stored XOR-swizzled
    // by ((row >> 1) & 7) so that a LINEAR LDS-DMA copy of 8 rows x 128 B is bank-conflict free for ds_read_b128
    float* rowp = dst + (idx - (c & 31));
    const int key = (r >> 1) & 7;
    if (prec == PRV2_PREC_F32) {
      const int e = c & 31;  // element e lives in logical slot e/4
      rowp[(((e >> 2) ^ key) << 2) + (e & 3)] = v;
    } else {
      // logical image: [32 x bf16 hi][32 x bf16 lo]; element e of hi in slot e/8, of lo in slot 4 + e/8
      __bf16 hi = (__bf16)v;
      __bf16 lo = (__bf16)(v - (float)hi);
      __bf16* d16 = reinterpret_cast<__bf16*>(rowp);
      const int e = c & 31;
      d16[(((e >> 3) ^ key) << 3) + (e & 7)] = hi;
      d16[((((e >> 3) + 4) ^ key) << 3) + (e & 7)] = prec == PRV2_PREC_BF16X3 ? lo : (__bf16)0.0f;
    }
  }
}

// tail tile (igemm.h: has_tail_tile): row r, k = 2*tap + c for the last two input channels, same 128-byte image
__global__ void __launch_bounds__(256) pack_tail_kernel(const float* __restrict__ src, const float* __restrict__ scale,
                                                        float* __restrict__ dst, int Cout, int Cin, int rows_pad, int prec) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows_pad * 32) return;
  const int r = idx >> 5, e = idx & 31;
  float v = 0.f;
  if (r < Cout && e < 18) {
    v = src[((long long)r * Cin + (Cin - 2 + (e & 1))) * 9 + (e >> 1)];
    if (scale) v *= scale[r];
  }
  const int key = (r >> 1) & 7;
  __bf16 hi = (__bf16)v;
  __bf16 lo = (__bf16)(v - (float)hi);
  __bf16* d16 = reinterpret_cast<__bf16*>(dst + (long long)r * 32);
  d16[(((e >> 3) ^ key) << 3) + (e & 7)] = hi;
  d16[((((e >> 3) + 4) ^ key) << 3) + (e & 7)] = prec == PRV2_PREC_BF16X3 ? lo : (__bf16)0.0f;
}

static inline bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// ---------------------------------------------------------------------------------------------
// small 1x1 convs (Cin, Cout <= 64: the 32->32 gates / out_convs at full resolution).  A GEMM tile would be 3/4
// padding and the layer is HBM-bound anyway: one thread = one pixel x 8 output channels on the fp32 VALU, the
// weights decoded once per block from the packed image (fp32, or bf16 hi + lo = 16 mantissa bits) into LDS.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) conv1x1_small_kernel(const IgemmParams p, int prec) {
  __shared__ __attribute__((aligned(16))) float wl[64 * 64];  // [cin][cout_pad8]
  const int ng = (p.Cout + 7) >> 3, ncp = ng * 8;
  for (int i = threadIdx.x; i < p.Cin_pad * ncp; i += blockDim.x) {
    const int co = i % ncp, c = i / ncp;
    float v = 0.f;
    if (co < p.Cout && c < p.Cin) {
      const char* row = reinterpret_cast<const char*>(p.w) + ((long long)co * p.Cin_pad + (c & ~31)) * 4;
      const int key = (co >> 1) & 7, e = c & 31;
      if (prec == PRV2_PREC_F32) {
        v = reinterpret_cast<const float*>(row)[(((e >> 2) ^ key) << 2) + (e & 3)];
      } else {
        const __bf16* d16 = reinterpret_cast<const __bf16*>(row);
        v = (float)d16[(((e >> 3) ^ key) << 3) + (e & 7)] + (float)d16[((((e >> 3) + 4) ^ key) << 3) + (e & 7)];
      }
    }
    wl[c * ncp + co] = v;
  }
  __syncthreads();
  const int ppb = 256 / ng;  // pixels per block
  const int grp = threadIdx.x % ng, pl = threadIdx.x / ng;
  if (pl >= ppb) return;
  const float x_floor = p.relu_in ? 0.f : -INFINITY;
  EpiCols ec0, ec1;
  const bool v0 = epi_cols(p, grp * 8, ec0), v1 = epi_cols(p, grp * 8 + 4, ec1);
  for (long long m = (long long)blockIdx.x * ppb + pl; m < p.M; m += (long long)gridDim.x * ppb) {
    const float* px = p.x + m * p.ldx;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < p.Cin; c += 4) {
      const f32x4 xv = floor4(*reinterpret_cast<const f32x4*>(px + c), x_floor);
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (c + e < p.Cin) {
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(&wl[(c + e) * ncp + grp * 8]);
          const f32x4 w1 = *reinterpret_cast<const f32x4*>(&wl[(c + e) * ncp + grp * 8 + 4]);
          a0.x = __builtin_fmaf(xs[e], w0.x, a0.x);  // (explicit FMAs: the build runs with -ffp-contract=off)
          a0.y = __builtin_fmaf(xs[e], w0.y, a0.y);
          a0.z = __builtin_fmaf(xs[e], w0.z, a0.z);
          a0.w = __builtin_fmaf(xs[e], w0.w, a0.w);
          a1.x = __builtin_fmaf(xs[e], w1.x, a1.x);
          a1.y = __builtin_fmaf(xs[e], w1.y, a1.y);
          a1.z = __builtin_fmaf(xs[e], w1.z, a1.z);
          a1.w = __builtin_fmaf(xs[e], w1.w, a1.w);
        }
      }
    }
    if (v0) epi_store(p, ec0, a0, m, m * p.ldy + ec0.co);
    if (v1) epi_store(p, ec1, a1, m, m * p.ldy + ec1.co);
  }
}

// ---------------------------------------------------------------------------------------------
// k x k convs over <= 4 INPUT channels (the refiner encoders' stems on the 4-channel crop: 4 -> 32, 3x3 stride 2, at 41 x 384 x 512 per
// batch): on the implicit-GEMM kernel a k-slab is 7/8 padding and the layer ran at 8 TFLOP/s = 0.7 TB/s (0.55 ms per batch against 0.1 ms
// of HBM traffic).  Direct form on the fp32 VALU like conv1x1_small_kernel: one thread = one output pixel x 8 output channels, one 16-byte
// load per tap, weights decoded once per block from the packed image into LDS.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) conv_few_in_kernel(const IgemmParams p, int prec) {
  __shared__ __attribute__((aligned(16))) float wl[49 * 4 * 64];  // [tap][cin 4][cout_pad8]
  const int ng = (p.Cout + 7) >> 3, ncp = ng * 8, taps = p.KH * p.KW;
  for (int i = threadIdx.x; i < taps * 4 * ncp; i += blockDim.x) {
    const int co = i % ncp, c = (i / ncp) & 3, tap = i / (4 * ncp);
    float v = 0.f;
    if (co < p.Cout && c < p.Cin) {
      const char* row = reinterpret_cast<const char*>(p.w) + (((long long)co * taps + tap) * p.Cin_pad) * 4;
      const int key = (co >> 1) & 7;
      if (prec == PRV2_PREC_F32) {
        v = reinterpret_cast<const float*>(row)[(((c >> 2) ^ key) << 2) + (c & 3)];
      } else {
        const __bf16* d16 = reinterpret_cast<const __bf16*>(row);
        v = (float)d16[(((c >> 3) ^ key) << 3) + (c & 7)] + (float)d16[((((c >> 3) + 4) ^ key) << 3) + (c & 7)];
      }
    }
    wl[(tap * 4 + c) * ncp + co] = v;
  }
  __syncthreads();
  const int ppb = 256 / ng;
  const int grp = threadIdx.x % ng, pl = threadIdx.x / ng;
  if (pl >= ppb) return;
  const float x_floor = p.relu_in ? 0.f : -INFINITY;
  EpiCols ec0, ec1;
  const bool v0 = epi_cols(p, grp * 8, ec0), v1 = epi_cols(p, grp * 8 + 4, ec1);
  const long long ohw = (long long)p.OH * p.OW;
  for (long long m = (long long)blockIdx.x * ppb + pl; m < p.M; m += (long long)gridDim.x * ppb) {
    const int n = (int)(m / ohw), r = (int)(m - n * ohw), oy = r / p.OW, ox = r - oy * p.OW;
    const float* img = p.x + (long long)n * p.x_bstride;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < p.KH; ++ky) {
      const int iy = oy * p.stride - p.pad + ky;
      if ((unsigned)iy >= (unsigned)p.H) continue;
      for (int kx = 0; kx < p.KW; ++kx) {
        const int ix = ox * p.stride - p.pad_x + kx;
        if ((unsigned)ix >= (unsigned)p.W) continue;
        const f32x4 xv = floor4(*reinterpret_cast<const f32x4*>(img + ((long long)iy * p.W + ix) * p.ldx), x_floor);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        const float* wt = &wl[(ky * p.KW + kx) * 4 * ncp + grp * 8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt + e * ncp), w1 = *reinterpret_cast<const f32x4*>(wt + e * ncp + 4);
          a0.x = __builtin_fmaf(xs[e], w0.x, a0.x);  // (channels >= Cin: zero weights; the pad channels of the input are finite)
          a0.y = __builtin_fmaf(xs[e], w0.y, a0.y);
          a0.z = __builtin_fmaf(xs[e], w0.z, a0.z);
          a0.w = __builtin_fmaf(xs[e], w0.w, a0.w);
          a1.x = __builtin_fmaf(xs[e], w1.x, a1.x);
          a1.y = __builtin_fmaf(xs[e], w1.y, a1.y);
          a1.z = __builtin_fmaf(xs[e], w1.z, a1.z);
          a1.w = __builtin_fmaf(xs[e], w1.w, a1.w);
        }
      }
    }
    const long long o = (long long)n * p.y_bstride + (long long)r * p.ldy;
    if (v0) epi_store(p, ec0, a0, m, o + ec0.co);
    if (v1) epi_store(p, ec1, a1, m, o + ec1.co);
  }
}

static bool conv_few_in_supported(const IgemmParams& p) {
  return p.Cin <= 4 && p.ldx >= 4 && p.ldx % 4 == 0 && p.KH * p.KW <= 49 && p.KH * p.KW > 1 && p.Cout >= 8 && p.Cout <= 64 && p.convt_k == 0 && !p.ln_w &&
         !p.w_tail && !p.rw && (long long)p.OH * p.OW >= 4096;  // per image: the choice must not depend on the batch
}

static bool conv1x1_small_supported(const IgemmParams& p) {
  // ... and the handful-of-outputs 1x1s behind a wider input (ZoeDepth's 80 -> 4 head conv on every V1 tile: 2.4 ms per 41 x 384 x 512 on
  // the generic MFMA kernel -- a 128-column tile for 4 columns -- against 0.7 ms of HBM traffic): one thread per pixel, 8 output slots
  const bool few_outputs = p.Cout < 8 && p.Cin_pad * 8 <= 64 * 64;
  return p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.convt_k == 0 && ((p.Cin <= 64 && p.Cout <= 64 && p.Cout >= 8) || few_outputs) &&
         !p.ln_w && p.x_bstride == (long long)p.H * p.W * p.ldx &&
         p.y_bstride == (long long)p.OH * p.OW * p.ldy && (long long)p.OH * p.OW >= 4096;  // per image: the choice (fp32 VALU vs MFMA) must not depend on the batch
}


}  // namespace prv2

using namespace prv2;

static inline int gemm_rows(int cout, int convt_k) { return convt_k > 0 ? convt_k * convt_k * cout : cout; }

extern "C" int64_t prv2_packed_weight_bytes(int32_t cout, int32_t cin, int32_t kh, int32_t kw, int32_t convt_k,
                                            int32_t prec) {
  int64_t taps = convt_k > 0 ? 1 : (int64_t)kh * kw;
  const int64_t rows_pad = roundup(gemm_rows(cout, convt_k), 128);
  return rows_pad * taps * roundup(cin, BK) * 4 + (has_tail_tile(cin, kh, kw, convt_k, prec) ? rows_pad * 128 : 0);
}

extern "C" int prv2_pack_conv_weight(const float* w_src, const float* bn_scale, void* w_packed, int32_t cout, int32_t cin,
                                     int32_t kh, int32_t kw, int32_t convt_k, int32_t prec, void* stream) {
  PRV2_REQUIRE(prec >= PRV2_PREC_F32 && prec <= PRV2_PREC_BF16, "pack_conv_weight: unknown precision mode %d", prec);
  PRV2_REQUIRE(w_src && w_packed && cout > 0 && cin > 0 && kh > 0 && kw > 0, "pack_conv_weight: bad arguments");
  PRV2_REQUIRE(convt_k == 0 || (kh == convt_k && kw == convt_k), "pack_conv_weight: convt needs kh == kw == k");
  int rows = gemm_rows(cout, convt_k), rows_pad = (int)roundup(rows, 128), cin_pad = (int)roundup(cin, BK);
  int64_t total = (int64_t)rows_pad * (convt_k > 0 ? 1 : kh * kw) * cin_pad;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w_src, bn_scale,
                     (float*)w_packed, cout, cin, kh, kw, convt_k, rows, rows_pad, cin_pad, prec);
  PRV2_LAUNCH_CHECK("pack_conv_weight");
  if (has_tail_tile(cin, kh, kw, convt_k, prec)) {
    hipLaunchKernelGGL(pack_tail_kernel, dim3((rows_pad * 32 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_src, bn_scale,
                       (float*)w_packed + total, cout, cin, rows_pad, prec);
    PRV2_LAUNCH_CHECK("pack_conv_weight(tail)");
  }
  return 0;
}

namespace prv2 {
int conv2d_impl(const prv2_conv_desc* d, const float* x, const void* w_packed, const float* bias, const float* ln_weight, const float* ln_bias,
                const float* gamma, const float* mul, const float* res, const float* res2, float* y, void* stream, const void* gate_w,
                const float* gate_bias, const prv2_ups_src* ups, const float* tail1, const float* tail2, int tail_h, int tail_w, const float* pre,
                int ld_pre);
}

extern "C" int prv2_conv2d(const prv2_conv_desc* d, const float* x, const void* w_packed, const float* bias,
                           const float* ln_weight, const float* ln_bias, const float* gamma, const float* mul,
                           const float* res, const float* res2, float* y, void* stream) {
  return conv2d_impl(d, x, w_packed, bias, ln_weight, ln_bias, gamma, mul, res, res2, y, stream, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0);
}

// gate_w != null: the GatedConvUnit tail at 32 / 128 channels (prv2_conv3x3_ln_gate routes here; conv3x3_m16.hip GATE): mul / res then
// belong to the final stage y = mul * sigmoid(W_g act(LN(conv + bias)) + gate_bias) (+ res)
// fused-upsample loader (prv2.h): the layer's shape contract, checked on the descriptor alone
static bool ups_shape_ok(const prv2_conv_desc* d, const prv2_ups_src* u) {
  return d && u && u->x && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->convt_k == 0 && !d->same_pad && d->prec != PRV2_PREC_F32 &&
         !d->force_generic && d->part == 0 && d->cout > 64 && d->w >= 24 && d->h >= 4 && !d->relu_in && u->channels > 0 && u->channels % 32 == 0 &&
         u->channels <= d->cin && u->ld % 4 == 0 && u->ld >= u->channels && u->h >= 1 && u->w >= 1 && (long long)u->h * u->w * u->ld < (1LL << 29) &&
         (long long)d->h * d->w * d->ldx < (1LL << 29) && !(getenv("PRV2_HALO_MFMA32") && getenv("PRV2_HALO_MFMA32")[0] == '1');
}

extern "C" int prv2_conv2d_ups_supported(const prv2_conv_desc* d, const prv2_ups_src* u) { return ups_shape_ok(d, u) ? 1 : 0; }

extern "C" int prv2_conv2d_ups(const prv2_conv_desc* d, const float* x, const prv2_ups_src* u, const void* w_packed, const float* bias,
                               const float* ln_weight, const float* ln_bias, const float* res, float* y, void* stream) {
  PRV2_REQUIRE(ups_shape_ok(d, u), "conv2d_ups: layer / source not covered (3x3 s1 p1, bf16 modes, cout > 64, width >= 24, channels %% 32 == 0, no input ReLU)");
  PRV2_REQUIRE(aligned16(u->x) && (u->bstride % 4 == 0), "conv2d_ups: source must be 16-byte aligned");
  return conv2d_impl(d, x, w_packed, bias, ln_weight, ln_bias, nullptr, nullptr, res, nullptr, y, stream, nullptr, nullptr, u, nullptr, nullptr, 0, 0);
}

// depth-pair tail (prv2.h): the layer must end in the lean store loop of the 16x16x32 halo kernels
static bool tail_shape_ok(const prv2_conv_desc* d) {
  return d && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->convt_k == 0 && !d->same_pad && d->prec != PRV2_PREC_F32 &&
         !d->force_generic && d->part == 0 && d->w >= 24 && d->h >= 4 && d->cout % 4 == 0 && d->ldy >= d->cout + 4 && d->ldy % 4 == 0 &&
         d->cout <= 128 && (long long)d->h * d->w * d->ldx < (1LL << 29) && (long long)d->h * d->w * d->ldy < (1LL << 31) &&
         !(getenv("PRV2_HALO_MFMA32") && getenv("PRV2_HALO_MFMA32")[0] == '1');
}

extern "C" int prv2_conv2d_tail_supported(const prv2_conv_desc* d) { return tail_shape_ok(d) ? 1 : 0; }

extern "C" int prv2_conv2d_tail(const prv2_conv_desc* d, const float* x, const void* w_packed, const float* bias, const float* ln_weight,
                                const float* ln_bias, const float* res, const float* p1, const float* p2, int32_t ph, int32_t pw, float* y,
                                void* stream) {
  PRV2_REQUIRE(tail_shape_ok(d) && p1 && p2 && ph > 0 && pw > 0 && !(ln_weight && res),
               "conv2d_tail: layer not covered (3x3 s1 p1, bf16 modes, width >= 24, cout %% 4 == 0 and <= 128, ldy >= cout + 4, LayerNorm or residual but not both)");
  return conv2d_impl(d, x, w_packed, bias, ln_weight, ln_bias, nullptr, nullptr, res, nullptr, y, stream, nullptr, nullptr, nullptr, p1, p2, ph, pw);
}

// pre-stage addend (prv2.h prv2_conv2d_pre): the layer must run on the 16x16x32 halo kernels or on the 256-column kernel
static bool pre_shape_ok(const prv2_conv_desc* d) {
  return d && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->convt_k == 0 && !d->same_pad && d->prec != PRV2_PREC_F32 &&
         !d->force_generic && d->part == 0 && d->w >= 24 && d->h >= 4 && d->cout % 4 == 0 && d->fmt == 0 &&
         (long long)d->h * d->w * d->ldx < (1LL << 29) && (long long)d->h * d->w * d->ldy < (1LL << 29) &&
         !(getenv("PRV2_HALO_MFMA32") && getenv("PRV2_HALO_MFMA32")[0] == '1');
}

extern "C" int prv2_conv2d_pre_supported(const prv2_conv_desc* d) { return pre_shape_ok(d) ? 1 : 0; }

extern "C" int prv2_conv2d_pre(const prv2_conv_desc* d, const float* x, const void* w_packed, const float* bias, const float* pre, int32_t ld_pre,
                               const float* ln_weight, const float* ln_bias, const float* res, float* y, void* stream) {
  PRV2_REQUIRE(pre_shape_ok(d) && pre, "conv2d_pre: layer not covered (3x3 s1 p1, bf16 modes, width >= 24, height >= 4, cout %% 4 == 0)");
  PRV2_REQUIRE(ld_pre >= d->cout && ld_pre % 4 == 0 && aligned16(pre) && (long long)d->h * d->w * ld_pre < (1LL << 29), "conv2d_pre: addend layout");
  PRV2_REQUIRE(!ln_weight || d->cout <= 128 || d->cout == 256, "conv2d_pre: fused LayerNorm needs cout <= 128 or == 256 (got %d)", d->cout);
  return conv2d_impl(d, x, w_packed, bias, ln_weight, ln_bias, nullptr, nullptr, res, nullptr, y, stream, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0,
                     pre, ld_pre);
}

int prv2::conv2d_impl(const prv2_conv_desc* d, const float* x, const void* w_packed, const float* bias, const float* ln_weight,
                      const float* ln_bias, const float* gamma, const float* mul, const float* res, const float* res2, float* y, void* stream,
                      const void* gate_w, const float* gate_bias, const prv2_ups_src* ups, const float* tail1, const float* tail2, int tail_h,
                      int tail_w, const float* pre, int ld_pre) {
  PRV2_REQUIRE(d && x && w_packed && y, "conv2d: null pointer");
  PRV2_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, "conv2d: bad sizes");
  PRV2_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->pad >= 0, "conv2d: bad kernel geometry");
  PRV2_REQUIRE(d->ldx >= d->cin, "conv2d: ldx %d < cin %d", d->ldx, d->cin);
  PRV2_REQUIRE(d->prec >= PRV2_PREC_F32 && d->prec <= PRV2_PREC_BF16, "conv2d: unknown precision mode %d", d->prec);
  PRV2_REQUIRE(d->act >= PRV2_ACT_NONE && d->act <= PRV2_ACT_SILU, "conv2d: unknown activation %d", d->act);
  PRV2_REQUIRE(aligned16(w_packed), "conv2d: packed weights must be 16-byte aligned");
  PRV2_REQUIRE((ln_weight == nullptr) == (ln_bias == nullptr), "conv2d: ln_weight and ln_bias go together");
  // 3x3 convs with 256 output channels (a layer property: the choice never depends on the batch): the workgroup holds the whole
  // channel row, so the LayerNorm is fused for this width too
  if (!ups && !gate_w && !d->force_generic && !gamma && !mul && !res2 && d->part == 0 && conv3x3_c256_eligible(d, x, res, y) && (!pre || ln_weight))
    return prv2_conv3x3_ln_gate_pre(d, x, w_packed, bias, pre, ld_pre, ln_weight, ln_bias, nullptr, nullptr, nullptr, res, y, stream);
  PRV2_REQUIRE(d->fmt == 0, "conv2d: pre-split (X2) operands are only taken by the 256-column 3x3 kernels (fmt %d, %d->%d k%d)", d->fmt, d->cin, d->cout, d->kh);
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.w = w_packed; p.bias = bias; p.gamma = gamma; p.mul = mul; p.res = res; p.res2 = res2; p.y = y;
  p.ln_w = ln_weight; p.ln_b = ln_bias; p.ln_eps = d->ln_eps;
  p.gate_w = gate_w; p.gate_bias = gate_bias;
  p.pre = pre; p.ld_pre = ld_pre;
  if (tail1) {
    p.tail1 = tail1; p.tail2 = tail2; p.tH = tail_h; p.tW = tail_w;
    p.tsy = ac_scale(tail_h, d->h); p.tsx = ac_scale(tail_w, d->w);
  }
  if (ups) {
    p.xu = ups->x; p.uH = ups->h; p.uW = ups->w; p.ldxu = ups->ld; p.ups_c = ups->channels;
    p.xu_bstride = ups->bstride ? ups->bstride : (long long)ups->h * ups->w * ups->ld;
    p.usy = ac_scale(ups->h, d->h); p.usx = ac_scale(ups->w, d->w);
  }
  PRV2_REQUIRE(!ln_weight || (d->cout <= 128 && d->convt_k == 0), "conv2d: fused LayerNorm needs cout <= 128 (got %d)", d->cout);
  p.N = d->n; p.H = d->h; p.W = d->w;
  p.Cin = d->cin; p.Cin_pad = (int)roundup(d->cin, BK); p.Cout = d->cout;
  p.convt_k = d->convt_k;
  if (d->convt_k > 0) {
    PRV2_REQUIRE(d->kh == d->convt_k && d->kw == d->convt_k && d->stride == d->convt_k && d->pad == 0,
                 "conv2d: convt_k needs kernel == stride == k, pad 0");
    PRV2_REQUIRE(!d->same_pad, "conv2d: same_pad is not defined for convt_k");
    p.KH = p.KW = 1; p.stride = 1; p.pad = p.pad_x = 0;
    p.OH = d->h; p.OW = d->w;
    p.Ncols = d->convt_k * d->convt_k * d->cout;
  } else {
    p.KH = d->kh; p.KW = d->kw; p.stride = d->stride; p.pad = p.pad_x = d->pad;
    p.OH = (d->h + 2 * d->pad - d->kh) / d->stride + 1;
    p.OW = (d->w + 2 * d->pad - d->kw) / d->stride + 1;
    if (d->same_pad) {  // timm Conv2dSame / TensorFlow "SAME": the high side gets the odd pixel, handled by the range checks
      p.OH = (d->h + d->stride - 1) / d->stride;
      p.OW = (d->w + d->stride - 1) / d->stride;
      const int ty = (p.OH - 1) * d->stride + d->kh - d->h, tx = (p.OW - 1) * d->stride + d->kw - d->w;
      p.pad = (ty > 0 ? ty : 0) / 2;
      p.pad_x = (tx > 0 ? tx : 0) / 2;
    }
    p.Ncols = d->cout;
  }
  PRV2_REQUIRE(p.OH > 0 && p.OW > 0, "conv2d: empty output");
  PRV2_REQUIRE(d->ldy >= d->cout, "conv2d: ldy %d < cout %d", d->ldy, d->cout);
  p.ldx = d->ldx; p.ldy = d->ldy; p.ld_mul = d->ld_mul; p.ld_res = d->ld_res; p.ld_res2 = d->ld_res2;
  PRV2_REQUIRE(!mul || d->ld_mul >= d->cout, "conv2d: ld_mul");
  PRV2_REQUIRE(!res || d->ld_res >= d->cout, "conv2d: ld_res");
  PRV2_REQUIRE(!res2 || d->ld_res2 >= d->cout, "conv2d: ld_res2");
  p.x_bstride = d->x_bstride ? d->x_bstride : (long long)d->h * d->w * d->ldx;
  const int kk = d->convt_k > 0 ? d->convt_k : 1;
  p.y_bstride = d->y_bstride ? d->y_bstride : (long long)p.OH * kk * p.OW * kk * d->ldy;
  p.M = (long long)d->n * p.OH * p.OW;
  p.relu_in = d->relu_in; p.act = d->act;
  PRV2_REQUIRE((d->ldx % 4 == 0) && aligned16(x) && (p.x_bstride % 4 == 0) && d->ldx >= roundup(d->cin, 4),
               "conv2d: x must be 16-byte aligned with ldx %% 4 == 0 and ldx >= roundup(cin,4) (ldx=%d cin=%d)", d->ldx, d->cin);
  p.vec_ok = 1;
  {
    // 16-byte epilogue: cout a multiple of 4 -- or (the 98 / 194 / 322 / 642 / 770-channel layers) rows whose only
    // bytes behind cout are the pad channels up to the pixel stride: those are then written too, as zeros
    const int c4 = (int)roundup(d->cout, 4);
    const bool padded = d->cout % 4 != 0 && d->convt_k == 0;
    bool ve = (d->ldy % 4 == 0) && aligned16(y) && (p.y_bstride % 4 == 0) && (!padded || d->ldy == c4);
    if (mul) ve = ve && (d->ld_mul % 4 == 0) && aligned16(mul) && (!padded || d->ld_mul == c4);
    if (res) ve = ve && (d->ld_res % 4 == 0) && aligned16(res) && (!padded || d->ld_res == c4);
    if (res2) ve = ve && (d->ld_res2 % 4 == 0) && aligned16(res2) && (!padded || d->ld_res2 == c4);
    if (d->convt_k > 0) ve = ve && (d->cout % 4 == 0);
    p.vec_epi = ve;
  }
  p.tiles_m = (int)cdiv(p.M, BM);
  hipStream_t s = (hipStream_t)stream;
  if (conv3x3_halo_supported(p) && !d->force_generic) {
    if (has_tail_tile(d->cin, d->kh, d->kw, d->convt_k, d->prec))
      p.w_tail = (const float*)w_packed + roundup(p.Ncols, 128) * 9 * p.Cin_pad;
    // The fusion pyramid is 392x518, 196x259, 98x130, 49x65, 25x33: every width is 32k + {6, 3, 2, 1, 1}, i.e. the
    // last 32-pixel tile column of the halo kernel would be 81-97 % padding.  A remainder of <= 8 columns goes to
    // a second launch instead: the same kernel with 32-row x 8-pixel tiles (bf16 modes), or the generic kernel's
    // column window (f32 mode); disjoint outputs, same stream.
    const int rem = p.W % 32;
    const bool strip = rem != 0 && rem <= 8 && p.W >= 64;
    p.tiles_x = strip ? p.W / 32 : (int)cdiv(p.W, 32);
    if (gate_w) {
      PRV2_REQUIRE(conv3x3_halo16_gate_usable(p, d->prec) && d->part == 0, "conv3x3_ln_gate: layer not covered by the gate kernel (%d->%d, %dx%d, prec %d)",
                   d->cin, d->cout, d->h, d->w, d->prec);
      if (strip) {
        p.rx0 = p.W - rem;
        p.rw = rem;
      }
      launch_conv3x3_halo16_gate(p, d->prec, s);
      PRV2_LAUNCH_CHECK("conv3x3_ln_gate(halo16)");
      return 0;
    }
    PRV2_REQUIRE(!ups || conv3x3_halo16_ups_usable(p, d->prec), "conv2d_ups: layer not covered by the 128-column halo kernel (%d->%d, %dx%d, prec %d)",
                 d->cin, d->cout, d->h, d->w, d->prec);
    PRV2_REQUIRE(!tail1 || (conv3x3_halo16_usable(p, d->prec) && p.vec_epi), "conv2d_tail: layer not covered by the 16x16x32 halo kernels (%d->%d, %dx%d, prec %d)",
                 d->cin, d->cout, d->h, d->w, d->prec);
    PRV2_REQUIRE(!pre || conv3x3_halo16_usable(p, d->prec), "conv2d_pre: layer not covered by the 16x16x32 halo kernels (%d->%d, %dx%d, prec %d)",
                 d->cin, d->cout, d->h, d->w, d->prec);
    if (strip && conv3x3_halo16_usable(p, d->prec)) {  // bf16 modes: tiles and strip are one launch
      PRV2_REQUIRE(d->part == 0, "conv2d: part=%d is only meaningful when the strip is a launch of its own (f32 mode)", d->part);
      p.rx0 = p.W - rem;
      p.rw = rem;
      launch_conv3x3_halo(p, d->prec, s);
      PRV2_LAUNCH_CHECK("conv2d(3x3 halo + strip)");
      return 0;
    }
    PRV2_REQUIRE(d->part >= 0 && d->part <= 2 && (d->part == 0 || strip), "conv2d: part=%d but this conv has no remainder strip", d->part);
    if (d->part != 2) {
      launch_conv3x3_halo(p, d->prec, s);
      PRV2_LAUNCH_CHECK("conv2d(3x3 halo)");
    }
    if (!strip || d->part == 1) return 0;
    p.w_tail = nullptr;  // f32 mode: the strip goes through the generic kernel's column window
    p.rx0 = p.W - rem;
    p.rw = rem;
    p.M = (long long)d->n * p.OH * rem;
    p.tiles_m = (int)cdiv(p.M, BM);
  } else
  if (gate_w || ups || tail1 || pre) {
    PRV2_REQUIRE(false, "%s: layer not covered by the halo kernels (%d->%d k%d, %dx%d)", ups ? "conv2d_ups" : (tail1 ? "conv2d_tail" : (pre ? "conv2d_pre" : "conv3x3_ln_gate")), d->cin, d->cout, d->kh, d->h, d->w);
  } else
  if (conv_few_in_supported(p) && !d->force_generic && aligned16(x)) {
    const int ng = (p.Cout + 7) >> 3;
    hipLaunchKernelGGL(conv_few_in_kernel, dim3(flat_grid(p.M * ng, 256)), dim3(256), 0, s, p, (int)d->prec);
    set_kernel("conv_few_in_kernel", 64, d->prec);
    PRV2_LAUNCH_CHECK("conv2d(few input channels)");
    return 0;
  }
  if (conv1x1_small_supported(p) && !d->force_generic) {
    const int ng = (p.Cout + 7) >> 3;
    hipLaunchKernelGGL(conv1x1_small_kernel, dim3(flat_grid(p.M * ng, 256)), dim3(256), 0, s, p, (int)d->prec);
    set_kernel("conv1x1_small_kernel", 64, d->prec);
    PRV2_LAUNCH_CHECK("conv2d(1x1 small)");
    return 0;
  }
  if (gemm16_supported(p, d->prec) && !d->force_generic) {
    launch_gemm16(p, d->prec, s);
    PRV2_LAUNCH_CHECK("conv2d(1x1 gemm16)");
    return 0;
  }
#define PRV2_LAUNCH_IGEMM(BN_, PREC_) \
  hipLaunchKernelGGL((igemm_kernel<BN_, PREC_>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p)
  // 64-column tiles also for wider layers whose 128-column grid would leave half the CUs idle (the 13 x 17 level of the
  // fusion pyramid: 25 row tiles) -- a workgroup's time is its serial K loop; not with a fused LayerNorm (needs the whole row)
  const bool narrow = p.Ncols > 64 && !p.ln_w && (long long)p.tiles_m * cdiv(p.Ncols, 128) <= 128;
  if (p.Ncols > 64 && !narrow) {
    p.tiles_n = (int)cdiv(p.Ncols, 128);
    set_kernel("igemm_kernel", 128, d->prec);
    if (d->prec == PRV2_PREC_F32) PRV2_LAUNCH_IGEMM(128, PRV2_PREC_F32);
    else if (d->prec == PRV2_PREC_BF16X3) PRV2_LAUNCH_IGEMM(128, PRV2_PREC_BF16X3);
    else PRV2_LAUNCH_IGEMM(128, PRV2_PREC_BF16);
  } else {
    p.tiles_n = (int)cdiv(p.Ncols, 64);
    set_kernel("igemm_kernel", 64, d->prec);
    if (d->prec == PRV2_PREC_F32) PRV2_LAUNCH_IGEMM(64, PRV2_PREC_F32);
    else if (d->prec == PRV2_PREC_BF16X3) PRV2_LAUNCH_IGEMM(64, PRV2_PREC_BF16X3);
    else PRV2_LAUNCH_IGEMM(64, PRV2_PREC_BF16);
  }
#undef PRV2_LAUNCH_IGEMM
  PRV2_LAUNCH_CHECK("conv2d");
  return 0;
}
