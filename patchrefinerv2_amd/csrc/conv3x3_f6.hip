// The 3x3 convs of the 256-channel GatedConvUnits in the fp16 + block-scaled-fp6 arithmetic ("f16f6", DESIGN.md section 9 item 0;
// estimator/models/blocks/bi_directional_fusion_model.py:40-51, 56-82):
//   conv3x3_c256_f6_kernel       GatedConvUnit.conv + the skip:  y = conv3x3(relu(x)) + bias + res, C -> 256              (stage 1)
//   conv3x3_c256_gate_f6_kernel  the unit's tail: conv3x3 over its pre-split ``out`` (+ pre), LayerNorm, ReLU, gate, sigmoid x mul (+ res)  (stage 2)
//
//   x w  ~=  f16(x) f16(w)  +  q6(x) q6(w - f16 w)  +  q6(x - f16 x) q6(w)
//
// q6 = fp6 e2m3 with one power-of-two (E8M0) scale per 32 channels.  Per 64 input channels and tap an accumulator takes TWO
// v_mfma_f32_16x16x32_f16 and ONE v_mfma_scale_f32_16x16x128_f8f6f4 (its four 32-k blocks: the two corrections of the two 32-channel
// slabs) instead of the six v_mfma_f32_16x16x32_bf16 of the bf16x3 scheme; measured rms error of a dot product against float64: 1.2e-5
// (bf16x3: 4.4e-6, one fp16 product: 2.9e-4; profiles/r03_f16f6_study.txt).  Instruction semantics: tools/probes/f16f6_probe.hip.
//
// Not conv3x3_gate.hip's pipeline with other MFMAs (that mock ran 1.30x: its LDS traffic -- weights by LDS-DMA, every wave re-reading
// them -- becomes the bound once the MFMA time halves; profiles/r05_experiments.txt #12).  The main loop (f6_body) instead:
//   * wave w of 8 owns output channels 32 w .. 32 w + 31 for ALL 128 pixels of the 8 x 16 tile (8 pixel runs x 2 row blocks of 16 weights:
//     16 accumulators).  Weights are the MFMA's A operand and come STRAIGHT from L2 into registers, fragment-major (one coalesced KB per
//     load, every byte fetched once per workgroup), refilled part by part as the three passes of a step (f16 slab 0, f16 slab 1, fp6)
//     finish with them -- no LDS for weights, no per-tap barrier.  An accumulator lane then holds 8 consecutive channels of one pixel.
//   * LDS holds only the activation halo: 10 x 18 pixels x one 64-channel superslab, already in operand format
//     [32 f16 | 32 f16 | 4 x 16 B fp6 (first halves) | 4 x (8 B fp6, scale, pad)], pitch 288 B (conflict-free ds_read_b128), two buffers,
//     ONE barrier per superslab (9 taps).  The raw input halo (fp32, or the pre-split X2 bytes) arrives by LDS-DMA (buffer_load ... lds: zero
//     fill outside the image) into a per-wave staging area; the wave that moved a pixel converts it (ReLU, x x_scale, fp16 + residual, block
//     maxima, two v_cvt_scalef32_2xpk16_fp6_f32), so no cross-wave hand-off is needed for the staging.
//   * a workgroup walks a SEQUENCE of tiles: the next tile's first superslab is staged under the current tile's last, the weight stream wraps.
// Stage 1 is persistent (one workgroup per CU); its epilogue (x out_scale, + bias, + res, fp32 or X2 output) stages half tiles as 1 KB pixel rows in the LDS that
// is free at a tile's end and leaves through a row store loop; stage 2 runs one tile per workgroup and hands its accumulators to conv3x3_gate.hip's epilogue through the C tile in LDS
// (conv3x3_gate_epi.h).
// Range: fp16 holds |x x_scale| <= 65504; larger values are clamped in the fp16 part and fall to the fp6 residual (finite, imprecise) --
// the caller keeps x_scale / the weights' scale (powers of two, undone by out_scale) such that this does not happen, and reads the
// observed maximum back through ``range_word`` (ops.F6Range).
// Timing switches for tools/probes/f6_ablate.sh / f6_bisect.sh / f6_stamps.sh (wrong results by design): F6_ABL_*, F6_DBG_*, F6_STAMPS; F6_D,
// F6_CV0 / F6_CV1: ring depth and conversion taps.
#include <cstdlib>
#include <type_traits>

#include "igemm.h"
#include "conv3x3_gate_epi.h"

namespace prv2 {

namespace f6 {
constexpr int TH = 8, TW = 16, HW_ = TW + 2, HALO = (TH + 2) * HW_;  // 180 halo pixels
constexpr int PIX = 288;                                              // bytes per halo pixel and superslab in LDS
constexpr int A_BYTES = HALO * PIX;                                   // 51,840
constexpr int ST_CHUNK = 1040, ST_DMAS = 6, ST_WAVE = ST_DMAS * ST_CHUNK;  // staging: 6 DMAs of 4 pixels x 256 B per wave (+16 B bank shift)
constexpr int ST_BASE = 2 * A_BYTES;
constexpr int LDS_BYTES = ST_BASE + 8 * ST_WAVE;                      // 153,600
constexpr int STEP_BYTES = 8 * 2 * 4 * 1024;                          // weights of one (superslab, tap): 8 waves x 2 row blocks x 4 parts x 1 KB
#ifndef F6_D
#define F6_D 4
#endif
constexpr int D = F6_D;                                               // activation fragments in flight (ring)
// the next superslab's halo is converted by waves 0-3 at tap CV_TAP0 and by waves 4-7 (their SIMD partners) at tap CV_TAP1: while one wave
// of a SIMD runs its ~350 conversion instructions the other keeps the matrix pipe busy (both at the same tap: the pipe idles for both)
#ifndef F6_CV0
#define F6_CV0 2
#endif
#ifndef F6_CV1
#define F6_CV1 5
#endif
constexpr int CV_TAP0 = F6_CV0, CV_TAP1 = F6_CV1;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(8 * ST_DMAS * 4 >= HALO, "staging covers the halo");
}  // namespace f6

struct F6Params {
  const float* x;
  const char* w;      // prv2_pack_conv3x3_f6_weight
  const float* bias;  // or null
  const float* res;   // or null
  float* y;
  unsigned* range;    // or null: atomicMax of the bits of max |relu(x) x_scale| seen
  int N, H, W, Cin, ldx, ldy, ld_res;
  long long x_bstride, y_bstride;
  float x_scale, out_scale;
  int relu_in, y_x2;
  long long* stamps;  // -DF6_STAMPS builds (tools/probes/f6_stamps.sh): per workgroup and wave [main-loop cycles, epilogue cycles, tiles, prologue cycles, total]
};

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// 32 floats (a: elements 0..15, b: 16..31) / scale -> 32 x e2m3, RNE; element i of a at 6-bit position 2 i, of b at 2 i + 1 (the same
// instruction packs the weights: positions agree by construction).  Early-clobber destination: hipcc 7.2 may allocate the builtin's
// destination over its scale operand (tools/probes/f16f6_probe.hip).
__device__ __forceinline__ u32x6 cvt_fp6(const f32x16& a, const f32x16& b, float scale) {
  u32x6 q;
  asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(q) : "v"(a), "v"(b), "v"(scale));
  return q;
}
// E8M0 scale of a block with maximum magnitude m (MX rule: exponent(m) - 2, e2m3's largest exponent); never 0 (an all-zero block
// divides by 2^-126) and never 255
__device__ __forceinline__ int e8m0_of(float m) {
  const int eb = (int)((__float_as_uint(m) >> 23) & 0xffu);
  return min(max(eb - 2, 1), 254);
}

// a wave-uniform 64-bit value into an SGPR pair (readfirstlane returns int: widen its halves as UNSIGNED -- a sign-extended low half
// whose bit 31 is set turns the address into 0xffffffff........)
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  return (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v) |
         ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32);
}

#ifndef F6_DBG_NOW
#define F6_LDW(dst, voff, sbase, imm) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #imm : "=v"(dst) : "v"(voff), "s"(sbase) : "memory")
#else  // (fault bisection: the same loads from the first kilobytes of the image)
#define F6_LDW(dst, voff, sbase, imm) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #imm : "=v"(dst) : "v"(lane * 16), "s"(p.w) : "memory")
#endif

// The main loop over a workgroup's tile sequence t_begin, t_begin + per, ... (K tiles): ``epi(n_img, y0, x0, acc)`` takes a finished tile's
// accumulators (lane = pixel m16 of run a, channels 32 wave + 8 (lane >> 4) .. + 7: acc[a][0] the first four, acc[a][1] the rest; UNSCALED: x
// out_scale is the epilogue's) and leaves them cleared.  X2IN: x arrives pre-split (conv3x3_gate.hip: per 8 channels [8 bf16 hi | 8 bf16 lo], the
// bytes of 8 floats): the staged bytes are decoded as hi + lo.
template <bool X2IN, class Epi>
__device__ __forceinline__ void f6_body(const F6Params& p, char* const smem, const int t_begin, const int per, const int K, Epi&& epi) {
  using namespace f6;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15, g = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)smem;
  const int tiles_x = (p.W + TW - 1) / TW, tiles_xy = tiles_x * ((p.H + TH - 1) / TH);
  auto tile_of = [&](int k, int& n, int& y0, int& x0) {
    const int t = t_begin + k * per;
    n = t / tiles_xy;
    const int r = t - n * tiles_xy, ty = r / tiles_x;
    y0 = ty * TH;
    x0 = (r - ty * tiles_x) * TW;
  };

  // ---- halo DMA: instruction i of this wave moves halo pixels 4 (6 wave + i) .. + 3 (lane >> 4), 16 bytes per lane ------------
  constexpr unsigned OOB = 0x80000000u;
  i32x4 rsrc;
  rsrc.z = __builtin_amdgcn_readfirstlane((int)(unsigned)((((long long)p.H * p.W - 1) * p.ldx + p.Cin) * 4));
  rsrc.w = 0x00020000;
  unsigned hoff[ST_DMAS];
  auto set_loader = [&](int k) {  // image base and halo offsets of tile k
    int n, y0, x0;
    tile_of(k, n, y0, x0);
    const unsigned long long img_base = (unsigned long long)(size_t)(p.x + (long long)n * p.x_bstride);
    rsrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)img_base);
    rsrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((img_base >> 32) & 0xffffu));
#pragma unroll
    for (int i = 0; i < ST_DMAS; ++i) {
      const int hp = (wave * ST_DMAS + i) * 4 + g;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = hp < HALO && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      hoff[i] = ok ? (unsigned)(((iy * p.W + ix) * p.ldx + m16 * 4) * 4) : OOB;
    }
  };
  set_loader(0);
  const int nss = p.Cin >> 6, nsteps = nss * 9;
#ifdef F6_DBG_NODMA
  i32x4 dbg_sink;
#endif
  const unsigned st_wave = lds0 + ST_BASE + wave * ST_WAVE;
  auto dma_halo = [&](int ss) {
    const unsigned cofs = (unsigned)(ss * 256);
#pragma unroll
    for (int i = 0; i < ST_DMAS; ++i) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(st_wave + i * ST_CHUNK);
      const unsigned voff = hoff[i] + cofs;
#ifndef F6_DBG_NODMA
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(voff), "s"(rsrc) : "memory");
#else
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dbg_sink) : "v"(voff), "s"(rsrc) : "memory");
#endif
    }
  };
  // ---- conversion item of a lane: staged pixel lp = lane >> 1 of this wave (halo pixel 24 wave + lp), 32-channel slab lane & 1 ---------
  const int lp = lane >> 1, cs = lane & 1, chp = wave * 24 + lp;
  const char* const st_rd = smem + ST_BASE + wave * ST_WAVE + (lp >> 2) * ST_CHUNK + (lp & 3) * 256 + cs * 128;
  const float xs = p.x_scale;
  const float relu_floor = p.relu_in ? 0.f : -__builtin_inff();
  float seen = 0.f;
  auto convert = [&](int wbuf) {
    if (lane < 48) {
      f32x16 a, b;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f32x4 ta, tb;
        if constexpr (X2IN) {  // chunks 2q / 2q + 1 of the 128 bytes = bf16 hi / lo of channels 8q .. 8q + 7: k = 2q + h takes elements 4h .. 4h + 3
          typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
          const int q = k >> 1, h = k & 1;
          const u32x2v ah = *reinterpret_cast<const u32x2v*>(st_rd + 32 * q + 8 * h), al = *reinterpret_cast<const u32x2v*>(st_rd + 32 * q + 16 + 8 * h);
          const u32x2v bh = *reinterpret_cast<const u32x2v*>(st_rd + 64 + 32 * q + 8 * h), bl = *reinterpret_cast<const u32x2v*>(st_rd + 64 + 32 * q + 16 + 8 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned sh = (e & 1) ? 0u : 16u;  // (element 2i in the low half of its dword)
            ta[e] = __uint_as_float((ah[e >> 1] << sh) & 0xffff0000u) + __uint_as_float((al[e >> 1] << sh) & 0xffff0000u);
            tb[e] = __uint_as_float((bh[e >> 1] << sh) & 0xffff0000u) + __uint_as_float((bl[e >> 1] << sh) & 0xffff0000u);
          }
        } else {
          ta = *reinterpret_cast<const f32x4*>(st_rd + 16 * k);
          tb = *reinterpret_cast<const f32x4*>(st_rd + 64 + 16 * k);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[4 * k + e] = fmaxf(ta[e], relu_floor) * xs;
          b[4 * k + e] = fmaxf(tb[e], relu_floor) * xs;
        }
      }
      float mx = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, fmaxf(fabsf(a[i]), fabsf(b[i])));
      seen = fmaxf(seen, mx);
      const int ex = e8m0_of(mx);
      const u32x6 qx = cvt_fp6(a, b, __uint_as_float((unsigned)ex << 23));
      f16x8 h[4];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const _Float16 ha = (_Float16)__builtin_amdgcn_fmed3f(a[i], -65504.f, 65504.f), hb = (_Float16)__builtin_amdgcn_fmed3f(b[i], -65504.f, 65504.f);
        h[i >> 3][i & 7] = ha;
        h[2 + (i >> 3)][i & 7] = hb;
        a[i] -= (float)ha;
        b[i] -= (float)hb;
      }
      float mr = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) mr = fmaxf(mr, fmaxf(fabsf(a[i]), fabsf(b[i])));
      const int er = e8m0_of(mr);
      const u32x6 qr = cvt_fp6(a, b, __uint_as_float((unsigned)er << 23));
      if (chp < HALO) {
        char* const dst = smem + wbuf * A_BYTES + chp * PIX;
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<f16x8*>(dst + cs * 64 + 16 * k) = h[k];
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u32x4*>(dst + 128 + 32 * cs) = u32x4{qx[0], qx[1], qx[2], qx[3]};
        *reinterpret_cast<u32x4*>(dst + 192 + 32 * cs) = u32x4{qx[4], qx[5], (unsigned)ex, 0u};
        *reinterpret_cast<u32x4*>(dst + 128 + 32 * cs + 16) = u32x4{qr[0], qr[1], qr[2], qr[3]};
        *reinterpret_cast<u32x4*>(dst + 192 + 32 * cs + 16) = u32x4{qr[4], qr[5], (unsigned)er, 0u};
      }
    }
  };

  // ---- weights: fragment-major [step][wave][row block j][part][lane] x 16 B; lane (row m16, k group g) -------------------------
  const unsigned wv0 = (unsigned)(wave * 8192 + lane * 16), wv1 = wv0 + 4096;
  i32x4 wf0[2], wf1[2], wql[2], wqh[2];
  auto w_of = [&](int s) { return (unsigned long long)(size_t)(p.w + (long long)(s == nsteps ? 0 : s) * STEP_BYTES); };  // (behind a tile's last step: the next tile's first)
#define F6_LOAD_F0(sb) do { F6_LDW(wf0[0], wv0, sb, 0); F6_LDW(wf0[1], wv1, sb, 0); } while (0)
#define F6_LOAD_F1(sb) do { F6_LDW(wf1[0], wv0, sb, 1024); F6_LDW(wf1[1], wv1, sb, 1024); } while (0)
#define F6_LOAD_Q(sb) do { F6_LDW(wql[0], wv0, sb, 2048); F6_LDW(wqh[0], wv0, sb, 3072); F6_LDW(wql[1], wv1, sb, 2048); F6_LDW(wqh[1], wv1, sb, 3072); } while (0)

  f32x4 acc[8][2];
#pragma unroll
  for (int a = 0; a < 8; ++a) acc[a][0] = acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: halo of superslab 0 -> buffer 0; weights of step 0 ------------------------------------------------------------
#ifdef F6_STAMPS
  const long long st_t0 = __builtin_readcyclecounter(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  dma_halo(0);
  {
    const unsigned long long sb = uniform64(w_of(0));
    F6_LOAD_F0(sb);
    F6_LOAD_F1(sb);
    F6_LOAD_Q(sb);
  }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the six DMAs (the weights stay in flight)
  convert(0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef F6_STAMPS
  const long long st_pro = __builtin_readcyclecounter() - st_t0;
#endif

  // activation fragments: lane (pixel m16 of the run, k group g); ring of D
  const unsigned xlane = (unsigned)(m16 * PIX + g * 16);
  i32x4 rlo[D], rhi[D];

#ifdef F6_STAMPS
  long long st_main = 0, st_epi = 0;
#endif
  for (int k = 0; k < K; ++k) {
#ifdef F6_STAMPS
  const long long st_a = __builtin_readcyclecounter();
#endif
  for (int ss = 0; ss < nss; ++ss) {
    const bool last_ss = ss + 1 == nss, more = !last_ss || k + 1 < K;  // (more: another superslab follows -- of this tile or of the next)
    const char* const xb = smem + ((k * nss + ss) & 1) * A_BYTES + xlane;
    auto read_item = [&](int it, int tap) {  // item = (pass it >> 3, run it & 7) of tap `tap`
      const int pp = it >> 3, a = it & 7, ky = tap / 3, kx = tap - 3 * ky;
      const char* q = xb + ((a + ky) * HW_ + kx) * PIX + (pp == 0 ? 0 : pp == 1 ? 64 : 128);
#ifdef F6_ABL_NOREAD  // (timing ablation: the fragment ring is filled once)
      if (k > 0 || ss > 0 || tap > 0 || it >= D) return;
#endif
      rlo[it % D] = *reinterpret_cast<const i32x4*>(q);
      if (pp == 2) rhi[it % D] = *reinterpret_cast<const i32x4*>(q + 64);
    };
    auto step = [&](auto tap_c) {
      constexpr int tap = decltype(tap_c)::value;
      constexpr int DM = tap == 0 ? ST_DMAS : 0;                  // DMAs issued in this step (behind its first weight refill)
      constexpr int DP = tap == 1 ? ST_DMAS : 0;                  // ... in the previous step
      const int s = ss * 9 + tap;
      const unsigned long long sb = uniform64(w_of(s + 1));
      if constexpr (tap == 0) {
#pragma unroll
        for (int it = 0; it < D - 1; ++it) read_item(it, 0);
      }
#pragma unroll
      for (int it = 0; it < 24; ++it) {
        const int pp = it >> 3, a = it & 7;
        // VMEM operations in issue order: ... f0' [DMAs] f1' q' | f0'' [DMAs] f1'' q'' ...: the counts of younger ones at each wait
        if (it == 0) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(wf0[0]), "+v"(wf0[1]), "+v"(wf1[0]), "+v"(wf1[1]) : "n"(6 + DP) : "memory");
        if (it == 8) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wf1[0]), "+v"(wf1[1]) : "n"(6 + DM) : "memory");
        if (it == 16) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(wql[0]), "+v"(wqh[0]), "+v"(wql[1]), "+v"(wqh[1]) : "n"(4 + DM) : "memory");
        {
          const int nt = it + D - 1;
          if (nt < 24) read_item(nt, tap);
          else if (tap < 8) read_item(nt - 24, tap + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (pp == 0) {
          const f16x8 xf = __builtin_bit_cast(f16x8, rlo[it % D]);
          acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf0[0]), xf, acc[a][0], 0, 0, 0);
          acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf0[1]), xf, acc[a][1], 0, 0, 0);
        } else if (pp == 1) {
          const f16x8 xf = __builtin_bit_cast(f16x8, rlo[it % D]);
          acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf1[0]), xf, acc[a][0], 0, 0, 0);
          acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf1[1]), xf, acc[a][1], 0, 0, 0);
        } else {
          const i32x4 xl = rlo[it % D], xh = rhi[it % D];
          const i32x8 xq = {xl.x, xl.y, xl.z, xl.w, xh.x, xh.y, 0, 0};
          const i32x8 q0 = {wql[0].x, wql[0].y, wql[0].z, wql[0].w, wqh[0].x, wqh[0].y, 0, 0};
          const i32x8 q1 = {wql[1].x, wql[1].y, wql[1].z, wql[1].w, wqh[1].x, wqh[1].y, 0, 0};
          acc[a][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(q0, xq, acc[a][0], 2, 2, 0, wqh[0].z, 0, xh.z);
          acc[a][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(q1, xq, acc[a][1], 2, 2, 0, wqh[1].z, 0, xh.z);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (it == 7) {
          F6_LOAD_F0(sb);
          if constexpr (tap == 0) {  // the next superslab's halo: of this tile, or superslab 0 of the next tile (behind the last: a repeat nobody converts)
            if (last_ss && k + 1 < K) set_loader(k + 1);
            dma_halo(last_ss ? (k + 1 < K ? 0 : ss) : ss + 1);
          }
#ifndef F6_ABL_NOCV
          if constexpr (tap == CV_TAP0) {
            if (more && wave < 4) convert((k * nss + ss + 1) & 1);
          }
          if constexpr (tap == CV_TAP1) {
            if (more && wave >= 4) convert((k * nss + ss + 1) & 1);
          }
#endif
        }
        if (it == 15) {
          F6_LOAD_F1(sb);
        }
        if (it == 23) F6_LOAD_Q(sb);
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
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // next halo converted by everyone, this one read by everyone
  }
  {
#ifdef F6_STAMPS
    const long long st_b = __builtin_readcyclecounter();
#endif
    int n_img, y0, x0;
    tile_of(k, n_img, y0, x0);
    epi(n_img, y0, x0, acc, ((k * nss + nss - 1) & 1) * A_BYTES);  // (+ the halo buffer the tile's last superslab was read from: free now)
#ifdef F6_STAMPS
    const long long st_c = __builtin_readcyclecounter();
    st_main += st_b - st_a; st_epi += st_c - st_b;
#endif
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(wf0[0]), "+v"(wf0[1]), "+v"(wf1[0]), "+v"(wf1[1]), "+v"(wql[0]), "+v"(wqh[0]), "+v"(wql[1]), "+v"(wqh[1])::"memory");
  if (p.range) {  // one atomic per wave, and only while it would raise the word
    float m = seen;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0 && __float_as_uint(m) > __hip_atomic_load(p.range, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p.range, __float_as_uint(m));
  }
#ifdef F6_STAMPS
  if (p.stamps && lane == 0) {
    long long* o = p.stamps + ((long long)blockIdx.x * 8 + wave) * 8;
    o[0] = st_main; o[1] = st_epi; o[2] = K; o[3] = st_pro; o[4] = __builtin_readcyclecounter() - st_t0; o[5] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
#endif
}

// ---- GatedConvUnit.conv: persistent workgroups, epilogue from registers ------------------------------------------------------------------------------
// An XCD takes a contiguous range of tiles, its workgroups walk it side by side (neighbouring tiles share halo rows in that XCD's L2; the weights are
// L2-resident per XCD anyway).  Epilogue of a tile: x out_scale, + bias, + res, fp32 or X2 -- 16-byte loads / stores of the lane's 8 channels.  It is
// bound by the CU's 64 B/clk of vector memory (128 KB in, 128 KB out per tile); requesting the residual rows under the main loop was measured: no
// gain, 9-27 registers spilled (profiles/r05_experiments.txt #15).
__global__ void __launch_bounds__(512, 2) conv3x3_c256_f6_kernel(const F6Params p) {
  using namespace f6;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int tiles_xy = ((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH), ntiles = p.N * tiles_xy;
  const int per = max((int)gridDim.x >> 3, 1), xcd = blockIdx.x & 7, jwg = blockIdx.x >> 3, chunk = (ntiles + 7) / 8;
  const int t_begin = xcd * chunk + jwg, t_end = min((xcd + 1) * chunk, ntiles);
  const int K = t_begin < t_end ? (t_end - t_begin + per - 1) / per : 0;
  if (K == 0) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), m16 = lane & 15, g = lane >> 4;
  const int c0 = wave * 32 + g * 8;
  const float os = p.out_scale;
#ifndef F6_EPI_DIRECT
  // Epilogue through LDS, two half tiles (4 runs = 64 pixel rows of 1 KB) at a time: a lane's accumulators are 32 bytes of 16 different pixels per
  // instruction -- as direct loads / stores that is 16 half-used cache lines per wave instruction and cost 16 k cycles per tile; from a row image every
  // instruction of the store loop takes two whole 1 KB pixel rows (thread = row tid >> 5 (+ 16 i), channels 8 (tid & 31) .. + 7).  Free LDS at this
  // point: the staging area (48 rows of 1040 B) and the halo buffer the tile's last superslab was read from (rows 48 .. 63); the OTHER halo buffer
  // already holds the next tile's first superslab.
  constexpr int RP = 1040;
  const int tid = threadIdx.x, srow = tid >> 5, c8 = (tid & 31) * 8;
  f6_body<false>(p, smem, t_begin, per, K, [&](int n_img, int y0, int x0, f32x4 (&acc)[8][2], const int free_buf) {
    auto row_ptr = [&](int r) { return smem + (r < 48 ? ST_BASE + r * RP : free_buf + (r - 48) * RP); };
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (p.bias) {
      b0 = *reinterpret_cast<const f32x4*>(p.bias + c0);
      b1 = *reinterpret_cast<const f32x4*>(p.bias + c0 + 4);
    }
    const long long img_px = (long long)n_img * p.H * p.W;
    const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res + img_px * p.ld_res : p.x), 0,
                                                                            p.res ? (int)(((unsigned)(p.H * p.W - 1) * p.ld_res + 256) * 4u) : 0, 0x00020000);
    float* const ybase = p.y + (long long)n_img * p.y_bstride + c8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // residual rows of this half, requested before the accumulators go to LDS: row r = srow + 16 i of the half = tile row 4 h + i, column srow
      f32x4 r0[4], r1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = y0 + 4 * h + i, ixs = x0 + srow;
        const unsigned off = (iy < p.H && ixs < p.W) ? (unsigned)(((iy * p.W + ixs) * p.ld_res + c8) * 4) : 0x80000000u;
        r0[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, off, 0, 0));
        r1[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, off + 16, 0, 0));
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        char* q = row_ptr(a * TW + m16) + c0 * 4;
        *reinterpret_cast<f32x4*>(q) = acc[4 * h + a][0] * os + b0;
        *reinterpret_cast<f32x4*>(q + 16) = acc[4 * h + a][1] * os + b1;
        acc[4 * h + a][0] = acc[4 * h + a][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = y0 + 4 * h + i, ixs = x0 + srow;
        const char* q = row_ptr(srow + 16 * i) + c8 * 4;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(q) + r0[i], v1 = *reinterpret_cast<const f32x4*>(q + 16) + r1[i];
#ifndef F6_DBG_NOEPI
        if (iy < p.H && ixs < p.W) {
          float* const dst = ybase + (long long)(iy * p.W + ixs) * p.ldy;
          if (p.y_x2) {  // pre-split output (conv3x3_gate.hip, head of the file): per 8 channels [8 bf16 hi | 8 bf16 lo]
            bf16x4 h0, l0, h1, l1;
            split_bf16(v0, h0, l0);
            split_bf16(v1, h1, l1);
            const bf16x8 hv = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7), lv = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16\n\ts_nop 1" ::"v"(dst), "v"(hv), "v"(lv) : "memory");
          } else {
            asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16\n\ts_nop 1" ::"v"(dst), "v"(v0), "v"(v1) : "memory");
          }
        }
#endif
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the rows are read out: the next half / the next tile's staging and conversion may take the LDS
    }
  });
#else
  f6_body<false>(p, smem, t_begin, per, K, [&](int n_img, int y0, int x0, f32x4 (&acc)[8][2], int) {
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (p.bias) {
      b0 = *reinterpret_cast<const f32x4*>(p.bias + c0);
      b1 = *reinterpret_cast<const f32x4*>(p.bias + c0 + 4);
    }
    // per image resource: rows below the image lie behind num_records (zero, no fault); a column right of it reads a pixel nobody stores
    const unsigned long long rb = (unsigned long long)(size_t)(p.res ? p.res + (long long)n_img * p.H * p.W * p.ld_res : p.x);
    i32x4 rr;
    rr.x = __builtin_amdgcn_readfirstlane((int)(unsigned)rb);
    rr.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((rb >> 32) & 0xffffu));
    rr.z = __builtin_amdgcn_readfirstlane(p.res ? (int)(((unsigned)(p.H * p.W - 1) * p.ld_res + 256) * 4u) : 0);
    rr.w = 0x00020000;
    const int ix = x0 + m16;
    const unsigned voff = (unsigned)(((y0 * p.W + ix) * p.ld_res + c0) * 4);
    const int row = p.W * p.ld_res * 4;
    f32x4 r0[8], r1[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const int so = __builtin_amdgcn_readfirstlane(a * row);
      asm volatile("buffer_load_dwordx4 %0, %2, %3, %4 offen\n\tbuffer_load_dwordx4 %1, %2, %3, %4 offen offset:16"
                   : "=&v"(r0[a]), "=&v"(r1[a]) : "v"(voff), "s"(rr), "s"(so) : "memory");
    }
    float* const ybase = p.y + (long long)n_img * p.y_bstride + c0;
    // first every run's arithmetic (the counted waits see loads only), then every store
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0[a]), "+v"(r1[a]) : "n"(2 * (7 - a)) : "memory");  // younger: the later runs' rows
      acc[a][0] = acc[a][0] * os + b0 + r0[a];
      acc[a][1] = acc[a][1] * os + b1 + r1[a];
    }
#pragma unroll
    for (int a = 0; a < 8; ++a) {
#ifndef F6_DBG_NOEPI
      if (y0 + a < p.H && ix < p.W) {
        const f32x4 v0 = acc[a][0], v1 = acc[a][1];
        float* const dst = ybase + (long long)((y0 + a) * p.W + ix) * p.ldy;
        // (s_nop: the hazard recognizer does not look into inline asm -- a VALU write of a wide store's data registers needs wait states behind it)
        if (p.y_x2) {  // pre-split output (conv3x3_gate.hip, head of the file): per 8 channels [8 bf16 hi | 8 bf16 lo]
          bf16x4 h0, l0, h1, l1;
          split_bf16(v0, h0, l0);
          split_bf16(v1, h1, l1);
          const bf16x8 hv = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7), lv = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
          asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16\n\ts_nop 1" ::"v"(dst), "v"(hv), "v"(lv) : "memory");
        } else {
          asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16\n\ts_nop 1" ::"v"(dst), "v"(v0), "v"(v1) : "memory");
        }
      }
#endif
      acc[a][0] = acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  });
#endif
}

// ---- GatedConvUnit tail (round 5, stage 2): the fp16 + fp6 main loop over the unit's pre-split ``out`` in front of conv3x3_gate.hip's epilogue -- C tile
// -> LDS, + pre (the conv's coarse half), LayerNorm, ReLU, the 256 x 256 gate GEMM (bf16x3: its A operand is formed in LDS, its weights are
// fragment-major in L2), sigmoid, x mul, + res.  One tile per workgroup like conv3x3_c256_gate_x2_kernel: the C tile (133 KB) takes the place of the
// halo buffers and the staging area, so nothing of a next tile can be staged under it.
template <bool X2IN>  // x / mul pre-split (the unit's X2 ``out`` or [out | coarse ROI] concat), or fp32 (the concat as the resizing ROI gather writes it)
__global__ void __launch_bounds__(512, 2) conv3x3_c256_gate_f6_kernel(const GateConvParams gp) {
  using namespace f6;
  static_assert(LDS_BYTES >= g256::EPI_FLOATS * 4, "the C tile fits the main loop's LDS");
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const IgemmParams& c = gp.c;
  F6Params p;
  p.x = c.x; p.w = reinterpret_cast<const char*>(c.w); p.bias = c.bias; p.res = nullptr; p.y = nullptr; p.range = gp.f6_range;
  p.N = c.N; p.H = c.H; p.W = c.W; p.Cin = c.Cin; p.ldx = c.ldx; p.ldy = c.ldy; p.ld_res = 0;
  p.x_bstride = c.x_bstride; p.y_bstride = c.y_bstride; p.x_scale = gp.f6_x_scale; p.out_scale = gp.f6_out_scale; p.relu_in = c.relu_in; p.y_x2 = 0;
  p.stamps = nullptr;
  int t = blockIdx.x;  // XCD-aware block -> tile, as conv3x3_gate.hip
  {
    const int ntiles = gridDim.x, q = ntiles >> 3, r = ntiles & 7, xcd = t & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), m16 = lane & 15, g = lane >> 4;
  const int c0 = wave * 32 + g * 8;
  const float os = p.out_scale;
  PRV2_STAMP(0);
  PRV2_STAMP(1);
  PRV2_CLK_STAMP(0);
  f6_body<X2IN>(p, smem, t, 1, 1, [&](int n_img, int y0, int x0, f32x4 (&acc)[8][2], int) {
    // the wrapped-around weight loads and the repeated halo DMA (nobody uses either) must have landed before the C tile takes the staging area
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    PRV2_STAMP(2);
    c256_epilogue<PRV2_PREC_BF16X3, true, X2IN>(gp, reinterpret_cast<float*>(smem), n_img, y0, x0, [&](float* ct) {
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
      if (c.bias) {
        b0 = *reinterpret_cast<const f32x4*>(c.bias + c0);
        b1 = *reinterpret_cast<const f32x4*>(c.bias + c0 + 4);
      }
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        float* q = ct + (a * TW + m16) * g256::CLD + c0;
        *reinterpret_cast<f32x4*>(q) = acc[a][0] * os + b0;
        *reinterpret_cast<f32x4*>(q + 4) = acc[a][1] * os + b1;
      }
    });
  });
}

// PyTorch [256][cin][3][3] fp32 weights x w_scale -> the fragment-major image the kernel streams: thread = (step, wave, row block j, lane):
// MFMA row r = lane & 15 is output channel 32 wave + 8 (r >> 2) + 4 j + (r & 3) (so that an accumulator lane holds 8 consecutive
// channels), k group g = lane >> 4: part 0 / 1 = f16 of channels 64 ss + 32 part + 8 g .. + 7; parts 2, 3 = fp6 block g of the
// K = 128 instruction = slab g >> 1, (g & 1) == 0: q6(w - f16 w) [meets q6(x)], == 1: q6(w) [meets q6(x - f16 x)], then its E8M0 scale
__global__ void __launch_bounds__(256) f6_pack_weight_kernel(const float* __restrict__ w, float w_scale, char* __restrict__ dst, int cin, int nsteps) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = idx & 63, j = (idx >> 6) & 1, wv = (idx >> 7) & 7, s = idx >> 10;
  if (s >= nsteps) return;
  const int ss = s / 9, tap = s - 9 * ss;
  const int r = lane & 15, g = lane >> 4;
  const int co = 32 * wv + 8 * (r >> 2) + 4 * j + (r & 3);
  const float* wr = w + (long long)co * cin * 9 + tap;
  char* o = dst + ((((long long)s * 8 + wv) * 2 + j) * 4) * 1024 + lane * 16;
#pragma unroll
  for (int part = 0; part < 2; ++part) {
    f16x8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (_Float16)(wr[(long long)(64 * ss + 32 * part + 8 * g + i) * 9] * w_scale);
    *reinterpret_cast<f16x8*>(o + part * 1024) = h;
  }
  f32x16 a, b;
  const int sl = g >> 1;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const float v = wr[(long long)(64 * ss + 32 * sl + i) * 9] * w_scale;
    const float q = (g & 1) ? v : v - (float)(_Float16)v;
    if (i < 16) a[i] = q;
    else b[i - 16] = q;
  }
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) m = fmaxf(m, fmaxf(fabsf(a[i]), fabsf(b[i])));
  const int e = e8m0_of(m);
  const u32x6 q = cvt_fp6(a, b, __uint_as_float((unsigned)e << 23));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  *reinterpret_cast<u32x4*>(o + 2 * 1024) = u32x4{q[0], q[1], q[2], q[3]};
  *reinterpret_cast<u32x4*>(o + 3 * 1024) = u32x4{q[4], q[5], (unsigned)e, 0u};
}

static inline bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

static bool f6_shape_ok(const prv2_conv_desc* d) {
  return d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->convt_k == 0 && !d->same_pad && d->cout == 256 && d->cin >= 64 &&
         d->cin % 64 == 0 && d->w >= f6::TW && (long long)d->h * d->w * d->ldx < (1LL << 29) && (long long)d->h * d->w * d->ldy < (1LL << 29);
}

}  // namespace prv2

using namespace prv2;

extern "C" int prv2_conv3x3_f6_supported(const prv2_conv_desc* d) { return d && f6_shape_ok(d) ? 1 : 0; }

extern "C" int64_t prv2_conv3x3_f6_weight_bytes(int32_t cout, int32_t cin) {
  return cout == 256 && cin >= 64 && cin % 64 == 0 ? (int64_t)(cin / 64) * 9 * f6::STEP_BYTES : 0;
}

extern "C" int prv2_pack_conv3x3_f6_weight(const float* w_src, float w_scale, void* w_packed, int32_t cout, int32_t cin, void* stream) {
  PRV2_REQUIRE(w_src && w_packed && aligned16(w_packed), "pack_conv3x3_f6_weight: null / unaligned pointer");
  PRV2_REQUIRE(prv2_conv3x3_f6_weight_bytes(cout, cin) > 0, "pack_conv3x3_f6_weight: 256 output channels, cin %% 64 == 0 (got %d -> %d)", cin, cout);
  PRV2_REQUIRE(w_scale > 0.f && (__builtin_bit_cast(unsigned, w_scale) & 0x7fffffu) == 0, "pack_conv3x3_f6_weight: w_scale is a power of two");
  const int nsteps = cin / 64 * 9;
  hipLaunchKernelGGL(f6_pack_weight_kernel, dim3((unsigned)(nsteps * 1024 / 256)), dim3(256), 0, (hipStream_t)stream, w_src, w_scale,
                     reinterpret_cast<char*>(w_packed), (int)cin, nsteps);
  PRV2_LAUNCH_CHECK("pack_conv3x3_f6_weight");
  return 0;
}

extern "C" int prv2_conv3x3_f6(const prv2_conv_desc* d, const float* x, const void* w_packed, const float* bias, const float* res, float x_scale,
                               float out_scale, uint32_t* range_word, float* y, void* stream) {
  PRV2_REQUIRE(d && x && w_packed && y, "conv3x3_f6: null pointer");
  PRV2_REQUIRE(f6_shape_ok(d), "conv3x3_f6: 3x3 s1 p1, cout 256, cin %% 64 == 0, width >= 16 (got %dx%d %d->%d k%d s%d)", d->h, d->w, d->cin, d->cout,
               d->kh, d->stride);
  PRV2_REQUIRE(d->act == PRV2_ACT_NONE && !(d->fmt & ~PRV2_FMT_Y_X2), "conv3x3_f6: no activation; the only format bit is PRV2_FMT_Y_X2");
  PRV2_REQUIRE(x_scale > 0.f && (__builtin_bit_cast(unsigned, x_scale) & 0x7fffffu) == 0 && out_scale > 0.f, "conv3x3_f6: x_scale is a power of two");
  F6Params p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.w = reinterpret_cast<const char*>(w_packed); p.bias = bias; p.res = res; p.y = y; p.range = range_word;
  p.N = d->n; p.H = d->h; p.W = d->w; p.Cin = d->cin; p.ldx = d->ldx; p.ldy = d->ldy; p.ld_res = d->ld_res;
  p.x_bstride = d->x_bstride ? d->x_bstride : (long long)d->h * d->w * d->ldx;
  p.y_bstride = d->y_bstride ? d->y_bstride : (long long)d->h * d->w * d->ldy;
  p.x_scale = x_scale; p.out_scale = out_scale; p.relu_in = d->relu_in; p.y_x2 = (d->fmt & PRV2_FMT_Y_X2) != 0;
  PRV2_REQUIRE(d->n > 0 && d->h > 0 && d->ldx >= d->cin && d->ldx % 4 == 0 && aligned16(x) && p.x_bstride % 4 == 0, "conv3x3_f6: x layout");
  PRV2_REQUIRE(d->ldy >= d->cout && d->ldy % 4 == 0 && aligned16(y) && p.y_bstride % 4 == 0, "conv3x3_f6: y layout");
  PRV2_REQUIRE(!bias || aligned16(bias), "conv3x3_f6: bias alignment");
  PRV2_REQUIRE(!res || (d->ld_res >= d->cout && d->ld_res % 4 == 0 && aligned16(res) && (long long)d->h * d->w * d->ld_res < (1LL << 29)), "conv3x3_f6: res layout");
  const int64_t ntiles = (int64_t)d->n * cdiv(d->h, f6::TH) * cdiv(d->w, f6::TW);
  PRV2_REQUIRE(ntiles < (1LL << 31), "conv3x3_f6: too many tiles");
  // persistent: one workgroup per CU (8 XCDs x 32), fewer when the tiles do not fill them
  static const int wgs = getenv("PRV2_F6_WGS") ? atoi(getenv("PRV2_F6_WGS")) : 256;  // A/B switch
  int64_t blocks = (ntiles + 7) / 8;
  blocks = (blocks > wgs / 8 ? wgs / 8 : blocks) * 8;
#ifdef F6_STAMPS
  p.stamps = getenv("PRV2_F6_STAMPS") ? (long long*)strtoull(getenv("PRV2_F6_STAMPS"), nullptr, 16) : nullptr;
#endif
  hipLaunchKernelGGL(conv3x3_c256_f6_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, p);
  set_kernel("conv3x3_c256_f6_kernel", 256, PRV2_PREC_F16F6);
  PRV2_LAUNCH_CHECK("conv3x3_f6");
  return 0;
}

extern "C" int prv2_conv3x3_ln_gate_f6(const prv2_conv_desc* d, const float* x, const void* w_packed, const float* bias, const float* pre, int32_t ld_pre,
                                       const float* ln_weight, const float* ln_bias, const void* gate_w_packed, const float* gate_bias, const float* mul,
                                       const float* res, float x_scale, float out_scale, uint32_t* range_word, float* y, void* stream) {
  PRV2_REQUIRE(d && x && w_packed && y && ln_weight && ln_bias && gate_w_packed, "conv3x3_ln_gate_f6: null pointer (the LayerNorm and the gate stage are part of the kernel)");
  PRV2_REQUIRE(f6_shape_ok(d), "conv3x3_ln_gate_f6: 3x3 s1 p1, cout 256, cin %% 64 == 0, width >= 16 (got %dx%d %d->%d k%d s%d)", d->h, d->w, d->cin, d->cout, d->kh,
               d->stride);
  const bool x2 = (d->fmt & PRV2_FMT_X_X2) != 0;
  PRV2_REQUIRE(!(d->fmt & ~(PRV2_FMT_X_X2 | PRV2_FMT_MUL_X2)) && (!mul || x2 == ((d->fmt & PRV2_FMT_MUL_X2) != 0)) && (mul || !(d->fmt & PRV2_FMT_MUL_X2)) && !d->relu_in,
               "conv3x3_ln_gate_f6: x and mul in the same format (fp32, or PRV2_FMT_X_X2 | PRV2_FMT_MUL_X2), no input ReLU, fp32 output");
  PRV2_REQUIRE(d->act == PRV2_ACT_RELU || d->act == PRV2_ACT_NONE, "conv3x3_ln_gate_f6: ReLU or no activation in front of the gate (act %d)", d->act);
  PRV2_REQUIRE(x_scale > 0.f && (__builtin_bit_cast(unsigned, x_scale) & 0x7fffffu) == 0 && out_scale > 0.f, "conv3x3_ln_gate_f6: x_scale is a power of two");
  const long long px = (long long)d->h * d->w;
  PRV2_REQUIRE(!pre || (ld_pre >= d->cout && ld_pre % 4 == 0 && aligned16(pre) && px * ld_pre < (1LL << 29)), "conv3x3_ln_gate_f6: pre layout");
  GateConvParams gp;
  memset(&gp, 0, sizeof(gp));
  IgemmParams& p = gp.c;
  p.x = x; p.w = w_packed; p.bias = bias; p.ln_w = ln_weight; p.ln_b = ln_bias; p.ln_eps = d->ln_eps; p.mul = mul; p.res = res; p.y = y;
  p.N = d->n; p.H = d->h; p.W = d->w; p.OH = d->h; p.OW = d->w;
  p.Cin = d->cin; p.Cin_pad = d->cin; p.Cout = d->cout; p.Ncols = d->cout;
  p.KH = 3; p.KW = 3; p.stride = 1; p.pad = 1; p.pad_x = 1;
  p.ldx = d->ldx; p.ldy = d->ldy; p.ld_mul = d->ld_mul; p.ld_res = d->ld_res;
  p.x_bstride = d->x_bstride ? d->x_bstride : px * d->ldx;
  p.y_bstride = d->y_bstride ? d->y_bstride : px * d->ldy;
  p.M = (long long)d->n * px;
  p.relu_in = 0; p.act = d->act; p.vec_ok = 1; p.vec_epi = 1;
  PRV2_REQUIRE(d->n > 0 && d->h > 0 && d->ldx >= d->cin && d->ldx % 4 == 0 && aligned16(x) && p.x_bstride % 4 == 0, "conv3x3_ln_gate_f6: x layout");
  PRV2_REQUIRE(d->ldy >= d->cout && d->ldy % 4 == 0 && aligned16(y) && p.y_bstride % 4 == 0, "conv3x3_ln_gate_f6: y layout");
  PRV2_REQUIRE(!mul || (d->ld_mul >= d->cout && d->ld_mul % 4 == 0 && aligned16(mul) && px * d->ld_mul < (1LL << 29)), "conv3x3_ln_gate_f6: mul layout");
  PRV2_REQUIRE(!res || (d->ld_res >= d->cout && d->ld_res % 4 == 0 && aligned16(res) && px * d->ld_res < (1LL << 29)), "conv3x3_ln_gate_f6: res layout");
  PRV2_REQUIRE(!bias || aligned16(bias), "conv3x3_ln_gate_f6: bias alignment");
  gp.gate_w = gate_w_packed; gp.gate_bias = gate_bias; gp.pre = pre; gp.ld_pre = ld_pre;
  gp.x_x2 = x2; gp.mul_x2 = mul && x2 ? 1 : 0; gp.y_x2 = 0;
  gp.f6_x_scale = x_scale; gp.f6_out_scale = out_scale; gp.f6_range = range_word;
#ifdef PRV2_GATE_STAMPS
  gp.stamps = getenv("PRV2_STAMP_PTR") ? (long long*)strtoull(getenv("PRV2_STAMP_PTR"), nullptr, 16) : nullptr;
#endif
  const int64_t blocks = (int64_t)d->n * cdiv(d->h, f6::TH) * cdiv(d->w, f6::TW);
  PRV2_REQUIRE(blocks < (1LL << 31), "conv3x3_ln_gate_f6: grid too large");
  if (x2) hipLaunchKernelGGL(conv3x3_c256_gate_f6_kernel<true>, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, gp);
  else hipLaunchKernelGGL(conv3x3_c256_gate_f6_kernel<false>, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, gp);
  set_kernel("conv3x3_c256_gate_f6_kernel", 256, PRV2_PREC_F16F6);
  PRV2_LAUNCH_CHECK("conv3x3_ln_gate_f6");
  return 0;
}
