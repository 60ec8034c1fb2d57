// Shared pieces of the implicit-GEMM kernels (igemm.hip: generic; conv3x3.hip: LDS-halo 3x3).
#pragma once
#include <type_traits>

#include "common.h"

namespace prv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct IgemmParams {
  const float* x;
  const void* w;
  const float* bias;
  const float* gamma;
  const float* mul;
  const float* res;
  const float* res2;
  const float* ln_w;   // fused channels-first LayerNorm (weight, bias) or null
  const float* ln_b;
  float ln_eps;
  float* y;
  int N, H, W, OH, OW;
  int Cin, Cin_pad, Cout, Ncols;  // Ncols = GEMM columns (= Cout, or k*k*Cout for convT)
  int KH, KW, stride, pad;  // pad = low-side padding in y
  int pad_x;                // low-side padding in x (== pad unless same_pad)
  int ldx, ldy, ld_mul, ld_res, ld_res2;
  long long x_bstride, y_bstride;
  long long M;
  int relu_in, act, convt_k, vec_ok, vec_epi;
  int tiles_m, tiles_n;
  int tiles_x;         // halo kernels: 32-pixel tile columns to process (the rest is the remainder strip)
  int strip_blocks;    // conv3x3_m16: leading workgroups of the grid that take the strip (32 x 8 tiles)
  int rx0, rw;         // remainder strip = output columns [rx0, rx0 + rw): tall halo tiles, or the generic kernel's window (rw = 0: none)
  int wave_map;        // conv3x3_m16: 1 = waves w, w + 4 (one SIMD) take the two channel halves and skip all-pad column blocks
  const void* w_tail;  // im2col tile of the last (Cin % 32 == 2) channels for the 16x16x32 halo kernel, or null
  // GatedConvUnit tail (conv3x3_m16.hip GATE / conv3x3_gate.hip): y = mul * sigmoid(W_g act(LN(conv + bias)) + gate_bias) (+ res)
  const void* gate_w;  // fragment-major Cout x Cout gate weights (prv2_pack_gate_weight), or null
  const float* gate_bias;
  // conv3x3_m16.hip UPS: input channels [0, ups_c) are bilinear(align_corners=True) samples of the low-resolution NHWC tensor xu
  // (uH x uW, pixel stride ldxu) taken while the halo is staged; channels [ups_c, Cin) come from x as usual.  null: none.
  const float* xu;
  int uH, uW, ldxu, ups_c;
  long long xu_bstride;
  float usy, usx;      // ac_scale(uH, H), ac_scale(uW, W)
  // conv3x3_m16.hip: [pred1 | pred2 | 0 | 0] written behind the Cout channels of every output pixel (prv2_conv2d_tail): the two
  // dense 1-channel maps tail1 / tail2 [N, tH, tW] resized bilinear(align_corners=True) to the output size.  null: none.
  const float* tail1;
  const float* tail2;
  int tH, tW;
  float tsy, tsx;
  // conv3x3_m16.hip: addend in front of the bias / LayerNorm / activation stage, [N, H, W, ld_pre >= Cout] (prv2_conv2d_pre: the conv's
  // coarse half from coarse_taps.hip).  null: none.
  const float* pre;
  int ld_pre;
};

// fragment-major gate weights: [16-column block cb][32-channel slab ks][hi, lo][lane 64] x 16 B, lane (m = lane & 15, g = lane >> 4)
// holding bf16 hi (resp. lo) of W[16 cb + m][32 ks + 8 g .. + 7]: one coalesced KB per MFMA B fragment
__host__ __device__ inline long long gate_frag_index(int c, int cb, int ks, int hl) { return (((long long)cb * (c / 32) + ks) * 2 + hl) * 64; }

// 3x3 convs whose Cin is a multiple of 32 plus the two appended depth maps ([feat | pred1 | pred2] concats:
// Cin = 34, 66, 98, 194, 322, 642, 770) carry one extra 128-byte tile per output row behind the regular packed
// weights: the 9 taps x 2 stray channels as ONE k = 32 slab (k = 2*tap + c, 18 used).  The 16x16x32 halo kernel
// then spends one MFMA step on them instead of nine steps of a 94 % empty slab (conv3x3_m16.hip).
static inline bool has_tail_tile(int cin, int kh, int kw, int convt_k, int prec) {
  return kh == 3 && kw == 3 && convt_k == 0 && prec != PRV2_PREC_F32 && cin > 32 && cin % 32 == 2;
}

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDS_LD = 36;  // floats per LDS row (32 + 4 pad)

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  v.x = fmaxf(v.x, 0.f);
  v.y = fmaxf(v.y, 0.f);
  v.z = fmaxf(v.z, 0.f);
  v.w = fmaxf(v.w, 0.f);
  return v;
}

__device__ __forceinline__ f32x4 floor4(f32x4 v, float lo) {
  v.x = fmaxf(v.x, lo);
  v.y = fmaxf(v.y, lo);
  v.z = fmaxf(v.z, lo);
  v.w = fmaxf(v.w, lo);
  return v;
}

__device__ __forceinline__ f32x4 zero_unless(f32x4 v, bool keep) {
  v.x = keep ? v.x : 0.f;
  v.y = keep ? v.y : 0.f;
  v.z = keep ? v.z : 0.f;
  v.w = keep ? v.w : 0.f;
  return v;
}

// fp32 -> bf16 hi + bf16 lo (v_cvt_pk_bf16_f32, round-to-nearest-even); hi + lo carries 16 mantissa bits
__device__ __forceinline__ void split_bf16(const f32x4 v, bf16x4& hi, bf16x4& lo) {
  hi = __builtin_convertvector(v, bf16x4);
  lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
}


// ---- MFMA fragments of one k-step for a wave: 2 row sub-tiles (32 rows each) x NJ column sub-tiles ----------
// a_row[i] / b_row[j] point at the lane's LDS row (A: 144-byte rows already offset by the lane half;
// B: either the same, or -- BSWZ -- unpadded 128-byte rows whose 16-byte slots are XOR-swizzled: slot c of
// the row lives at c ^ b_swz[j]; that is the image an LDS-DMA, which can only write linearly, produces from
// the pre-swizzled packed weights).  PREC_F32: k-step = 8 channels (4 MFMAs of k=2: lane half h takes channels
// 8ks+4h..+3 so MFMA e multiplies channel 8ks+4h+e on both operands); bf16 modes: k-step = 16 channels
// (v_mfma_f32_32x32x16_bf16: lane (r32, half) holds A[row r32][k = 8*half + j]).
template <int NJ, int PREC, int NI = 2>
struct Frags {
  f32x4 a[NI], b[NJ];
  bf16x8 ah[NI], al[NI], bh[NJ], bl[NJ];
};
template <int PREC>
constexpr int ksteps() { return PREC == PRV2_PREC_F32 ? 4 : 2; }

template <int NJ, int PREC, bool BSWZ, int NI = 2>
__device__ __forceinline__ void read_frags(Frags<NJ, PREC, NI>& f, int ks, const char* const (&a_row)[NI],
                                           const char* const (&b_row)[NJ], const int (&b_swz)[NJ], int half16) {
  auto b_at = [&](int j, int byte_off) -> const char* {
    if constexpr (BSWZ) return b_row[j] + ((((byte_off + half16) >> 4) ^ b_swz[j]) << 4);
    else return b_row[j] + byte_off;
  };
  if constexpr (PREC == PRV2_PREC_F32) {
#pragma unroll
    for (int i = 0; i < NI; ++i) f.a[i] = *reinterpret_cast<const f32x4*>(a_row[i] + ks * 32);
#pragma unroll
    for (int j = 0; j < NJ; ++j) f.b[j] = *reinterpret_cast<const f32x4*>(b_at(j, ks * 32));
  } else {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      f.ah[i] = *reinterpret_cast<const bf16x8*>(a_row[i] + ks * 32);
      if constexpr (PREC == PRV2_PREC_BF16X3) f.al[i] = *reinterpret_cast<const bf16x8*>(a_row[i] + 64 + ks * 32);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      f.bh[j] = *reinterpret_cast<const bf16x8*>(b_at(j, ks * 32));
      if constexpr (PREC == PRV2_PREC_BF16X3) f.bl[j] = *reinterpret_cast<const bf16x8*>(b_at(j, 64 + ks * 32));
    }
  }
}

// PART: 0 = all MFMAs of the k-step; 1 / 3 / 4 = its three portions (2 = 3 and 4 together).  bf16x3: the lo*hi,
// hi*lo and hi*hi products; other modes: row sub-tile 0, the rest, nothing.  Lets a caller drop other
// instructions (next fragment reads, global loads, LDS stores) between the portions.
template <int NJ, int PREC, int NI = 2, int PART = 0>
__device__ __forceinline__ void mma_frags(f32x16 (&acc)[NI][NJ], const Frags<NJ, PREC, NI>& f) {
  constexpr bool P1 = PART == 0 || PART == 1, P3 = PART == 0 || PART == 2 || PART == 3,
                 P4 = PART == 0 || PART == 2 || PART == 4;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const bool mine = (i == 0 && NI > 1) ? P1 : P3;  // single-product modes
      if constexpr (PREC == PRV2_PREC_F32) {
        if (mine) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i].x, f.b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i].y, f.b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i].z, f.b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i].w, f.b[j].w, acc[i][j], 0, 0, 0);
        }
      } else if constexpr (PREC == PRV2_PREC_BF16X3) {
        if (P1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
        if (P3) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
        if (P4) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
      } else {
        if (mine) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
      }
    }
}

// One BK=32 slab (all k-steps), fragments read on the fly
template <int NJ, int PREC, bool BSWZ = false, int NI = 2>
__device__ __forceinline__ void mma_slab(f32x16 (&acc)[NI][NJ], const char* const (&a_row)[NI],
                                         const char* const (&b_row)[NJ], const int (&b_swz)[NJ] = {}, int half16 = 0) {
#pragma unroll
  for (int ks = 0; ks < ksteps<PREC>(); ++ks) {
    Frags<NJ, PREC, NI> f;
    read_frags<NJ, PREC, BSWZ, NI>(f, ks, a_row, b_row, b_swz, half16);
    mma_frags<NJ, PREC, NI>(acc, f);
  }
}

// registers (one float4 of 4 channels) -> LDS row image of the precision mode
template <int PREC>
__device__ __forceinline__ void stage_a(float* row, int chunk, const f32x4 v) {
  if constexpr (PREC == PRV2_PREC_F32) {
    *reinterpret_cast<f32x4*>(row + chunk * 4) = v;
  } else {
    bf16x4 hi, lo;
    split_bf16(v, hi, lo);
    *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(row) + chunk * 8) = hi;
    if constexpr (PREC == PRV2_PREC_BF16X3) *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(row) + 64 + chunk * 8) = lo;
  }
}

// Fused epilogue for one output row segment of 4 channels (see include/prv2.h::prv2_conv2d).
struct EpiCols {
  int co, sub_y, sub_x, nvalid;
  bool vec;
  float bias[4], gam[4], lnw[4], lnb[4];
};

__device__ __forceinline__ bool epi_cols(const IgemmParams& p, int ncol, EpiCols& c) {
  if (ncol >= p.Ncols) return false;
  const int kk = p.convt_k > 0 ? p.convt_k : 1;
  c.co = ncol;
  c.sub_y = 0;
  c.sub_x = 0;
  if (p.convt_k > 0) {
    int t = ncol / p.Cout;
    c.co = ncol - t * p.Cout;
    c.sub_y = t / kk;
    c.sub_x = t - c.sub_y * kk;
  }
  c.nvalid = min(4, (p.convt_k > 0 ? p.Cout - c.co : p.Ncols - ncol));
  c.vec = p.vec_epi;  // (nvalid < 4 only with pad channels behind cout: written as zeros)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    c.bias[e] = (e < c.nvalid && p.bias) ? p.bias[c.co + e] : 0.f;
    c.gam[e] = (e < c.nvalid && p.gamma) ? p.gamma[c.co + e] : 1.f;
    c.lnw[e] = (e < c.nvalid && p.ln_w) ? p.ln_w[c.co + e] : 1.f;
    c.lnb[e] = (e < c.nvalid && p.ln_b) ? p.ln_b[c.co + e] : 0.f;
  }
  return true;
}

// Runs f(integral_constant<int, act>) for the layer's activation.  The store loops of the kernels go through this:
// with the activation a runtime value inside the loop, every iteration carried all five variants (erff, expf,
// log1pf expansions x 4 channels) -- ~30 KB of epilogue code per kernel against an 8 KB main loop.
template <class F>
__device__ __forceinline__ void dispatch_act(int act, F&& f) {
  switch (act) {
    case PRV2_ACT_RELU: f(std::integral_constant<int, PRV2_ACT_RELU>{}); break;
    case PRV2_ACT_GELU: f(std::integral_constant<int, PRV2_ACT_GELU>{}); break;
    case PRV2_ACT_SIGMOID: f(std::integral_constant<int, PRV2_ACT_SIGMOID>{}); break;
    case PRV2_ACT_SOFTPLUS: f(std::integral_constant<int, PRV2_ACT_SOFTPLUS>{}); break;
    case PRV2_ACT_SILU: f(std::integral_constant<int, PRV2_ACT_SILU>{}); break;
    default: f(std::integral_constant<int, PRV2_ACT_NONE>{}); break;
  }
}

// m = dense output pixel index (for mul/res/res2), o = element offset of y[m, co]; ACT >= 0: compile-time activation
template <int ACT = -1>
__device__ __forceinline__ void epi_store(const IgemmParams& p, const EpiCols& c, const f32x4 cv, long long m, long long o,
                                          float ln_mean = 0.f, float ln_rstd = 1.f) {
  float v[4] = {cv.x, cv.y, cv.z, cv.w};
  float mulv[4] = {1.f, 1.f, 1.f, 1.f}, resv[4] = {0.f, 0.f, 0.f, 0.f}, res2v[4] = {0.f, 0.f, 0.f, 0.f};
  if (c.vec) {
    if (p.mul) { f32x4 t = *reinterpret_cast<const f32x4*>(p.mul + m * p.ld_mul + c.co); mulv[0] = t.x; mulv[1] = t.y; mulv[2] = t.z; mulv[3] = t.w; }
    if (p.res) { f32x4 t = *reinterpret_cast<const f32x4*>(p.res + m * p.ld_res + c.co); resv[0] = t.x; resv[1] = t.y; resv[2] = t.z; resv[3] = t.w; }
    if (p.res2) { f32x4 t = *reinterpret_cast<const f32x4*>(p.res2 + m * p.ld_res2 + c.co); res2v[0] = t.x; res2v[1] = t.y; res2v[2] = t.z; res2v[3] = t.w; }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (e < c.nvalid && p.mul) mulv[e] = p.mul[m * p.ld_mul + c.co + e];
      if (e < c.nvalid && p.res) resv[e] = p.res[m * p.ld_res + c.co + e];
      if (e < c.nvalid && p.res2) res2v[e] = p.res2[m * p.ld_res2 + c.co + e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float t = v[e] + c.bias[e];
    if (p.ln_w) t = (t - ln_mean) * ln_rstd * c.lnw[e] + c.lnb[e];
    t = act_apply(t, ACT >= 0 ? ACT : p.act);
    if (p.gamma) t *= c.gam[e];
    if (p.mul) t = mulv[e] * t;
    if (p.res) t += resv[e];
    if (p.res2) t += res2v[e];
    v[e] = e < c.nvalid ? t : 0.f;
  }
  if (c.vec) {
    f32x4 ov = {v[0], v[1], v[2], v[3]};
    // Inline asm: hipcc puts s_waitcnt vmcnt(0) at the header of a loop that contains a store it knows about, i.e.
    // every iteration of the kernels' store loops waited ~900 cycles for the previous row's store to be acknowledged
    // (14 k of the 280 k cycles of a 512->256 tile: tools/probes/halo_phase_stamps.py).  A store needs no wait at all
    // (s_nop 1: the data registers are read right after issue); hidden stores only make hipcc's counted waits for the
    // gate / residual loads wait longer, never shorter (vmcnt retires in issue order).
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p.y + o), "v"(ov) : "memory");
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < c.nvalid) p.y[o + e] = v[e];
  }
}

// Row statistics for the fused LayerNorm: the C tile sits in LDS as [rows][cld]; thread t (< rows) owns row t.
// Two passes (mean, then biased variance of the centred values) exactly like convs.py:25-27.  Summation order: four partial sums --
// partial g over the channels 32 s + 8 g + i (i < 8, ascending) -- combined as (p0 + p1) + (p2 + p3): the order in which the four
// lanes of a pixel hold and reduce the channels in a wave-local epilogue (the round-3 four-wave experiment kernels, since removed), so that a layer's result does not depend on
// which of the kernels took it.
__device__ __forceinline__ void ln_row_stats(const IgemmParams& p, const float* ctile, int cld, int rows, int tid,
                                             float* stats /* [2*rows] in LDS */) {
  if (tid < rows) {
    const float* r = ctile + tid * cld;
    // Whole groups of eight channels come in as two 16-byte reads, added in the same ascending order (same bits).  One row per lane on a
    // pitch of 4 (mod 8) floats is conflict free for ds_read_b128's lane groups, but as scalar reads every 32-lane half met on 8 banks
    // (4-way): the LayerNorm layers of the narrow kernels spent a quarter of their LDS cycles in bank conflicts
    // (profiles/r04_bf16x3_pmc_lds_frame.txt; tools/lds_bank_model.py).
    const bool vec = (cld & 3) == 0 && (reinterpret_cast<size_t>(ctile) & 15) == 0;
    auto bias_of = [&](int c) { return p.bias ? p.bias[c] : 0.f; };
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < p.Cout; c0 += 32)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int b = c0 + 8 * g;
        if (vec && b + 8 <= p.Cout) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(r + b), v1 = *reinterpret_cast<const f32x4*>(r + b + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[g] += v0[e] + bias_of(b + e);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[g] += v1[e] + bias_of(b + 4 + e);
        } else {
          for (int c = b; c < b + 8 && c < p.Cout; ++c) s[g] += r[c] + bias_of(c);
        }
      }
    const float mean = ((s[0] + s[1]) + (s[2] + s[3])) / (float)p.Cout;
    float q[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < p.Cout; c0 += 32)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int b = c0 + 8 * g;
        if (vec && b + 8 <= p.Cout) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(r + b), v1 = *reinterpret_cast<const f32x4*>(r + b + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = v0[e] + bias_of(b + e) - mean;
            q[g] += d * d;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = v1[e] + bias_of(b + 4 + e) - mean;
            q[g] += d * d;
          }
        } else {
          for (int c = b; c < b + 8 && c < p.Cout; ++c) {
            const float d = r[c] + bias_of(c) - mean;
            q[g] += d * d;
          }
        }
      }
    stats[tid] = mean;
    stats[rows + tid] = 1.0f / sqrtf(((q[0] + q[1]) + (q[2] + q[3])) / (float)p.Cout + p.ln_eps);
  }
}

// conv3x3.hip: LDS-halo kernel for 3x3 / stride 1 / pad 1; returns false when the shape is not covered
bool conv3x3_halo_supported(const IgemmParams& p);
void launch_conv3x3_halo(IgemmParams& p, int prec, hipStream_t stream);
bool conv3x3_halo16_usable(const IgemmParams& p, int prec);
void launch_conv3x3_halo16(IgemmParams& p, int prec, hipStream_t stream);  // tiles [0, tiles_x) + strip [rx0, rx0 + rw)
int conv2d_impl(const prv2_conv_desc* d, const float* x, const void* w_packed, const float* bias, const float* ln_weight, const float* ln_bias,
                const float* gamma, const float* mul, const float* res, const float* res2, float* y, void* stream, const void* gate_w,
                const float* gate_bias, const prv2_ups_src* ups = nullptr, const float* tail1 = nullptr, const float* tail2 = nullptr,
                int tail_h = 0, int tail_w = 0, const float* pre = nullptr, int ld_pre = 0);  // igemm.hip: prv2_conv2d, with the optional gate stage of the 32 / 128-channel layers /
                                                  // the fused-upsample loader / the depth-pair tail
bool conv3x3_halo16_ups_usable(const IgemmParams& p, int prec);  // p.xu layers (prv2_conv2d_ups): the 128-column 16x16x32 halo kernel
bool conv3x3_halo16_gate_usable(const IgemmParams& p, int prec);                 // p.gate_w layers: Cout == 32 or 128, Cin % 32 == 0
void launch_conv3x3_halo16_gate(IgemmParams& p, int prec, hipStream_t stream);
// conv3x3_gate.hip: 3x3 convs with 256 output channels (8 x 16 pixel tiles x all channels; fused LayerNorm / gate tail)
bool conv3x3_c256_eligible(const prv2_conv_desc* d, const float* x, const float* res, const float* y);
// gemm_m16.hip: dense 1x1 / linear layers in the bf16 modes
bool gemm16_supported(const IgemmParams& p, int prec);
void launch_gemm16(IgemmParams& p, int prec, hipStream_t stream);

}  // namespace prv2
