// Flash-style softmax((q*scale) k^T) v for the DINOv2 ViT blocks (head_dim 64, no mask, 1025
// tokens; external/depth_anything_v2/dinov2_layers/attention.py:49-62).
//
// One workgroup = 128 queries of one (batch, head): 4 waves x 32 query rows.  Keys/values are
// streamed in tiles of 64 through LDS (shared by the 4 waves); each wave keeps its Q fragments
// (pre-scaled) and its 32x64 output accumulator in registers and runs the online softmax on the
// 32x32 MFMA accumulator layout directly: a row's 64 scores sit in one register index across
// the 32 lanes of a wave half, so row max / row sum are 5-step butterflies inside the half.
// P goes through a per-wave LDS strip to become the A operand of the P*V product.
// fp32 path: v_mfma_f32_32x32x2_f32 (exact products, fp32 accumulate) -- same arithmetic class
// as the reference's fp32 matmuls.
#include <type_traits>

#include "common.h"

namespace prv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int AT_BQ = 128;   // queries per workgroup
constexpr int AT_BK = 64;    // keys per tile
constexpr int AT_LD = 68;    // LDS row stride (64 + 4 pad floats): b128 reads conflict free

__device__ __forceinline__ float half_max(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// HAS_BIAS: scores += bias[head][q][key] before the softmax (BEiT relative position bias; rows of ldb floats)
template <bool HAS_BIAS>
__global__ void __launch_bounds__(256) attention_f32_kernel(const float* __restrict__ qkv, int B, int N, int heads,
                                                            float scale, const float* __restrict__ bias, int ldb,
                                                            float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float smem[(2 * AT_BK + 4 * 32) * AT_LD];
  float* Ks = smem;
  float* Vs = smem + AT_BK * AT_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, half = lane >> 5;
  float* Ps = smem + 2 * AT_BK * AT_LD + wave * 32 * AT_LD;

  const int qt = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  const int D3 = 3 * heads * 64;
  const float* base = qkv + (long long)b * N * D3 + head * 64;

  // Q fragments: lane (r32, half) holds Q[q0 + r32][8ks + 4half + e], pre-scaled
  const int q_row = qt * AT_BQ + wave * 32 + r32;
  float4 qf[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q_row < N) v = *reinterpret_cast<const float4*>(base + (long long)q_row * D3 + ks * 8 + half * 4);
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    qf[ks] = v;
  }

  f32x16 o_acc[2];
  float m_run[16], l_run[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    o_acc[0][e] = 0.f;
    o_acc[1][e] = 0.f;
    m_run[e] = -INFINITY;
    l_run[e] = 0.f;
  }

  const int ld_row = tid >> 4, ld_c = (tid & 15) * 4;  // 16 rows x 16 float4 per pass
  for (int k0 = 0; k0 < N; k0 += AT_BK) {
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = ld_row + 16 * i, key = k0 + row;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (key < N) {
        const float* src = base + (long long)key * D3 + ld_c;
        kv = *reinterpret_cast<const float4*>(src + heads * 64);
        vv = *reinterpret_cast<const float4*>(src + 2 * heads * 64);
      }
      *reinterpret_cast<float4*>(&Ks[row * AT_LD + ld_c]) = kv;
      *reinterpret_cast<float4*>(&Vs[row * AT_LD + ld_c]) = vv;
    }
    __syncthreads();

    // S = Q K^T : two 32x32 tiles (keys 0..31, 32..63 of this tile)
    f32x16 s_acc[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s_acc[0][e] = 0.f;
      s_acc[1][e] = 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float4 kb = *reinterpret_cast<const float4*>(&Ks[(j * 32 + r32) * AT_LD + ks * 8 + half * 4]);
        s_acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[ks].x, kb.x, s_acc[j], 0, 0, 0);
        s_acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[ks].y, kb.y, s_acc[j], 0, 0, 0);
        s_acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[ks].z, kb.z, s_acc[j], 0, 0, 0);
        s_acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[ks].w, kb.w, s_acc[j], 0, 0, 0);
      }
    }
    // online softmax; element (reg, lane): row = (reg&3)+8*(reg>>2)+4*half, key = k0 + j*32 + r32
    const bool kv0 = (k0 + r32) < N, kv1 = (k0 + 32 + r32) < N;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float s0 = kv0 ? s_acc[0][e] : -INFINITY, s1 = kv1 ? s_acc[1][e] : -INFINITY;
      if (HAS_BIAS) {
        const int qb = qt * AT_BQ + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        const float* br = bias + ((long long)head * N + (qb < N ? qb : 0)) * ldb + k0 + r32;
        if (kv0) s0 += br[0];
        if (kv1) s1 += br[32];
      }
      float mx = half_max(fmaxf(s0, s1));
      float m_new = fmaxf(m_run[e], mx);
      float corr = expf(m_run[e] - m_new);  // exp(-inf) = 0 on the first tile
      float p0 = kv0 ? expf(s0 - m_new) : 0.f, p1 = kv1 ? expf(s1 - m_new) : 0.f;
      l_run[e] = l_run[e] * corr + half_sum(p0 + p1);
      m_run[e] = m_new;
      o_acc[0][e] *= corr;
      o_acc[1][e] *= corr;
      const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
      Ps[row * AT_LD + r32] = p0;
      Ps[row * AT_LD + 32 + r32] = p1;
    }
    // the P strip is private to this wave: make the writes visible to its own later reads
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();

    // O += P V : A = P[q][key] (float4 along keys), B = V[key][d] (4 x b32, d on lanes)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      float4 pa = *reinterpret_cast<const float4*>(&Ps[r32 * AT_LD + ks * 8 + half * 4]);
      const float* vrow = &Vs[(ks * 8 + half * 4) * AT_LD + r32];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        o_acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.x, vrow[0 * AT_LD + j * 32], o_acc[j], 0, 0, 0);
        o_acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.y, vrow[1 * AT_LD + j * 32], o_acc[j], 0, 0, 0);
        o_acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.z, vrow[2 * AT_LD + j * 32], o_acc[j], 0, 0, 0);
        o_acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.w, vrow[3 * AT_LD + j * 32], o_acc[j], 0, 0, 0);
      }
    }
  }

  // out[b, q, head, d]: d = j*32 + r32 on lanes, rows in registers
  const int D = heads * 64;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
    const int q = qt * AT_BQ + wave * 32 + row;
    if (q < N) {
      float inv = 1.0f / l_run[e];
      float* dst = out + ((long long)b * N + q) * D + head * 64 + r32;
      dst[0] = o_acc[0][e] * inv;
      dst[32] = o_acc[1][e] * inv;
    }
  }
}

}  // namespace prv2

using namespace prv2;

int launch_attention_bf16x3(const float* qkv, int b, int ntok, int heads, const float* bias, int ld_bias, float* out, void* out_ss,
                            void* workspace, int64_t workspace_bytes, hipStream_t s);

extern "C" int prv2_attention_bias(const float* qkv, int32_t b, int32_t ntok, int32_t heads, int32_t hd, const float* bias,
                                   int32_t ld_bias, float* out, int32_t prec, void* workspace, int64_t workspace_bytes,
                                   void* stream);

extern "C" int prv2_attention(const float* qkv, int32_t b, int32_t ntok, int32_t heads, int32_t hd, float* out,
                              int32_t prec, void* workspace, int64_t workspace_bytes, void* stream) {
  return prv2_attention_bias(qkv, b, ntok, heads, hd, nullptr, 0, out, prec, workspace, workspace_bytes, stream);
}

extern "C" int prv2_attention_ss(const float* qkv, int32_t b, int32_t ntok, int32_t heads, int32_t hd, const float* bias, int32_t ld_bias,
                                 void* out_ss, void* workspace, int64_t workspace_bytes, void* stream) {
  PRV2_REQUIRE(qkv && out_ss, "attention_ss: null pointer");
  PRV2_REQUIRE(b > 0 && ntok > 0 && heads > 0 && hd == 64, "attention: head_dim must be 64 (got %d)", hd);
  PRV2_REQUIRE((reinterpret_cast<uintptr_t>(qkv) & 15) == 0 && (reinterpret_cast<uintptr_t>(out_ss) & 15) == 0, "attention_ss: 16-byte alignment");
  PRV2_REQUIRE(!bias || ld_bias == PRV2_ATTENTION_BIAS_IMAGE || (ld_bias >= prv2::roundup(ntok, 64) && ld_bias % 4 == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0),
               "attention: bias rows must be 16-byte aligned and padded to a multiple of 64 keys (ld_bias %d, ntok %d)", ld_bias, ntok);
  int rc = launch_attention_bf16x3(qkv, b, ntok, heads, bias, ld_bias, nullptr, out_ss, workspace, workspace_bytes, (hipStream_t)stream);
  if (rc) return rc;
  PRV2_LAUNCH_CHECK("attention_ss");
  return 0;
}

extern "C" int prv2_attention_bias(const float* qkv, int32_t b, int32_t ntok, int32_t heads, int32_t hd, const float* bias,
                                   int32_t ld_bias, float* out, int32_t prec, void* workspace, int64_t workspace_bytes,
                                   void* stream) {
  PRV2_REQUIRE(qkv && out, "attention: null pointer");
  PRV2_REQUIRE(!bias || (ld_bias == PRV2_ATTENTION_BIAS_IMAGE && prec != PRV2_PREC_F32) ||
                   (ld_bias >= prv2::roundup(ntok, 64) && ld_bias % 4 == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0),
               "attention: bias rows must be 16-byte aligned and padded to a multiple of 64 keys (ld_bias %d, ntok %d); the packed image is for the bf16 modes", ld_bias, ntok);
  PRV2_REQUIRE(b > 0 && ntok > 0 && heads > 0 && hd == 64, "attention: head_dim must be 64 (got %d)", hd);
  PRV2_REQUIRE(prec >= PRV2_PREC_F32 && prec <= PRV2_PREC_BF16, "attention: unknown precision mode %d", prec);
  PRV2_REQUIRE((reinterpret_cast<uintptr_t>(qkv) & 15) == 0, "attention: qkv must be 16-byte aligned");
  if (prec != PRV2_PREC_F32) {  // bf16 mode also runs the bf16x3 kernel (attention is never the bottleneck there)
    int rc = launch_attention_bf16x3(qkv, b, ntok, heads, bias, ld_bias, out, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
    if (rc) return rc;
    PRV2_LAUNCH_CHECK("attention(bf16x3)");
    return 0;
  }
  dim3 grid((unsigned)cdiv(ntok, AT_BQ), (unsigned)heads, (unsigned)b);
  if (bias) hipLaunchKernelGGL(attention_f32_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, qkv, b, ntok, heads, 0.125f, bias, ld_bias, out);
  else hipLaunchKernelGGL(attention_f32_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, qkv, b, ntok, heads, 0.125f, bias, ld_bias, out);
  PRV2_LAUNCH_CHECK("attention");
  return 0;
}

// =================================================================================================
// Split-bf16 (bf16x3) attention.
//
// Pre-pass (qkv_split_kernel): q (pre-scaled by 1/8 -- exact), k -> rows [64 x bf16 hi | 64 x bf16 lo];
// v -> transposed planes Vt[b, head, d, key] (hi, lo), keys zero-padded to a multiple of 64.  One pass
// over qkv; afterwards every operand of the attention kernel is a plain 16-byte copy away from its
// MFMA fragment (no conversion arithmetic in the hot loop).
//
// attention_bf16x3_kernel: one workgroup = 128 queries of one (batch, head), 4 waves x 32 queries.
// S^T = K Q^T is computed (keys on accumulator rows, queries on lanes), so
//   * the online softmax of a query is a per-LANE reduction over its 32 accumulator registers plus one
//     cross-half shuffle (instead of 16 five-step butterflies), and the O rescale is a per-lane scalar;
//   * P^T in accumulator layout IS the B operand of O^T = V^T P^T (cdna_hip_programming.md section 3,
//     "an accumulator tile as the next MFMA's operand"): registers 8s..8s+7 -> k-step s, no LDS round trip.
// V^T fragments follow the permuted k order of that trick: element j of lane half h is key
// 16s + 8(j>>2) + 4h + (j&3) -> two ds_read_b64 per fragment from the [d][key] LDS image.
// All products are hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 softmax.
// =================================================================================================
namespace prv2 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split4(const f32x4 v, bf16x4& hi, bf16x4& lo) {
  hi = __builtin_convertvector(v, bf16x4);
  lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
}

// grid (ceil(N/64), heads, B), 256 threads
__global__ void __launch_bounds__(256) qkv_split_kernel(const float* __restrict__ qkv, int N, int heads, int Npad,
                                                        __bf16* __restrict__ Qs, __bf16* __restrict__ Ks,
                                                        __bf16* __restrict__ VtH, __bf16* __restrict__ VtL) {
  __shared__ float vt[64][65];
  const int t0 = blockIdx.x * 64, head = blockIdx.y, b = blockIdx.z;
  const int D3 = 3 * heads * 64;
  const int tid = threadIdx.x, d4 = (tid & 15) * 4;
  const long long bh = (long long)b * heads + head;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int tl = (tid >> 4) + 16 * i, tok = t0 + tl;
    f32x4 q = {0.f, 0.f, 0.f, 0.f}, k = q, v = q;
    if (tok < N) {
      const float* src = qkv + ((long long)b * N + tok) * D3 + head * 64 + d4;
      // pre-scaled by hd^-0.5 * log2(e): the kernel's softmax runs in the base-2 domain (one v_exp_f32 per score instead of
      // expf's ten instructions; the loop was VALU-issue bound: 786 VALU per 48 MFMAs)
      q = *reinterpret_cast<const f32x4*>(src) * (0.125f * 1.4426950408889634f);
      k = *reinterpret_cast<const f32x4*>(src + heads * 64);
      v = *reinterpret_cast<const f32x4*>(src + 2 * heads * 64);
      bf16x4 hi, lo;
      __bf16* qd = Qs + (bh * N + tok) * 128 + d4;
      split4(q, hi, lo);
      *reinterpret_cast<bf16x4*>(qd) = hi;
      *reinterpret_cast<bf16x4*>(qd + 64) = lo;
      __bf16* kd = Ks + (bh * N + tok) * 128 + d4;
      split4(k, hi, lo);
      *reinterpret_cast<bf16x4*>(kd) = hi;
      *reinterpret_cast<bf16x4*>(kd + 64) = lo;
    }
    vt[tl][d4] = v.x; vt[tl][d4 + 1] = v.y; vt[tl][d4 + 2] = v.z; vt[tl][d4 + 3] = v.w;
  }
  __syncthreads();
  // transposed write: thread = (d, 16-key group); padded keys (tok >= N) carry zeros
  const int d = tid >> 2, kg = (tid & 3) * 16;
  __bf16* oh = VtH + (bh * 64 + d) * Npad + t0 + kg;
  __bf16* ol = VtL + (bh * 64 + d) * Npad + t0 + kg;
#pragma unroll
  for (int j = 0; j < 16; j += 4) {
    f32x4 v = {vt[kg + j][d], vt[kg + j + 1][d], vt[kg + j + 2][d], vt[kg + j + 3][d]};
    bf16x4 hi, lo;
    split4(v, hi, lo);
    *reinterpret_cast<bf16x4*>(oh + j) = hi;
    *reinterpret_cast<bf16x4*>(ol + j) = lo;
  }
}

constexpr int AB_KP = 272;  // K tile row pitch (bytes): 128 hi + 128 lo + 16 pad -> conflict-free ds_read_b128
constexpr int AB_VP = 264;  // V^T tile row pitch (bytes): conflict-free ds_read_b64 (66 dwords: 2r mod 64)

__device__ __forceinline__ f32x16 mfma3(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

// (launch bounds: THREE workgroups per CU -- the kernel needed 169 (with bias 202) registers, one more than three waves per SIMD
//  allow; asking for 168 costs nothing and a third independent wave per SIMD fills the softmax / staging gaps of the other two)
// BIAS: 0 none, 1 rows [heads][N][ldb] (every lane walks its own query's row: 64 cache lines per load instruction), 2 the
// prv2_pack_attention_bias image -- the same values pre-multiplied by log2 e in the order the S^T accumulators want them, so that a
// key tile's bias is eight coalesced 1 KB loads per wave, requested BEFORE the S^T MFMAs (BEiT: 0.17 -> see profiles/r04_*)
// (the bias-image variant needs 198 registers: under the three-workgroup cap of 168 it spilled 12 of them and reloaded them from scratch
//  inside the key-tile loop -- two workgroups per CU without scratch: BEiT-L attention 0.675 -> 0.630 ms per 41 x 769-token launch, same bits)
#ifndef ATT_WG_BIAS
#define ATT_WG_BIAS 2
#endif
template <int BIAS>
__global__ void __launch_bounds__(256, BIAS == 2 ? ATT_WG_BIAS : 3) attention_bf16x3_kernel(const __bf16* __restrict__ Qs,
                                                                  const __bf16* __restrict__ Ks,
                                                                  const __bf16* __restrict__ VtH,
                                                                  const __bf16* __restrict__ VtL, int N, int Npad,
                                                                  int heads, const float* __restrict__ bias, int ldb,
                                                                  float* __restrict__ out, char* __restrict__ out_ss) {
  __shared__ __attribute__((aligned(16))) char smem[64 * AB_KP + 64 * AB_VP];  // 34304 B; reused for the output strips
  char* const Kt = smem;
  char* const Vt = smem + 64 * AB_KP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, half = lane >> 5;
  // 1-D grid, XCD-aware: workgroups are dealt round-robin over the 8 XCDs (ids b, b + 8, ... share one), so consecutive ids
  // are remapped to one XCD -- the query tiles of a (batch, head) then read its K / V through ONE L2 instead of eight
  // (PMC before: L2 hit rate 26 %, 5.5x the unique bytes fetched)
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int qtiles = (N + AT_BQ - 1) / AT_BQ;
  const int qt = bid % qtiles, head = (bid / qtiles) % heads, b = bid / (qtiles * heads);
  const long long bh = (long long)b * heads + head;

  // Q fragments (B operand of S^T): lane (q = r32, half) holds Q[q][16ks + 8half + j]
  const int q_row = qt * AT_BQ + wave * 32 + r32;
  const bool wave_active = qt * AT_BQ + wave * 32 < N;  // wave-uniform
  bf16x8 qh[4], ql[4];
  {
    const __bf16* qp = Qs + (bh * N + (q_row < N ? q_row : 0)) * 128 + half * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qh[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
      ql[ks] = *reinterpret_cast<const bf16x8*>(qp + 64 + ks * 16);
    }
  }
  f32x16 o_acc[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) { o_acc[0][e] = 0.f; o_acc[1][e] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  const int ld_row = tid >> 2, ld_q = tid & 3;  // staging: 64 rows x 4 quarters (64 B each) per plane
  // K / V tile k+1 travels from global memory into registers WHILE tile k is multiplied, and is written to LDS behind the
  // barrier that retires tile k (load-early / write-late): the load latency used to be exposed once per 64 keys
  f32x4 kreg[4];
  uint2 vreg[8];
  auto fetch = [&](int k0) {
    // K rows: [hi 128 B | lo 128 B] (256 B contiguous in Ks); V^T rows: 128 B from each plane
    const int key = k0 + ld_row;
    const char* ksrc = reinterpret_cast<const char*>(Ks + (bh * N + (key < N ? key : 0)) * 128) + ld_q * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) kreg[i] = *reinterpret_cast<const f32x4*>(ksrc + i * 16);
    const long long vrow = (bh * 64 + ld_row) * Npad + k0;
    const char* vsrc = reinterpret_cast<const char*>((ld_q < 2 ? VtH : VtL) + vrow) + (ld_q & 1) * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) vreg[i] = *reinterpret_cast<const uint2*>(vsrc + i * 8);
  };
  fetch(0);
  auto tile = [&](int k0, auto mask_c) {
    constexpr bool MASK = decltype(mask_c)::value;  // keys behind N exist only in the last tile
    __syncthreads();
    {
      const bool kvalid = (k0 + ld_row) < N;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(Kt + ld_row * AB_KP + ld_q * 64 + i * 16) = kvalid ? kreg[i] : z;
#pragma unroll
      for (int i = 0; i < 8; ++i)  // 8-byte stores: the 264-byte pitch is only 8-byte aligned
        *reinterpret_cast<uint2*>(Vt + ld_row * AB_VP + ld_q * 64 + i * 8) = vreg[i];
    }
    __syncthreads();
#ifndef PRV2_ATT_ABL_NOFETCH
    if (k0 + AT_BK < N) fetch(k0 + AT_BK);
#endif
    // A wave without a single valid query only helps staging K / V: 1025 (DINOv2) and 769 (BEiT) tokens are 8 resp. 6 full
    // query tiles plus ONE query -- three of the last workgroup's four waves would otherwise run all 96 MFMAs per key tile on
    // garbage (a ninth / seventh of the launch's matrix work)
    if (!wave_active) return;

    f32x4 btile[BIAS == 2 ? 8 : 1];
    if constexpr (BIAS == 2) {
      // image: [head][query block of 32][key tile of 64][i = 4 t + g][lane][4]; ldb = key tiles per row of blocks
      const int q32 = qt * (AT_BQ / 32) + wave, q32n = ((N + AT_BQ - 1) / AT_BQ) * (AT_BQ / 32);
      const f32x4* bt = reinterpret_cast<const f32x4*>(bias) + ((((long long)head * q32n + q32) * ldb + (k0 >> 6)) * 8) * 64 + lane;
#pragma unroll
      for (int i = 0; i < 8; ++i) btile[i] = bt[i * 64];
    }
    // S^T tiles: rows = keys (t*32 + row), cols = this wave's 32 queries
    f32x16 st[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { st[0][e] = 0.f; st[1][e] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const char* kr = Kt + (t * 32 + r32) * AB_KP + ks * 32 + half * 16;
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(kr);
        const bf16x8 kl = *reinterpret_cast<const bf16x8*>(kr + 128);
        st[t] = mfma3(kh, kl, qh[ks], ql[ks], st[t]);
      }
    if constexpr (BIAS == 1) {
      // + bias[head][this lane's query][key]: register e = 4g + i is key 8g + 4half + i -> one 16-byte load per g
      const float* br = bias + ((long long)head * N + (q_row < N ? q_row : 0)) * ldb + k0 + 4 * half;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(br + t * 32 + 8 * g) * 1.4426950408889634f;  // (base-2 domain)
          st[t][4 * g] += bv.x; st[t][4 * g + 1] += bv.y; st[t][4 * g + 2] += bv.z; st[t][4 * g + 3] += bv.w;
        }
    }
    if constexpr (BIAS == 2) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bv = btile[4 * t + g];
          st[t][4 * g] += bv.x; st[t][4 * g + 1] += bv.y; st[t][4 * g + 2] += bv.z; st[t][4 * g + 3] += bv.w;
        }
    }
    // online softmax of this lane's query column (keys live in the registers of both lane halves); scores are base-2 logits
    if constexpr (MASK) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = k0 + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
          st[t][e] = key < N ? st[t][e] : -INFINITY;
        }
    }
#ifndef PRV2_ATT_ABL_NOSOFTMAX  // timing ablation (results wrong; profiles/r03_experiments.txt)
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, st[t][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float corr = __builtin_amdgcn_exp2f(m_run - m_new);  // exp2(-inf) = 0 on the first tile
    float rs = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        st[t][e] = __builtin_amdgcn_exp2f(st[t][e] - m_new);  // exp2(-inf) = 0 for masked keys
        rs += st[t][e];
      }
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * corr + rs;
    m_run = m_new;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o_acc[0][e] *= corr; o_acc[1][e] *= corr; }
#else
    l_run += st[0][0];
#endif

    // O^T += V^T P^T : B fragments straight from the P^T accumulators, A = V^T rows from LDS
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f32x4 p0 = {st[t][8 * s], st[t][8 * s + 1], st[t][8 * s + 2], st[t][8 * s + 3]};
        f32x4 p1 = {st[t][8 * s + 4], st[t][8 * s + 5], st[t][8 * s + 6], st[t][8 * s + 7]};
        bf16x4 h0, l0, h1, l1;
#ifndef PRV2_ATT_ABL_NOSPLIT
        split4(p0, h0, l0);
        split4(p1, h1, l1);
#else
        h0 = __builtin_bit_cast(bf16x4, __builtin_shufflevector(p0, p0, 0, 1)); l0 = __builtin_bit_cast(bf16x4, __builtin_shufflevector(p0, p0, 2, 3));
        h1 = __builtin_bit_cast(bf16x4, __builtin_shufflevector(p1, p1, 0, 1)); l1 = __builtin_bit_cast(bf16x4, __builtin_shufflevector(p1, p1, 2, 3));
#endif
        const bf16x8 ph = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 pl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        const int koff = (t * 32 + s * 16 + half * 4) * 2;  // bytes; element j -> key + 8*(j>>2) + (j&3)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const char* vr = Vt + (dt * 32 + r32) * AB_VP + koff;
          const bf16x4 va = *reinterpret_cast<const bf16x4*>(vr), vb = *reinterpret_cast<const bf16x4*>(vr + 16);
          const bf16x4 wa = *reinterpret_cast<const bf16x4*>(vr + 128), wb = *reinterpret_cast<const bf16x4*>(vr + 144);
          const bf16x8 vh = __builtin_shufflevector(va, vb, 0, 1, 2, 3, 4, 5, 6, 7);
          const bf16x8 vl = __builtin_shufflevector(wa, wb, 0, 1, 2, 3, 4, 5, 6, 7);
          o_acc[dt] = mfma3(vh, vl, ph, pl, o_acc[dt]);
        }
      }
  };
  int k0 = 0;
  for (; k0 + AT_BK <= N; k0 += AT_BK) tile(k0, std::false_type{});
  if (k0 < N) tile(k0, std::true_type{});

  // O^T accumulators: col = query (lane), row = d.  Transpose through a per-wave LDS strip [32 q][64 d + pad]
  __syncthreads();
  float* strip = reinterpret_cast<float*>(smem) + wave * 32 * 66;
  const float inv = 1.0f / l_run;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) strip[r32 * 66 + dt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half] = o_acc[dt][e] * inv;
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  const int D = heads * 64;
  if (out_ss) {
    // the output feeds exactly one consumer, the proj Linear: written in its operand format (gemm_ss.hip: per 32 channels
    // [4 x 16 B bf16 hi | 4 x 16 B bf16 lo], slot c at c ^ ((row >> 1) & 7)); this head's 64 channels = two such groups
    for (int i = lane; i < 32 * 8; i += 64) {  // 32 rows x 8 groups of 8 channels
      const int qr = i >> 3, kg = i & 7;
      const int q = qt * AT_BQ + wave * 32 + qr;
      if (q < N) {
        const float* sp = strip + qr * 66 + kg * 8;
        const f32x4 v0 = {sp[0], sp[1], sp[2], sp[3]}, v1 = {sp[4], sp[5], sp[6], sp[7]};
        bf16x4 h0, l0, h1, l1;
        split4(v0, h0, l0);
        split4(v1, h1, l1);
        const long long row = (long long)b * N + q;
        char* const rowp = out_ss + row * ((long long)D * 4) + (head * 2 + (kg >> 2)) * 128;
        const int chunk = kg & 3, k2 = (int)((row >> 1) & 7);
        *reinterpret_cast<bf16x8*>(rowp + ((chunk ^ k2) << 4)) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        *reinterpret_cast<bf16x8*>(rowp + (((4 + chunk) ^ k2) << 4)) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
    return;
  }
  for (int i = lane; i < 32 * 16; i += 64) {  // 32 rows x 16 float4
    const int qr = i >> 4, c4 = (i & 15) * 4;
    const int q = qt * AT_BQ + wave * 32 + qr;
    if (q < N) {
      const float* sp = strip + qr * 66 + c4;
      f32x4 v = {sp[0], sp[1], sp[2], sp[3]};
      *reinterpret_cast<f32x4*>(out + ((long long)b * N + q) * D + head * 64 + c4) = v;
    }
  }
}

}  // namespace prv2

// =================================================================================================
// The same attention on the qkv Linear's SPLIT-SWIZZLED output (prv2_gemm_ss_qkv): no pre-pass.
//
// The qkv GEMM's epilogue already writes every channel as bf16 hi + bf16 lo (gemm_ss.hip: per 32 channels one 128-byte group
// [4 x 16 B hi | 4 x 16 B lo], 16-byte slot c of row r stored at c ^ ((r >> 1) & 7)) and scales the q third by hd^-0.5 log2 e -- exactly the
// values qkv_split_kernel produced, so this kernel's operands are 16-byte copies of that buffer and its results are bit-equal to
// attention_bf16x3_kernel's.  A head's 64 channels are two such groups = 256 contiguous bytes per token.  What the pre-pass also did was
// transpose V; here V stays [key][d] (rows like K's) in LDS and the A operand of O^T = V^T P^T comes out of ds_read_b64_tr_b16 (the hardware's
// 4 x 16 transposing read; cdna_hip_programming.md T10): lane group G = lane >> 4 reads keys 16 s + 4 (G >> 1) + (0..3) [and + 8] of d columns
// 32 dt + 16 (G & 1) + (0..15): the fragment's permuted k order (element j <-> key 16 s + 8 (j >> 2) + 4 half + (j & 3)) as before.
// V row pitch 320 B: the four key rows a 32-lane half reads are 64-byte segments at 0 / 64 / 128 / 192 mod 256 -- conflict free; staging writes
// (both tiles): 8 consecutive lanes = 128 contiguous bytes -- conflict free (the old image's ld_row / ld_q map was 2-way on every write: 21 % of
// the kernel's LDS cycles, profiles/r05_f16f6_pmc_vit_blocks_b14.txt).
// The pre-pass, its 341 MB per ViT-L block at 14 crops and the fp32 qkv tensor are gone.
// =================================================================================================
namespace prv2 {

typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int AQ_VP = 320;  // V tile row pitch (bytes): 128 hi + 128 lo + 64 pad

template <int BIAS>
__global__ void __launch_bounds__(256, BIAS == 0 ? 3 : ATT_WG_BIAS) attention_qkvss_kernel(const char* __restrict__ qkv_ss, int N, int heads,
                                                                                            const float* __restrict__ bias, int ldb,
                                                                                            float* __restrict__ out, char* __restrict__ out_ss) {
  __shared__ __attribute__((aligned(16))) char smem[64 * AB_KP + 64 * AQ_VP];  // 37888 B; reused for the output strips (33792 B)
  char* const Kt = smem;
  char* const Vt = smem + 64 * AB_KP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, half = lane >> 5;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int qtiles = (N + AT_BQ - 1) / AT_BQ;
  const int qt = bid % qtiles, head = (bid / qtiles) % heads, b = bid / (qtiles * heads);
  const int D = heads * 64;
  const long long ld = (long long)D * 12;                      // bytes per token row of [q | k | v]
  const char* const qbase = qkv_ss + (long long)head * 256;    // this head's q columns; k at + D * 4, v at + D * 8
  const long long row_b = (long long)b * N;                    // first row of this image (the swizzle key is a function of the GLOBAL row)

  // Q fragments (B operand of S^T): lane (q = r32, half) holds Q[q][16 ks + 8 half + j]: 32-channel group ks >> 1, logical slot 2 (ks & 1) + half
  const int q_row = qt * AT_BQ + wave * 32 + r32;
  const bool wave_active = qt * AT_BQ + wave * 32 < N;  // wave-uniform
  bf16x8 qh[4], ql[4];
  {
    const long long r = row_b + (q_row < N ? q_row : 0);
    const char* qp = qbase + r * ld;
    const int key = (int)((r >> 1) & 7);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int sl = 2 * (ks & 1) + half;
      qh[ks] = *reinterpret_cast<const bf16x8*>(qp + (ks >> 1) * 128 + ((sl ^ key) << 4));
      ql[ks] = *reinterpret_cast<const bf16x8*>(qp + (ks >> 1) * 128 + (((4 + sl) ^ key) << 4));
    }
  }
  f32x16 o_acc[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) { o_acc[0][e] = 0.f; o_acc[1][e] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  // staging: a tile = 64 keys x 16 destination slots of 16 B ([8 x hi | 8 x lo] of the head's 64 channels); thread -> slots tid + 256 i:
  // key (tid >> 4) + 16 i, slot tid & 15 -- 16 lanes read one key's 256 bytes, 8 lanes write 128 contiguous LDS bytes
  const int st_key = tid >> 4, st_slot = tid & 15;
  const int st_grp = (st_slot & 7) >> 2, st_log = (st_slot & 3) + 4 * (st_slot >> 3);  // source group / logical slot of destination slot
  f32x4 kreg[4], vreg[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = k0 + st_key + 16 * i;
      const long long r = row_b + (key < N ? key : 0);
      const char* src = qbase + r * ld + (long long)D * 4 + st_grp * 128 + ((st_log ^ (int)((r >> 1) & 7)) << 4);
      kreg[i] = *reinterpret_cast<const f32x4*>(src);
      vreg[i] = *reinterpret_cast<const f32x4*>(src + (long long)D * 4);
    }
  };
  fetch(0);
  // transposing V reads: lane (G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3) addresses key row 4 (G >> 1) + q, d columns 16 (G & 1) + 4 p
  const int tr_off = (4 * (lane >> 5) + ((lane >> 2) & 3)) * AQ_VP + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  auto tile = [&](int k0, auto mask_c) {
    constexpr bool MASK = decltype(mask_c)::value;
    __syncthreads();
    {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool kvalid = (k0 + st_key + 16 * i) < N;
        *reinterpret_cast<f32x4*>(Kt + (st_key + 16 * i) * AB_KP + st_slot * 16) = kvalid ? kreg[i] : z;
        *reinterpret_cast<f32x4*>(Vt + (st_key + 16 * i) * AQ_VP + st_slot * 16) = kvalid ? vreg[i] : z;
      }
    }
    __syncthreads();
    if (k0 + AT_BK < N) fetch(k0 + AT_BK);
    if (!wave_active) return;

    f32x4 btile[BIAS == 2 ? 8 : 1];
    if constexpr (BIAS == 2) {
      const int q32 = qt * (AT_BQ / 32) + wave, q32n = ((N + AT_BQ - 1) / AT_BQ) * (AT_BQ / 32);
      const f32x4* bt = reinterpret_cast<const f32x4*>(bias) + ((((long long)head * q32n + q32) * ldb + (k0 >> 6)) * 8) * 64 + lane;
#pragma unroll
      for (int i = 0; i < 8; ++i) btile[i] = bt[i * 64];
    }
    f32x16 st[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { st[0][e] = 0.f; st[1][e] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const char* kr = Kt + (t * 32 + r32) * AB_KP + ks * 32 + half * 16;
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(kr);
        const bf16x8 kl = *reinterpret_cast<const bf16x8*>(kr + 128);
        st[t] = mfma3(kh, kl, qh[ks], ql[ks], st[t]);
      }
    if constexpr (BIAS == 1) {
      const float* br = bias + ((long long)head * N + (q_row < N ? q_row : 0)) * ldb + k0 + 4 * half;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(br + t * 32 + 8 * g) * 1.4426950408889634f;
          st[t][4 * g] += bv.x; st[t][4 * g + 1] += bv.y; st[t][4 * g + 2] += bv.z; st[t][4 * g + 3] += bv.w;
        }
    }
    if constexpr (BIAS == 2) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bv = btile[4 * t + g];
          st[t][4 * g] += bv.x; st[t][4 * g + 1] += bv.y; st[t][4 * g + 2] += bv.z; st[t][4 * g + 3] += bv.w;
        }
    }
    if constexpr (MASK) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = k0 + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
          st[t][e] = key < N ? st[t][e] : -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, st[t][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float corr = __builtin_amdgcn_exp2f(m_run - m_new);
    float rs = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        st[t][e] = __builtin_amdgcn_exp2f(st[t][e] - m_new);
        rs += st[t][e];
      }
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * corr + rs;
    m_run = m_new;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o_acc[0][e] *= corr; o_acc[1][e] *= corr; }

    // O^T += V^T P^T : B fragments straight from the P^T accumulators, A = V^T by transposing reads of the [key][d] image
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f32x4 p0 = {st[t][8 * s], st[t][8 * s + 1], st[t][8 * s + 2], st[t][8 * s + 3]};
        f32x4 p1 = {st[t][8 * s + 4], st[t][8 * s + 5], st[t][8 * s + 6], st[t][8 * s + 7]};
        bf16x4 h0, l0, h1, l1;
        split4(p0, h0, l0);
        split4(p1, h1, l1);
        const bf16x8 ph = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 pl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
          const char* vr = Vt + (t * 32 + s * 16) * AQ_VP + dt * 64 + tr_off;
          const s16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vr));
          const s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vr + 8 * AQ_VP));
          const s16x4 wa = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vr + 128));
          const s16x4 wb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vr + 8 * AQ_VP + 128));
          const bf16x8 vh = __builtin_bit_cast(bf16x8, __builtin_shufflevector(va, vb, 0, 1, 2, 3, 4, 5, 6, 7));
          const bf16x8 vl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(wa, wb, 0, 1, 2, 3, 4, 5, 6, 7));
          o_acc[dt] = mfma3(vh, vl, ph, pl, o_acc[dt]);
        }
      }
  };
  int k0 = 0;
  for (; k0 + AT_BK <= N; k0 += AT_BK) tile(k0, std::false_type{});
  if (k0 < N) tile(k0, std::true_type{});

  __syncthreads();
  float* strip = reinterpret_cast<float*>(smem) + wave * 32 * 66;
  const float inv = 1.0f / l_run;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) strip[r32 * 66 + dt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half] = o_acc[dt][e] * inv;
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  if (out_ss) {
    for (int i = lane; i < 32 * 8; i += 64) {
      const int qr = i >> 3, kg = i & 7;
      const int q = qt * AT_BQ + wave * 32 + qr;
      if (q < N) {
        const float* sp = strip + qr * 66 + kg * 8;
        const f32x4 v0 = {sp[0], sp[1], sp[2], sp[3]}, v1 = {sp[4], sp[5], sp[6], sp[7]};
        bf16x4 h0, l0, h1, l1;
        split4(v0, h0, l0);
        split4(v1, h1, l1);
        const long long row = row_b + q;
        char* const rowp = out_ss + row * ((long long)D * 4) + (head * 2 + (kg >> 2)) * 128;
        const int chunk = kg & 3, k2 = (int)((row >> 1) & 7);
        *reinterpret_cast<bf16x8*>(rowp + ((chunk ^ k2) << 4)) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        *reinterpret_cast<bf16x8*>(rowp + (((4 + chunk) ^ k2) << 4)) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
    return;
  }
  for (int i = lane; i < 32 * 16; i += 64) {
    const int qr = i >> 4, c4 = (i & 15) * 4;
    const int q = qt * AT_BQ + wave * 32 + qr;
    if (q < N) {
      const float* sp = strip + qr * 66 + c4;
      f32x4 v = {sp[0], sp[1], sp[2], sp[3]};
      *reinterpret_cast<f32x4*>(out + (row_b + q) * D + head * 64 + c4) = v;
    }
  }
}

}  // namespace prv2

extern "C" int prv2_attention_qkv_ss(const void* qkv_ss, int32_t b, int32_t ntok, int32_t heads, int32_t hd, const float* bias, int32_t ld_bias,
                                     float* out, void* out_ss, void* stream) {
  using namespace prv2;
  PRV2_REQUIRE(qkv_ss && (out || out_ss) && !(out && out_ss), "attention_qkv_ss: null pointer / exactly one of out, out_ss");
  PRV2_REQUIRE(b > 0 && ntok > 0 && heads > 0 && hd == 64, "attention: head_dim must be 64 (got %d)", hd);
  PRV2_REQUIRE((reinterpret_cast<uintptr_t>(qkv_ss) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(out_ss) & 15) == 0,
               "attention_qkv_ss: 16-byte alignment");
  PRV2_REQUIRE(!bias || ld_bias == PRV2_ATTENTION_BIAS_IMAGE || (ld_bias >= roundup(ntok, 64) && ld_bias % 4 == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0),
               "attention: bias rows must be 16-byte aligned and padded to a multiple of 64 keys (ld_bias %d, ntok %d)", ld_bias, ntok);
  const char* q = reinterpret_cast<const char*>(qkv_ss);
  char* oss = reinterpret_cast<char*>(out_ss);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)(cdiv(ntok, AT_BQ) * heads * b));
  if (bias && ld_bias == PRV2_ATTENTION_BIAS_IMAGE) {
    int q32n = (int)cdiv(ntok, AT_BQ) * (AT_BQ / 32), ktn = (int)cdiv(ntok, AT_BK);
    (void)q32n;
    hipLaunchKernelGGL(attention_qkvss_kernel<2>, grid, dim3(256), 0, s, q, ntok, heads, bias, ktn, out, oss);
  } else if (bias) hipLaunchKernelGGL(attention_qkvss_kernel<1>, grid, dim3(256), 0, s, q, ntok, heads, bias, ld_bias, out, oss);
  else hipLaunchKernelGGL(attention_qkvss_kernel<0>, grid, dim3(256), 0, s, q, ntok, heads, bias, ld_bias, out, oss);
  PRV2_LAUNCH_CHECK("attention_qkv_ss");
  return 0;
}

namespace prv2 {
// bias rows [heads][N][ldb] -> the image attention_bf16x3_kernel<2> reads: thread = one f32x4 of it
__global__ void __launch_bounds__(256) pack_attention_bias_kernel(const float* __restrict__ bias, int heads, int N, int ldb, float* __restrict__ dst,
                                                                  int q32n, int ktn) {
  const long long total = (long long)heads * q32n * ktn * 8 * 64;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63), i = (int)((idx >> 6) & 7);
    long long rest = idx >> 9;
    const int kt = (int)(rest % ktn);
    rest /= ktn;
    const int q32 = (int)(rest % q32n), head = (int)(rest / q32n);
    const int r32 = lane & 31, half = lane >> 5, t = i >> 2, g = i & 3;
    const int q = q32 * 32 + r32, key = kt * 64 + t * 32 + 8 * g + 4 * half;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q < N) {
      const float* src = bias + ((long long)head * N + q) * ldb + key;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (key + e < N) v[e] = src[e] * 1.4426950408889634f;  // (the kernel's base-2 domain: the product it used to form per tile)
    }
    reinterpret_cast<f32x4*>(dst)[idx] = v;
  }
}
}  // namespace prv2

static void bias_image_dims(int ntok, int& q32n, int& ktn) {
  q32n = (int)prv2::cdiv(ntok, prv2::AT_BQ) * (prv2::AT_BQ / 32);
  ktn = (int)prv2::cdiv(ntok, prv2::AT_BK);
}

extern "C" int64_t prv2_attention_bias_image_bytes(int32_t heads, int32_t ntok) {
  int q32n, ktn;
  bias_image_dims(ntok, q32n, ktn);
  return (int64_t)heads * q32n * ktn * 8 * 64 * 16;
}

extern "C" int prv2_pack_attention_bias(const float* bias, int32_t heads, int32_t ntok, int32_t ld_bias, float* image, void* stream) {
  PRV2_REQUIRE(bias && image && heads > 0 && ntok > 0 && ld_bias >= ntok && (reinterpret_cast<uintptr_t>(image) & 15) == 0, "pack_attention_bias: bad arguments");
  int q32n, ktn;
  bias_image_dims(ntok, q32n, ktn);
  const long long total = (long long)heads * q32n * ktn * 8 * 64;
  hipLaunchKernelGGL(prv2::pack_attention_bias_kernel, dim3(prv2::flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, bias, (int)heads, (int)ntok, (int)ld_bias,
                     image, q32n, ktn);
  PRV2_LAUNCH_CHECK("pack_attention_bias");
  return 0;
}

extern "C" int64_t prv2_attention_workspace_bytes(int32_t b, int32_t ntok, int32_t heads, int32_t prec) {
  if (prec == PRV2_PREC_F32) return 0;
  const int64_t npad = prv2::roundup(ntok, 64);
  return (int64_t)b * heads * (2 * (int64_t)ntok * 128 + 2 * 64 * npad) * 2 + 256;
}

int launch_attention_bf16x3(const float* qkv, int b, int ntok, int heads, const float* bias, int ld_bias, float* out, void* out_ss,
                            void* workspace, int64_t workspace_bytes, hipStream_t s) {
  using namespace prv2;
  const int64_t need = prv2_attention_workspace_bytes(b, ntok, heads, PRV2_PREC_BF16X3);
  PRV2_REQUIRE(workspace && workspace_bytes >= need, "attention: workspace too small (%lld < %lld bytes)",
               (long long)workspace_bytes, (long long)need);
  PRV2_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "attention: workspace must be 256-byte aligned");
  const int npad = (int)roundup(ntok, 64);
  __bf16* Qs = reinterpret_cast<__bf16*>(workspace);
  __bf16* Ks = Qs + (int64_t)b * heads * ntok * 128;
  __bf16* VtH = Ks + (int64_t)b * heads * ntok * 128;
  __bf16* VtL = VtH + (int64_t)b * heads * 64 * npad;
  dim3 g1((unsigned)(npad / 64), (unsigned)heads, (unsigned)b);
  hipLaunchKernelGGL(qkv_split_kernel, g1, dim3(256), 0, s, qkv, ntok, heads, npad, Qs, Ks, VtH, VtL);
  dim3 g2((unsigned)(cdiv(ntok, AT_BQ) * heads * b));
  char* oss = reinterpret_cast<char*>(out_ss);
  if (bias && ld_bias == PRV2_ATTENTION_BIAS_IMAGE) {  // the prv2_pack_attention_bias image: ldb carries the key tiles per block row
    int q32n, ktn;
    bias_image_dims(ntok, q32n, ktn);
    hipLaunchKernelGGL(attention_bf16x3_kernel<2>, g2, dim3(256), 0, s, Qs, Ks, VtH, VtL, ntok, npad, heads, bias, ktn, out, oss);
  } else if (bias) hipLaunchKernelGGL(attention_bf16x3_kernel<1>, g2, dim3(256), 0, s, Qs, Ks, VtH, VtL, ntok, npad, heads, bias, ld_bias, out, oss);
  else hipLaunchKernelGGL(attention_bf16x3_kernel<0>, g2, dim3(256), 0, s, Qs, Ks, VtH, VtL, ntok, npad, heads, bias, ld_bias, out, oss);
  return 0;
}
