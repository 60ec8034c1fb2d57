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

__global__ void __launch_bounds__(256) attention_f32_kernel(const float* __restrict__ qkv, int B, int N, int heads,
                                                            float scale, float* __restrict__ out) {
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

extern "C" int prv2_attention(const float* qkv, int32_t b, int32_t ntok, int32_t heads, int32_t hd, float* out,
                              int32_t prec, void* stream) {
  PRV2_REQUIRE(qkv && out, "attention: null pointer");
  PRV2_REQUIRE(b > 0 && ntok > 0 && heads > 0 && hd == 64, "attention: head_dim must be 64 (got %d)", hd);
  // every precision mode currently runs the exact fp32-MFMA kernel (attention is ~1% of a V2 frame)
  PRV2_REQUIRE(prec >= PRV2_PREC_F32 && prec <= PRV2_PREC_BF16, "attention: unknown precision mode %d", prec);
  PRV2_REQUIRE((reinterpret_cast<uintptr_t>(qkv) & 15) == 0, "attention: qkv must be 16-byte aligned");
  dim3 grid((unsigned)cdiv(ntok, AT_BQ), (unsigned)heads, (unsigned)b);
  hipLaunchKernelGGL(attention_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, b, ntok, heads, 0.125f, out);
  PRV2_LAUNCH_CHECK("attention");
  return 0;
}
