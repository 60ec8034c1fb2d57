// Row LayerNorm(+act), ViT token plumbing, and the two small direct convolutions (Cout==1,
// depthwise).  All HBM-bound: 16-byte accesses along the channel dimension, one wave per row
// for the reductions (no LDS, no barriers).
#include <stdarg.h>

#include "common.h"

PRV2_NO_PACKED_FP32_BEGIN  // (common.h)

namespace prv2 {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// the kernel the calling thread's last prv2_conv2d dispatched to (prv2_last_kernel): "name<BN,prec>"
static thread_local char g_kernel[96] = "";
void set_kernel(const char* name, int bn, int prec) {
  static const char* const pn[4] = {"f32", "bf16x3", "bf16", "f16f6"};
  snprintf(g_kernel, sizeof(g_kernel), "%s<%d,%s>", name, bn, prec >= 0 && prec < 4 ? pn[prec] : "?");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one wave per row; the row lives in registers between the two passes (C <= 64*4*MAXV)
// OUT_SS: the normalised row is written in the split-swizzled operand format of gemm_ss.hip (128 bytes per 32 channels:
// [4 x 16 B bf16 hi | 4 x 16 B bf16 lo], slot c at c ^ ((row >> 1) & 7)); y / ldy then address bytes (ldy = 4 * C)
template <int MAXV, bool OUT_SS = false>
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, int64_t rows, int C, int ldx,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        float eps, int act, float* __restrict__ y, int ldy) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int nv = C >> 2;  // float4 per row
  for (int64_t r = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * wpb) {
    const float4* px = reinterpret_cast<const float4*>(x + r * ldx);
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      int j = lane + 64 * i;
      if (j < nv) {
        v[i] = px[j];
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
    }
    float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      int j = lane + 64 * i;
      if (j < nv) {
        float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + bb * bb) + (c * c + d * d);
      }
    }
    float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    float4* py = reinterpret_cast<float4*>(y + r * ldy);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      int j = lane + 64 * i;
      if (j < nv) {
        float4 ww = reinterpret_cast<const float4*>(w)[j], bb = reinterpret_cast<const float4*>(b)[j], o;
        o.x = act_apply((v[i].x - mean) * rstd * ww.x + bb.x, act);
        o.y = act_apply((v[i].y - mean) * rstd * ww.y + bb.y, act);
        o.z = act_apply((v[i].z - mean) * rstd * ww.z + bb.z, act);
        o.w = act_apply((v[i].w - mean) * rstd * ww.w + bb.w, act);
        if constexpr (!OUT_SS) py[j] = o;
        else {
          // lanes 2m / 2m + 1 hold channels 8m .. 8m + 7: both assemble the eight values, the even lane stores the
          // bf16 hi slot, the odd lane the lo slot (C % 8 == 0: partners are both inside the row)
          float4 other;
          other.x = __shfl_xor(o.x, 1, 64); other.y = __shfl_xor(o.y, 1, 64); other.z = __shfl_xor(o.z, 1, 64); other.w = __shfl_xor(o.w, 1, 64);
          const bool odd = lane & 1;
          const float4 lo4 = odd ? other : o, hi4 = odd ? o : other;  // channels 8m..8m+3, 8m+4..8m+7
          typedef float f4v __attribute__((ext_vector_type(4)));
          typedef __bf16 b4v __attribute__((ext_vector_type(4)));
          typedef __bf16 b8v __attribute__((ext_vector_type(8)));
          const f4v a0 = {lo4.x, lo4.y, lo4.z, lo4.w}, a1 = {hi4.x, hi4.y, hi4.z, hi4.w};
          const b4v h0 = __builtin_convertvector(a0, b4v), h1 = __builtin_convertvector(a1, b4v);
          b8v outv;
          if (!odd) outv = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
          else {
            const b4v l0 = __builtin_convertvector(a0 - __builtin_convertvector(h0, f4v), b4v);
            const b4v l1 = __builtin_convertvector(a1 - __builtin_convertvector(h1, f4v), b4v);
            outv = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
          }
          const int grp = j >> 1;  // 8-channel group of the row
          char* rowp = reinterpret_cast<char*>(y) + r * (int64_t)ldy + (grp >> 2) * 128;
          const int slot = ((grp & 3) + (odd ? 4 : 0)) ^ (int)((r >> 1) & 7);
          *reinterpret_cast<b8v*>(rowp + (slot << 4)) = outv;
        }
      }
    }
  }
}

// scalar fallback (C % 4 != 0 or unaligned): one wave per row, re-reads the row
__global__ void __launch_bounds__(256) layernorm_scalar_kernel(const float* __restrict__ x, int64_t rows, int C, int ldx,
                                                               const float* __restrict__ w, const float* __restrict__ b,
                                                               float eps, int act, float* __restrict__ y, int ldy) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * wpb) {
    const float* px = x + r * ldx;
    float s = 0.f;
    for (int j = lane; j < C; j += 64) s += px[j];
    float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int j = lane; j < C; j += 64) {
      float d = px[j] - mean;
      q += d * d;
    }
    float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    for (int j = lane; j < C; j += 64) y[r * ldy + j] = act_apply((px[j] - mean) * rstd * w[j] + b[j], act);
  }
}

__global__ void __launch_bounds__(256) patchify_kernel(const float* __restrict__ img, int B, int gh, int gw, int p,
                                                       int ldi, float* __restrict__ rows, int ldo) {
  const int kcols = p * p * 3;
  int64_t total = (int64_t)B * gh * gw * ldo;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int col = (int)(idx % ldo);
    int64_t row = idx / ldo;
    float v = 0.f;
    if (col < kcols) {
      int c = col % 3, kx = (col / 3) % p, ky = col / (3 * p);
      int gx = (int)(row % gw), gy = (int)((row / gw) % gh), b = (int)(row / ((int64_t)gw * gh));
      int64_t pix = ((int64_t)b * gh * p + gy * p + ky) * ((int64_t)gw * p) + gx * p + kx;
      v = img[pix * ldi + c];
    }
    rows[idx] = v;
  }
}

__global__ void __launch_bounds__(256) assemble_tokens_kernel(const float* __restrict__ emb,
                                                              const float* __restrict__ cls,
                                                              const float* __restrict__ pos, int B, int np, int D,
                                                              float* __restrict__ tok) {
  int64_t total = (int64_t)B * (np + 1) * D;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int d = (int)(idx % D);
    int t = (int)((idx / D) % (np + 1));
    int b = (int)(idx / ((int64_t)D * (np + 1)));
    float v = t == 0 ? cls[d] : emb[((int64_t)b * np + (t - 1)) * D + d];
    tok[idx] = v + pos[(int64_t)t * D + d];
  }
}

// Cout == 1, 1x1: one thread per output pixel, float4 over channels (a wave reads 64 consecutive
// pixel rows = one contiguous run)
__global__ void __launch_bounds__(256) conv_cout1_k1_kernel(const float* __restrict__ x, int64_t total, int Cin, int ldx,
                                                            const float* __restrict__ wgt,
                                                            const float* __restrict__ bias, int act, float scale,
                                                            const float* __restrict__ res, int clamp0,
                                                            float* __restrict__ y) {
  extern __shared__ float wl[];  // [Cin]
  for (int i = threadIdx.x; i < Cin; i += blockDim.x) wl[i] = wgt[i];
  __syncthreads();
  const float b0 = bias ? bias[0] : 0.f;
  const bool vec = (Cin % 4 == 0) && (ldx % 4 == 0);
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const float* px = x + idx * ldx;
    float acc = 0.f;
    if (vec) {
      for (int c = 0; c < Cin; c += 4) {
        float4 v = *reinterpret_cast<const float4*>(px + c);
        acc += v.x * wl[c] + v.y * wl[c + 1] + v.z * wl[c + 2] + v.w * wl[c + 3];
      }
    } else {
      for (int c = 0; c < Cin; ++c) acc += px[c] * wl[c];
    }
    float v = act_apply(acc + b0, act) * scale;
    if (res) v += res[idx];
    if (clamp0) v = v > 0.f ? v : 0.f;
    y[idx] = v;
  }
}

// Cout == 1, 3x3: workgroup = 8 x 32 output pixels (one per thread).  Per 32-channel slab the
// 10 x 34 halo is staged in LDS with coalesced 16-byte loads (rows padded to 36 floats: the
// per-pixel ds_read_b128 of a wave are conflict free) and every thread reads its 9 taps from
// LDS; weights sit in LDS as [tap][32] and are broadcast.  HBM traffic = one read of x.
constexpr int C1_TH = 8, C1_TW = 32, C1_HW = C1_TW + 2, C1_HALO = (C1_TH + 2) * C1_HW, C1_LD = 36;

__global__ void __launch_bounds__(256) conv_cout1_k3_kernel(const float* __restrict__ x, int N, int H, int W, int Cin,
                                                            int ldx, const float* __restrict__ wgt, int act,
                                                            float scale, const float* __restrict__ bias,
                                                            const float* __restrict__ res, int clamp0,
                                                            float* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) float tile[C1_HALO * C1_LD];
  __shared__ __attribute__((aligned(16))) float wl[9 * 32];
  const int tiles_x = (W + C1_TW - 1) / C1_TW, tiles_y = (H + C1_TH - 1) / C1_TH;
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y;
  const int n = t / tiles_y;
  const int y0 = ty * C1_TH, x0 = tx * C1_TW;
  const int tid = threadIdx.x;
  const int py = tid >> 5, px = tid & 31;
  const int chunk = tid & 7, prow = tid >> 3;
  const float* img = x + (int64_t)n * H * W * ldx;
  float acc = 0.f;
  for (int cb = 0; cb < Cin; cb += 32) {
    __syncthreads();
    for (int i = tid; i < 9 * 32; i += 256) {
      int c = cb + (i & 31), tap = i >> 5;
      wl[i] = c < Cin ? wgt[c * 9 + tap] : 0.f;
    }
    for (int hp = prow; hp < C1_HALO; hp += 32) {
      const int hy = hp / C1_HW, hx = hp - hy * C1_HW;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const int c = cb + chunk * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && c < Cin) {
        const float* src = img + ((int64_t)iy * W + ix) * ldx + c;
        if (c + 4 <= Cin) v = *reinterpret_cast<const float4*>(src);
        else { v.x = src[0]; if (c + 1 < Cin) v.y = src[1]; if (c + 2 < Cin) v.z = src[2]; }
      }
      *reinterpret_cast<float4*>(&tile[hp * C1_LD + chunk * 4]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float* pt = &tile[((py + tap / 3) * C1_HW + px + tap % 3) * C1_LD];
      const float* pw = &wl[tap * 32];
#pragma unroll
      for (int c = 0; c < 32; c += 4) {
        float4 v = *reinterpret_cast<const float4*>(pt + c);
        float4 ww = *reinterpret_cast<const float4*>(pw + c);
        acc += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
      }
    }
  }
  const int oy = y0 + py, ox = x0 + px;
  if (oy < H && ox < W) {
    const int64_t idx = ((int64_t)n * H + oy) * W + ox;
    float v = act_apply(acc + (bias ? bias[0] : 0.f), act) * scale;
    if (res) v += res[idx];
    if (clamp0) v = v > 0.f ? v : 0.f;
    y[idx] = v;
  }
}

// depthwise kxk, float4 over channels; weights tap-major [k*k][C]
__global__ void __launch_bounds__(256) dwconv_kernel(const float* __restrict__ x, int N, int H, int W, int C, int ldx,
                                                     const float* __restrict__ wgt, const float* __restrict__ bias, int k,
                                                     int stride, int act, int pad_y, int pad_x, int OH, int OW,
                                                     float* __restrict__ y, int ldy) {
  const int cg = C >> 2;
  int64_t total = (int64_t)N * OH * OW * cg;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(idx % cg) * 4;
    int64_t pix = idx / cg;
    int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH);
    int n = (int)(pix / ((int64_t)OW * OH));
    float4 acc = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < k; ++ky) {
      int iy = oy * stride + ky - pad_y;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < k; ++kx) {
        int ix = ox * stride + kx - pad_x;
        if (ix < 0 || ix >= W) continue;
        float4 v = *reinterpret_cast<const float4*>(x + (((int64_t)n * H + iy) * W + ix) * ldx + c);
        float4 ww = *reinterpret_cast<const float4*>(wgt + (int64_t)(ky * k + kx) * C + c);
        acc.x += v.x * ww.x;
        acc.y += v.y * ww.y;
        acc.z += v.z * ww.z;
        acc.w += v.w * ww.w;
      }
    }
    acc = make_float4(act_apply(acc.x, act), act_apply(acc.y, act), act_apply(acc.z, act), act_apply(acc.w, act));
    *reinterpret_cast<float4*>(y + pix * ldy + c) = acc;
  }
}

// depthwise KxK, stride 1: one thread = 4 channels x PX consecutive output pixels of a row.  The naive kernel above issues
// 2*K*K 16-byte loads per output float4 (load-issue bound: 7 TFLOP/s on the 7x7 ConvNeXt layers); here a row of PX + K - 1
// inputs serves PX outputs and a weight is loaded once per PX outputs.  Same summation order per output (bias, then
// taps row-major), out-of-image taps contribute +0.
template <int K, int PX>
__global__ void __launch_bounds__(256) dwconv_strip_kernel(const float* __restrict__ x, int N, int H, int W, int C, int ldx,
                                                           const float* __restrict__ wgt, const float* __restrict__ bias,
                                                           int act, float* __restrict__ y, int ldy) {
  constexpr int PAD = K / 2;
  const int cg = C >> 2;
  const int strips = (W + PX - 1) / PX;
  const int64_t total = (int64_t)N * H * strips * cg;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % cg) * 4;
  int64_t t = idx / cg;
  const int sx = (int)(t % strips);
  t /= strips;
  const int oy = (int)(t % H), n = (int)(t / H);
  const int ox0 = sx * PX;
  float4 acc[PX];
  const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < PX; ++i) acc[i] = b4;
#pragma unroll 1  // (fully unrolled hipcc hoists all K rows of loads: 512 VGPRs and spills)
  for (int ky = 0; ky < K; ++ky) {
    const int iy = oy + ky - PAD;
    const bool rok = (unsigned)iy < (unsigned)H;
    const float* row = x + (((int64_t)n * H + (rok ? iy : 0)) * W) * ldx + c;
    float4 in[PX + K - 1];
#pragma unroll
    for (int i = 0; i < PX + K - 1; ++i) {
      const int ix = ox0 + i - PAD;
      in[i] = (rok && (unsigned)ix < (unsigned)W) ? *reinterpret_cast<const float4*>(row + (int64_t)ix * ldx)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const float4 ww = *reinterpret_cast<const float4*>(wgt + (int64_t)(ky * K + kx) * C + c);
#pragma unroll
      for (int i = 0; i < PX; ++i) {
        acc[i].x += in[i + kx].x * ww.x;
        acc[i].y += in[i + kx].y * ww.y;
        acc[i].z += in[i + kx].z * ww.z;
        acc[i].w += in[i + kx].w * ww.w;
      }
    }
  }
  float* out = y + (((int64_t)n * H + oy) * W + ox0) * ldy + c;
#pragma unroll
  for (int i = 0; i < PX; ++i) {
    if (ox0 + i >= W) break;
    const float4 v = make_float4(act_apply(acc[i].x, act), act_apply(acc[i].y, act), act_apply(acc[i].z, act), act_apply(acc[i].w, act));
    *reinterpret_cast<float4*>(out + (int64_t)i * ldy) = v;
  }
}

// squeeze: out[n][c] = mean over the HW pixels, in two deterministic stages (fixed summation order, no atomics):
// stage 1: grid (channel-group blocks, pixel chunks, n); a block = CGW channel groups (float4) x 256/CGW pixel lanes sums its
// chunk into part[n][chunk][c]; stage 2 adds the chunks in index order.  (A first version with one block per image and
// channel block ran at 20 GB/s on the 24-channel 196 x 259 maps of EfficientNet's first stage: 14 blocks on 256 CUs.)
__global__ void __launch_bounds__(256) avgpool_partial_kernel(const float* __restrict__ x, int64_t HW, int C, int ldx, int cgw_log2,
                                                              int chunk_len, float* __restrict__ part) {
  __shared__ float4 sm[256];
  const int cgw = 1 << cgw_log2;
  const int cl = threadIdx.x & (cgw - 1), pl = threadIdx.x >> cgw_log2, npl = 256 >> cgw_log2;
  const int c = (blockIdx.x * cgw + cl) * 4;
  const int chunk = blockIdx.y, n = blockIdx.z, chunks = gridDim.y;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C) {
    const int64_t p0 = (int64_t)chunk * chunk_len, p1 = p0 + chunk_len < HW ? p0 + chunk_len : HW;
    const float* base = x + (int64_t)n * HW * ldx + c;
    for (int64_t pix = p0 + pl; pix < p1; pix += npl) {
      const float4 v = *reinterpret_cast<const float4*>(base + pix * ldx);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  sm[threadIdx.x] = acc;
  __syncthreads();
  if (pl == 0 && c < C) {
    float4 t = sm[cl];
    for (int g = 1; g < npl; ++g) {
      const float4 u = sm[(g << cgw_log2) + cl];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    *reinterpret_cast<float4*>(part + ((int64_t)n * chunks + chunk) * C + c) = t;
  }
}

__global__ void __launch_bounds__(256) avgpool_final_kernel(const float* __restrict__ part, int N, int C, int chunks, float inv,
                                                            float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * C) return;
  const int n = idx / C, c = idx - n * C;
  float s = 0.f;
  for (int k = 0; k < chunks; ++k) s += part[((int64_t)n * chunks + k) * C + c];
  out[idx] = s * inv;
}

// squeeze-excite bottleneck g[n][:] = sigmoid(W2 silu(W1 m[n][:] + b1) + b2) in fp32, two small launches that cover the chip:
// reduce: one 64-lane block per (bottleneck channel, image) -- a dot product over C; expand: one thread per (channel, image),
// W2 passed transposed ([cse][C]) so that the loop over the bottleneck reads coalesced rows.
// (As two [n, C] row GEMMs on the generic MFMA kernel these were 2 x 28 us for ~1 MFLOP; as ONE workgroup per image 230 us.)
__global__ void __launch_bounds__(64) se_reduce_kernel(const float* __restrict__ mean, int C, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, int CSE, float* __restrict__ r) {
  const int j = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
  const float* wr = w1 + (int64_t)j * C;
  const float* m = mean + (int64_t)n * C;
  float acc = 0.f;
  for (int c = lane; c < C; c += 64) acc += wr[c] * m[c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);  // fixed-order tree
  if (lane == 0) r[(int64_t)n * CSE + j] = act_apply(acc + (b1 ? b1[j] : 0.f), PRV2_ACT_SILU);
}

__global__ void __launch_bounds__(256) se_expand_kernel(const float* __restrict__ r, int C, int CSE, const float* __restrict__ w2t,
                                                        const float* __restrict__ b2, float* __restrict__ g) {
  const int c = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (c >= C) return;
  const float* rn = r + (int64_t)n * CSE;
  float acc = b2 ? b2[c] : 0.f;
  for (int j = 0; j < CSE; ++j) acc += w2t[(int64_t)j * C + c] * rn[j];
  g[(int64_t)n * C + c] = act_apply(acc, PRV2_ACT_SIGMOID);
}

// excite: x[n, pix, c] *= s[n][c]
__global__ void __launch_bounds__(256) channel_scale_kernel(float* __restrict__ x, int64_t HW, int C, int ldx,
                                                            const float* __restrict__ s, int64_t total) {
  const int cg = C >> 2;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cg) * 4;
    const int64_t pix = idx / cg;
    const int n = (int)(pix / HW);
    float4* q = reinterpret_cast<float4*>(x + pix * ldx + c);
    const float4 g = *reinterpret_cast<const float4*>(s + (int64_t)n * C + c);
    float4 v = *q;
    v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
    *q = v;
  }
}

__global__ void __launch_bounds__(256) add_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b,
                                                  int ldb, int64_t rows, int C, float* __restrict__ y, int ldy) {
  int64_t total = rows * C;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(idx % C);
    int64_t r = idx / C;
    y[r * ldy + c] = a[r * lda + c] + b[r * ldb + c];
  }
}

// zero the pad channels [c, ld) of an NHWC buffer (they are read by the conv loaders against zero weights and must be
// finite); one thread per pixel -- a full-buffer memset for 2 of 100 channels costs 50x the traffic
__global__ void __launch_bounds__(256) zero_pad_kernel(float* __restrict__ y, int64_t rows, int c, int ld) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x)
    for (int k = c; k < ld; ++k) y[r * ld + k] = 0.f;
}

// one thread per (pixel, bin): the pixel's attractor points are re-read from L1 by its n_bins threads
__global__ void __launch_bounds__(256) zoe_attractor_kernel(const float* __restrict__ attr, int ld_attr, int n_attr,
                                                            const float* __restrict__ bins, int ld_bins, int n_bins,
                                                            float alpha, int64_t rows, float* __restrict__ out,
                                                            int ld_out) {
  int64_t total = rows * n_bins;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int j = (int)(idx % n_bins);
    int64_t r = idx / n_bins;
    const float c = bins[r * ld_bins + j];
    const float* a = attr + r * ld_attr;
    float s = 0.f;
    for (int i = 0; i < n_attr; ++i) {
      float dx = a[i] - c;
      s += dx / (1.0f + alpha * (dx * dx));
    }
    out[r * ld_out + j] = c + s / (float)n_attr;
  }
}

// one thread per pixel: log-binomial logits over the bins, softmax at temperature t, expectation of the centres.
// The per-bin Stirling term depends on (k, K) only: computed once per block into LDS (same float operations as before).  The
// pixel's K bin centres are read through an LDS tile filled with coalesced row loads (a thread walking its own 256-byte row made
// every load instruction touch 64 cache lines: 5.2 ms per 41 x 384 x 512 launch, 16 ms per frame of the V1 ZoeDepth workload).
template <int KMAX>
__global__ void __launch_bounds__(256) zoe_logbinom_depth_kernel(const float* __restrict__ pt, int ld_pt,
                                                                 const float* __restrict__ centers, int ld_c, int K,
                                                                 float min_temp, float max_temp, int64_t rows,
                                                                 float* __restrict__ depth) {
  constexpr int PITCH = KMAX + 1;  // (odd pitch: the 64 lanes of a wave read 64 different banks)
  __shared__ float ctile[256 * PITCH];
  __shared__ float lbs[KMAX];
  const float p_eps = 1e-4f, eps = 1e-4f, sb = 1e-7f;  // ConditionalLogBinomial.p_eps, LogBinomial eps, log_binom eps
  const float n = (float)(K - 1) + sb;
  const float nlogn = n * logf(n);
  if ((int)threadIdx.x < K) {
    const float kk = (float)threadIdx.x + sb;
    lbs[threadIdx.x] = nlogn - kk * logf(kk) - (n - kk) * logf(n - kk + sb);
  }
  for (int64_t r0 = (int64_t)blockIdx.x * 256; r0 < rows; r0 += (int64_t)gridDim.x * 256) {
    __syncthreads();  // (lbs written / the previous tile consumed)
    for (int idx = threadIdx.x; idx < 256 * K; idx += 256) {  // coalesced: consecutive threads walk consecutive bins of a row
      const int rl = idx / K, k = idx - rl * K;
      const int64_t r = r0 + rl;
      ctile[rl * PITCH + k] = r < rows ? centers[r * ld_c + k] : 0.f;
    }
    __syncthreads();
    const int64_t r = r0 + threadIdx.x;
    if (r >= rows) continue;
    const float* q = pt + r * ld_pt;
    float p0 = q[0] + p_eps, p1 = q[1] + p_eps, t0 = q[2] + p_eps, t1 = q[3] + p_eps;
    float p = p0 / (p0 + p1);
    float t = (max_temp - min_temp) * (t0 / (t0 + t1)) + min_temp;
    float omp = fminf(fmaxf(1.0f - p, eps), 1.0f);
    p = fminf(fmaxf(p, eps), 1.0f);
    const float lp = logf(p), lomp = logf(omp);
    const float* c = ctile + threadIdx.x * PITCH;
    // pass 1: max logit; pass 2: softmax-weighted sum
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) {
      float y = (lbs[k] + (float)k * lp + (float)(K - 1 - k) * lomp) / t;
      mx = fmaxf(mx, y);
    }
    float den = 0.f, num = 0.f;
    for (int k = 0; k < K; ++k) {
      float y = (lbs[k] + (float)k * lp + (float)(K - 1 - k) * lomp) / t;
      float e = expf(y - mx);
      den += e;
      num += e * c[k];
    }
    depth[r] = num / den;
  }
}

// any number of bins (no LDS tile): one thread per pixel
__global__ void __launch_bounds__(256) zoe_logbinom_depth_generic_kernel(const float* __restrict__ pt, int ld_pt,
                                                                         const float* __restrict__ centers, int ld_c, int K,
                                                                         float min_temp, float max_temp, int64_t rows,
                                                                         float* __restrict__ depth) {
  const float p_eps = 1e-4f, eps = 1e-4f, sb = 1e-7f;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
    const float* q = pt + r * ld_pt;
    float p0 = q[0] + p_eps, p1 = q[1] + p_eps, t0 = q[2] + p_eps, t1 = q[3] + p_eps;
    float p = p0 / (p0 + p1);
    float t = (max_temp - min_temp) * (t0 / (t0 + t1)) + min_temp;
    float omp = fminf(fmaxf(1.0f - p, eps), 1.0f);
    p = fminf(fmaxf(p, eps), 1.0f);
    const float lp = logf(p), lomp = logf(omp);
    const float n = (float)(K - 1) + sb;
    const float nlogn = n * logf(n);
    const float* c = centers + r * ld_c;
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) {
      float kk = (float)k + sb;
      float lb = nlogn - kk * logf(kk) - (n - kk) * logf(n - kk + sb);
      float y = (lb + (float)k * lp + (float)(K - 1 - k) * lomp) / t;
      mx = fmaxf(mx, y);
    }
    float den = 0.f, num = 0.f;
    for (int k = 0; k < K; ++k) {
      float kk = (float)k + sb;
      float lb = nlogn - kk * logf(kk) - (n - kk) * logf(n - kk + sb);
      float y = (lb + (float)k * lp + (float)(K - 1 - k) * lomp) / t;
      float e = expf(y - mx);
      den += e;
      num += e * c[k];
    }
    depth[r] = num / den;
  }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace prv2

using namespace prv2;

extern "C" int prv2_abi_version(void) { return PRV2_ABI_VERSION; }
extern "C" const char* prv2_last_error(void) { return prv2::g_err; }
extern "C" const char* prv2_last_kernel(void) { return prv2::g_kernel; }

extern "C" int prv2_layernorm(const float* x, int64_t rows, int32_t c, int32_t ldx, const float* weight,
                              const float* bias, float eps, int32_t act, float* y, int32_t ldy, void* stream) {
  PRV2_REQUIRE(x && y && weight && bias, "layernorm: null pointer");
  PRV2_REQUIRE(rows > 0 && c > 0 && ldx >= c && ldy >= c, "layernorm: bad geometry rows=%lld c=%d", (long long)rows, c);
  int grid = (int)(cdiv(rows, 4) < 8192 ? cdiv(rows, 4) : 8192);
  hipStream_t s = (hipStream_t)stream;
  bool vec = (c % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && aligned16(x) && aligned16(y) && aligned16(weight) &&
             aligned16(bias) && c <= 2048;
  if (!vec)
    hipLaunchKernelGGL(layernorm_scalar_kernel, dim3(grid), dim3(256), 0, s, x, rows, c, ldx, weight, bias, eps, act, y, ldy);
  else if (c <= 256)
    hipLaunchKernelGGL(layernorm_kernel<1>, dim3(grid), dim3(256), 0, s, x, rows, c, ldx, weight, bias, eps, act, y, ldy);
  else if (c <= 512)
    hipLaunchKernelGGL(layernorm_kernel<2>, dim3(grid), dim3(256), 0, s, x, rows, c, ldx, weight, bias, eps, act, y, ldy);
  else if (c <= 1024)
    hipLaunchKernelGGL(layernorm_kernel<4>, dim3(grid), dim3(256), 0, s, x, rows, c, ldx, weight, bias, eps, act, y, ldy);
  else
    hipLaunchKernelGGL(layernorm_kernel<8>, dim3(grid), dim3(256), 0, s, x, rows, c, ldx, weight, bias, eps, act, y, ldy);
  PRV2_LAUNCH_CHECK("layernorm");
  return 0;
}

extern "C" int prv2_layernorm_ss(const float* x, int64_t rows, int32_t c, int32_t ldx, const float* weight, const float* bias, float eps,
                                 void* y_ss, void* stream) {
  PRV2_REQUIRE(x && y_ss && weight && bias, "layernorm_ss: null pointer");
  PRV2_REQUIRE(rows > 0 && c > 0 && c % 32 == 0 && c <= 2048 && ldx >= c && ldx % 4 == 0 && aligned16(x) && aligned16(y_ss) &&
                   aligned16(weight) && aligned16(bias),
               "layernorm_ss: c must be a multiple of 32 (<= 2048), rows 16-byte aligned (rows=%lld c=%d)", (long long)rows, c);
  int grid = (int)(cdiv(rows, 4) < 8192 ? cdiv(rows, 4) : 8192);
  hipStream_t s = (hipStream_t)stream;
  float* y = reinterpret_cast<float*>(y_ss);
  const int ldy = c * 4;  // bytes
  if (c <= 256) hipLaunchKernelGGL((layernorm_kernel<1, true>), dim3(grid), dim3(256), 0, s, x, rows, c, ldx, weight, bias, eps, 0, y, ldy);
  else if (c <= 512) hipLaunchKernelGGL((layernorm_kernel<2, true>), dim3(grid), dim3(256), 0, s, x, rows, c, ldx, weight, bias, eps, 0, y, ldy);
  else if (c <= 1024) hipLaunchKernelGGL((layernorm_kernel<4, true>), dim3(grid), dim3(256), 0, s, x, rows, c, ldx, weight, bias, eps, 0, y, ldy);
  else hipLaunchKernelGGL((layernorm_kernel<8, true>), dim3(grid), dim3(256), 0, s, x, rows, c, ldx, weight, bias, eps, 0, y, ldy);
  PRV2_LAUNCH_CHECK("layernorm_ss");
  return 0;
}

extern "C" int prv2_patchify(const float* img, int32_t b, int32_t gh, int32_t gw, int32_t p, int32_t ldi, float* rows,
                             int32_t ldo, void* stream) {
  PRV2_REQUIRE(img && rows && b > 0 && gh > 0 && gw > 0 && p > 0 && ldi >= 3 && ldo >= p * p * 3, "patchify: bad arguments");
  int64_t total = (int64_t)b * gh * gw * ldo;
  hipLaunchKernelGGL(patchify_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, img, b, gh, gw, p,
                     ldi, rows, ldo);
  PRV2_LAUNCH_CHECK("patchify");
  return 0;
}

extern "C" int prv2_assemble_tokens(const float* emb, const float* cls, const float* pos, int32_t b, int32_t np,
                                    int32_t dim, float* tokens, void* stream) {
  PRV2_REQUIRE(emb && cls && pos && tokens && b > 0 && np > 0 && dim > 0, "assemble_tokens: bad arguments");
  int64_t total = (int64_t)b * (np + 1) * dim;
  hipLaunchKernelGGL(assemble_tokens_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, emb, cls, pos,
                     b, np, dim, tokens);
  PRV2_LAUNCH_CHECK("assemble_tokens");
  return 0;
}

extern "C" int prv2_conv2d_cout1(const float* x, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t ldx,
                                 const float* wgt, int32_t k, const float* bias, int32_t act, float scale,
                                 const float* res, int32_t clamp0, float* y, void* stream) {
  PRV2_REQUIRE(x && wgt && y, "conv2d_cout1: null pointer");
  PRV2_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && ldx >= cin && (k == 1 || k == 3), "conv2d_cout1: bad geometry");
  PRV2_REQUIRE(ldx % 4 == 0 && aligned16(x), "conv2d_cout1: x must be 16-byte aligned with ldx %% 4 == 0");
  int64_t total = (int64_t)n * h * w;
  if (k == 1) {
    PRV2_REQUIRE((size_t)cin * 4 <= 64 * 1024, "conv2d_cout1: cin too large");
    hipLaunchKernelGGL(conv_cout1_k1_kernel, dim3(flat_grid(total, 256)), dim3(256), (size_t)cin * 4, (hipStream_t)stream,
                       x, total, cin, ldx, wgt, bias, act, scale, res, clamp0, y);
  } else {
    const int tiles = n * (int)cdiv(h, C1_TH) * (int)cdiv(w, C1_TW);
    hipLaunchKernelGGL(conv_cout1_k3_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, x, n, h, w, cin, ldx, wgt,
                       act, scale, bias, res, clamp0, y);
  }
  PRV2_LAUNCH_CHECK("conv2d_cout1");
  return 0;
}

extern "C" int prv2_dwconv2d_ex(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ldx, const float* wgt,
                                const float* bias, int32_t k, int32_t stride, int32_t act, int32_t same_pad, float* y,
                                int32_t ldy, void* stream) {
  PRV2_REQUIRE(x && wgt && y, "dwconv2d: null pointer");
  PRV2_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= c && ldy >= c &&
                   (k == 3 || k == 5 || k == 7) && (stride == 1 || stride == 2),
               "dwconv2d: bad geometry c=%d k=%d stride=%d", c, k, stride);
  PRV2_REQUIRE(act >= PRV2_ACT_NONE && act <= PRV2_ACT_SILU, "dwconv2d: unknown activation %d", act);
  PRV2_REQUIRE(aligned16(x) && aligned16(y) && aligned16(wgt), "dwconv2d: pointers must be 16-byte aligned");
  int oh = (h + 2 * (k / 2) - k) / stride + 1, ow = (w + 2 * (k / 2) - k) / stride + 1;
  int pad_y = k / 2, pad_x = k / 2;
  if (same_pad) {  // timm Conv2dSame / TensorFlow "SAME" (see prv2_conv_desc.same_pad)
    oh = (h + stride - 1) / stride;
    ow = (w + stride - 1) / stride;
    const int ty = (oh - 1) * stride + k - h, tx = (ow - 1) * stride + k - w;
    pad_y = (ty > 0 ? ty : 0) / 2;
    pad_x = (tx > 0 ? tx : 0) / 2;
  }
  int64_t total = (int64_t)n * oh * ow * (c / 4);
  if (stride == 1 && w >= 8 && (((w + 7) / 8) * 8 - w) * 8 <= w) {  // strip kernel: 8 output pixels per thread, when the last
    // strip of a row wastes <= 1/8 of the work (stride 1: "SAME" == symmetric k/2)
    constexpr int PX = 8;
    const int64_t threads = (int64_t)n * h * ((w + PX - 1) / PX) * (c / 4);
    PRV2_REQUIRE((threads + 255) / 256 < (1LL << 31), "dwconv2d: too large");
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (k == 3) hipLaunchKernelGGL((dwconv_strip_kernel<3, PX>), grid, dim3(256), 0, (hipStream_t)stream, x, n, h, w, c, ldx, wgt, bias, act, y, ldy);
    else if (k == 5) hipLaunchKernelGGL((dwconv_strip_kernel<5, PX>), grid, dim3(256), 0, (hipStream_t)stream, x, n, h, w, c, ldx, wgt, bias, act, y, ldy);
    else hipLaunchKernelGGL((dwconv_strip_kernel<7, PX>), grid, dim3(256), 0, (hipStream_t)stream, x, n, h, w, c, ldx, wgt, bias, act, y, ldy);
    PRV2_LAUNCH_CHECK("dwconv2d");
    return 0;
  }
  hipLaunchKernelGGL(dwconv_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, n, h, w, c, ldx, wgt,
                     bias, k, stride, act, pad_y, pad_x, oh, ow, y, ldy);
  PRV2_LAUNCH_CHECK("dwconv2d");
  return 0;
}

extern "C" int prv2_dwconv2d(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ldx, const float* wgt,
                             const float* bias, int32_t k, int32_t stride, int32_t relu, float* y, int32_t ldy,
                             void* stream) {
  return prv2_dwconv2d_ex(x, n, h, w, c, ldx, wgt, bias, k, stride, relu ? PRV2_ACT_RELU : PRV2_ACT_NONE, 0, y, ldy, stream);
}

static int avgpool_chunks(int64_t hw) {
  int64_t ch = (hw + 511) / 512;
  return (int)(ch < 1 ? 1 : (ch > 128 ? 128 : ch));
}

extern "C" int64_t prv2_global_avgpool_workspace_floats(int32_t n, int64_t hw, int32_t c) {
  return (int64_t)n * avgpool_chunks(hw) * c;
}

extern "C" int prv2_global_avgpool(const float* x, int32_t n, int64_t hw, int32_t c, int32_t ldx, float* out, float* workspace,
                                   void* stream) {
  PRV2_REQUIRE(x && out && workspace && n > 0 && hw > 0 && c > 0 && c % 4 == 0 && ldx % 4 == 0 && ldx >= c,
               "global_avgpool: bad arguments");
  PRV2_REQUIRE(aligned16(x) && aligned16(workspace), "global_avgpool: pointers must be 16-byte aligned");
  const int cg = c / 4, chunks = avgpool_chunks(hw);
  int cgw_log2 = 0;
  while ((1 << cgw_log2) < cg && cgw_log2 < 6) ++cgw_log2;
  const int chunk_len = (int)((hw + chunks - 1) / chunks);
  hipLaunchKernelGGL(avgpool_partial_kernel, dim3((cg + (1 << cgw_log2) - 1) >> cgw_log2, chunks, n), dim3(256), 0, (hipStream_t)stream, x,
                     hw, c, ldx, cgw_log2, chunk_len, workspace);
  PRV2_LAUNCH_CHECK("global_avgpool");
  hipLaunchKernelGGL(avgpool_final_kernel, dim3((n * c + 255) / 256), dim3(256), 0, (hipStream_t)stream, workspace, n, c, chunks,
                     1.0f / (float)hw, out);
  PRV2_LAUNCH_CHECK("global_avgpool");
  return 0;
}

extern "C" int prv2_se_gate(const float* mean, int32_t n, int32_t c, const float* w1, const float* b1, int32_t cse, const float* w2t,
                            const float* b2, float* g, float* workspace, void* stream) {
  PRV2_REQUIRE(mean && w1 && w2t && g && workspace && n > 0 && c > 0 && cse > 0, "se_gate: bad arguments");
  hipLaunchKernelGGL(se_reduce_kernel, dim3(cse, n), dim3(64), 0, (hipStream_t)stream, mean, c, w1, b1, cse, workspace);
  PRV2_LAUNCH_CHECK("se_gate");
  hipLaunchKernelGGL(se_expand_kernel, dim3((c + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, workspace, c, cse, w2t, b2, g);
  PRV2_LAUNCH_CHECK("se_gate");
  return 0;
}

extern "C" int prv2_channel_scale(float* x, int32_t n, int64_t hw, int32_t c, int32_t ldx, const float* s, void* stream) {
  PRV2_REQUIRE(x && s && n > 0 && hw > 0 && c > 0 && c % 4 == 0 && ldx % 4 == 0 && ldx >= c, "channel_scale: bad arguments");
  PRV2_REQUIRE(aligned16(x) && aligned16(s), "channel_scale: pointers must be 16-byte aligned");
  const int64_t total = (int64_t)n * hw * (c / 4);
  hipLaunchKernelGGL(channel_scale_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, hw, c, ldx, s, total);
  PRV2_LAUNCH_CHECK("channel_scale");
  return 0;
}

extern "C" int prv2_add(const float* a, int32_t lda, const float* b, int32_t ldb, int64_t rows, int32_t c, float* y,
                        int32_t ldy, void* stream) {
  PRV2_REQUIRE(a && b && y && rows > 0 && c > 0 && lda >= c && ldb >= c && ldy >= c, "add: bad arguments");
  hipLaunchKernelGGL(add_kernel, dim3(flat_grid(rows * c, 256)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, rows, c,
                     y, ldy);
  PRV2_LAUNCH_CHECK("add");
  return 0;
}

extern "C" int prv2_zero_pad_channels(float* y, int64_t rows, int32_t c, int32_t ld, void* stream) {
  PRV2_REQUIRE(y && rows > 0 && c > 0 && ld >= c, "zero_pad_channels: bad arguments");
  if (ld == c) return 0;
  hipLaunchKernelGGL(zero_pad_kernel, dim3(flat_grid(rows, 256)), dim3(256), 0, (hipStream_t)stream, y, rows, c, ld);
  PRV2_LAUNCH_CHECK("zero_pad_channels");
  return 0;
}

extern "C" int prv2_zoe_attractor(const float* attr, int32_t ld_attr, int32_t n_attr, const float* bins, int32_t ld_bins,
                                  int32_t n_bins, float alpha, int64_t rows, float* out, int32_t ld_out, void* stream) {
  PRV2_REQUIRE(attr && bins && out && rows > 0 && n_attr > 0 && n_bins > 0 && ld_attr >= n_attr && ld_bins >= n_bins &&
                   ld_out >= n_bins, "zoe_attractor: bad arguments");
  hipLaunchKernelGGL(zoe_attractor_kernel, dim3(flat_grid(rows * n_bins, 256)), dim3(256), 0, (hipStream_t)stream, attr,
                     ld_attr, n_attr, bins, ld_bins, n_bins, alpha, rows, out, ld_out);
  PRV2_LAUNCH_CHECK("zoe_attractor");
  return 0;
}

extern "C" int prv2_zoe_logbinom_depth(const float* pt, int32_t ld_pt, const float* centers, int32_t ld_c, int32_t n_bins,
                                       float min_temp, float max_temp, int64_t rows, float* depth, void* stream) {
  PRV2_REQUIRE(pt && centers && depth && rows > 0 && n_bins > 1 && ld_pt >= 4 && ld_c >= n_bins,
               "zoe_logbinom_depth: bad arguments");
  if (n_bins <= 64)
    hipLaunchKernelGGL(zoe_logbinom_depth_kernel<64>, dim3(flat_grid(rows, 256)), dim3(256), 0, (hipStream_t)stream, pt, ld_pt, centers, ld_c, n_bins, min_temp,
                       max_temp, rows, depth);
  else
    hipLaunchKernelGGL(zoe_logbinom_depth_generic_kernel, dim3(flat_grid(rows, 256)), dim3(256), 0, (hipStream_t)stream, pt, ld_pt, centers, ld_c, n_bins,
                       min_temp, max_temp, rows, depth);
  PRV2_LAUNCH_CHECK("zoe_logbinom_depth");
  return 0;
}

PRV2_NO_PACKED_FP32_END
