// Overlap blend: the reference's RunningAverageMap (estimator/models/utils.py:22-49) kept on
// the device.  The running mean is ORDER dependent and has a ct>0 predicate (pass-1 pixels with
// zero Gaussian weight keep their pasted value), so it cannot be a sum(w*p)/sum(w).  To stay
// order-exact AND parallel, each thread owns one map pixel and walks the tile list in order
// (gather form): K is small (<= a few dozen), the maps are HBM-resident, one read + one write
// of the touched pixels per call.
#include "common.h"

PRV2_NO_PACKED_FP32_BEGIN  // (common.h)

namespace prv2 {

template <bool PASTE>
__global__ void __launch_bounds__(256) blend_kernel(float* __restrict__ avg, float* __restrict__ cnt, int MH, int MW,
                                                    const float* __restrict__ pred, int ph, int pw,
                                                    const float* __restrict__ mask, const int* __restrict__ tiles, int K,
                                                    int th, int tw, float sy, float sx, int y_lo, int x_lo, int bh,
                                                    int bw) {
  // threads cover the bounding box [y_lo, y_lo+bh) x [x_lo, x_lo+bw) of all tiles
  int64_t total = (int64_t)bh * bw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int x = x_lo + (int)(idx % bw);
    int y = y_lo + (int)(idx / bw);
    int64_t o = (int64_t)y * MW + x;
    float a = avg[o], c = cnt[o];
    bool touched = false;
    for (int k = 0; k < K; ++k) {
      int h0 = tiles[2 * k], w0 = tiles[2 * k + 1];
      int ty = y - h0, tx = x - w0;
      if (ty < 0 || ty >= th || tx < 0 || tx >= tw) continue;
      float ct = mask[(int64_t)ty * tw + tx];
      int py = (ph == th) ? ty : nearest_src(ty, sy, ph);
      int px = (pw == tw) ? tx : nearest_src(tx, sx, pw);
      float p = pred[((int64_t)k * ph + py) * pw + px];
      if (PASTE) {
        // count_map[tile] = blur_mask ; pred_depth[tile] = temp_depth   (baseline_pretrain.py:352-355)
        a = p;
        c = ct;
        touched = true;
      } else if (ct > 0.f) {
        // avg = (p*ct + count*avg) / (count + ct) ; count += ct          (utils.py:31-36)
        float num = p * ct + c * a;
        float den = c + ct;
        a = num / den;
        c = den;
        touched = true;
      }
    }
    if (touched) {
      avg[o] = a;
      cnt[o] = c;
    }
  }
}

__global__ void __launch_bounds__(256) blend_resize_kernel(const float* __restrict__ avg, const float* __restrict__ cnt,
                                                           int H, int W, float* __restrict__ avg_o,
                                                           float* __restrict__ cnt_o, int oh, int ow, float ny, float nx,
                                                           float by, float bx) {
  int64_t total = (int64_t)oh * ow;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int ox = (int)(idx % ow), oy = (int)(idx / ow);
    avg_o[idx] = avg[(int64_t)nearest_src(oy, ny, H) * W + nearest_src(ox, nx, W)];
    AxisTap ty = ac_tap(oy, by, H), tx = ac_tap(ox, bx, W);
    float v00 = cnt[(int64_t)ty.i0 * W + tx.i0], v01 = cnt[(int64_t)ty.i0 * W + tx.i1];
    float v10 = cnt[(int64_t)ty.i1 * W + tx.i0], v11 = cnt[(int64_t)ty.i1 * W + tx.i1];
    cnt_o[idx] = ty.w0 * (tx.w0 * v00 + tx.w1 * v01) + ty.w1 * (tx.w0 * v10 + tx.w1 * v11);
  }
}

template <bool PASTE>
static int blend_launch(const char* name, float* avg, float* cnt, int MH, int MW, const float* pred, int ph, int pw,
                        const float* mask, const int* tiles, int K, int th, int tw, void* stream) {
  PRV2_REQUIRE(avg && cnt && pred && mask && tiles, "%s: null pointer", name);
  PRV2_REQUIRE(K > 0 && th > 0 && tw > 0 && ph > 0 && pw > 0 && th <= MH && tw <= MW, "%s: bad geometry", name);
  // tile coordinates live on the device; cover the whole map (cheap: one pass over <= 33 MB maps)
  int64_t total = (int64_t)MH * MW;
  hipLaunchKernelGGL(blend_kernel<PASTE>, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, avg, cnt, MH, MW,
                     pred, ph, pw, mask, tiles, K, th, tw, (float)ph / (float)th, (float)pw / (float)tw, 0, 0, MH, MW);
  PRV2_LAUNCH_CHECK(name);
  return 0;
}

}  // namespace prv2

using namespace prv2;

extern "C" int prv2_blend_paste(float* avg, float* cnt, int32_t map_h, int32_t map_w, const float* pred, int32_t ph,
                                int32_t pw, const float* mask, const int32_t* tiles, int32_t k, int32_t th, int32_t tw,
                                void* stream) {
  return blend_launch<true>("blend_paste", avg, cnt, map_h, map_w, pred, ph, pw, mask, tiles, k, th, tw, stream);
}

extern "C" int prv2_blend_update(float* avg, float* cnt, int32_t map_h, int32_t map_w, const float* pred, int32_t ph,
                                 int32_t pw, const float* mask, const int32_t* tiles, int32_t k, int32_t th, int32_t tw,
                                 void* stream) {
  return blend_launch<false>("blend_update", avg, cnt, map_h, map_w, pred, ph, pw, mask, tiles, k, th, tw, stream);
}

extern "C" int prv2_blend_resize(const float* avg, const float* cnt, int32_t h, int32_t w, float* avg_out,
                                 float* cnt_out, int32_t oh, int32_t ow, void* stream) {
  PRV2_REQUIRE(avg && cnt && avg_out && cnt_out && h > 0 && w > 0 && oh > 0 && ow > 0, "blend_resize: bad arguments");
  int64_t total = (int64_t)oh * ow;
  hipLaunchKernelGGL(blend_resize_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, avg, cnt, h, w,
                     avg_out, cnt_out, oh, ow, (float)h / (float)oh, (float)w / (float)ow, ac_scale(h, oh),
                     ac_scale(w, ow));
  PRV2_LAUNCH_CHECK("blend_resize");
  return 0;
}

PRV2_NO_PACKED_FP32_END
