// PyTorch-ROCm custom ops over the C ABI (include/prv2.h):  torch.ops.prv2.*
//
// The reference is pure PyTorch: what its hot path dispatches are torch ops.  This file registers the
// replacements under TORCH_LIBRARY(prv2, ...) so that they take / return at::Tensor, run on the CURRENT HIP stream of
// the tensor's device, allocate outputs through torch's caching allocator (or fill a caller's ``out`` view), raise
// through TORCH_CHECK (-> Python RuntimeError) and keep no state.  Activations are NHWC fp32; a tensor may be a channel
// slice of a wider buffer (stride(-1) == 1, stride(-2) == ld), which is how a torch.cat is written in place.
// Host-only translation unit: all device code lives behind the C ABI in libprv2_hip.so.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <torch/library.h>

#include <algorithm>
#include <vector>

#include "../../include/prv2.h"

namespace {

using at::Tensor;
using c10::optional;

// device guard + the stream torch considers current on that device.  PyTorch-ROCm presents HIP devices under the "cuda"
// device type ("masquerading"): the guard is the generic one, the stream comes from the masquerading accessor.
struct Launch {
  c10::DeviceGuard guard;
  void* stream;
  explicit Launch(const Tensor& t)
      : guard(t.device()), stream((void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream()) {}
};

void ok(int code, const char* what) { TORCH_CHECK(code == 0, "prv2::", what, " failed (code ", code, "): ", prv2_last_error()); }

void dev_f32(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), "prv2: ", name, " must be a GPU tensor (there is no CPU path)");
  TORCH_CHECK(t.scalar_type() == at::kFloat, "prv2: ", name, " must be float32");
}

// NHWC view [n, h, w, c] with unit channel stride, pixel stride ld and dense rows / images on top of it
int64_t nhwc_ld(const Tensor& t, const char* name, bool free_batch_stride = false) {
  dev_f32(t, name);
  TORCH_CHECK(t.dim() == 4, "prv2: ", name, " must be NHWC [n, h, w, c]");
  const int64_t ld = t.stride(2);
  TORCH_CHECK(t.stride(3) == 1 && ld >= t.size(3) && t.stride(1) == t.size(2) * ld &&
                  (t.size(0) == 1 || free_batch_stride || t.stride(0) == t.size(1) * t.size(2) * ld),
              "prv2: ", name, " must be an NHWC tensor or a channel slice of one (strides ", t.strides(), ")");
  return ld;
}

const float* opt_ptr(const optional<Tensor>& t, const char* name, int64_t numel = -1) {
  if (!t.has_value()) return nullptr;
  dev_f32(*t, name);
  TORCH_CHECK(t->is_contiguous(), "prv2: ", name, " must be contiguous");
  TORCH_CHECK(numel < 0 || t->numel() == numel, "prv2: ", name, " has ", t->numel(), " elements, expected ", numel);
  return t->data_ptr<float>();
}

Tensor alloc_nhwc(const Tensor& like, int64_t n, int64_t h, int64_t w, int64_t c) {
  const int64_t ld = (c + 3) / 4 * 4;  // pad channels must be finite: zero the whole buffer when there are any
  Tensor buf = ld == c ? at::empty({n, h, w, ld}, like.options()) : at::zeros({n, h, w, ld}, like.options());
  return ld == c ? buf : buf.narrow(3, 0, c);
}

Tensor out_or_alloc(const optional<Tensor>& out, const Tensor& like, int64_t n, int64_t h, int64_t w, int64_t c, const char* what) {
  if (!out.has_value()) return alloc_nhwc(like, n, h, w, c);
  TORCH_CHECK(out->dim() == 4 && out->size(0) == n && out->size(1) == h && out->size(2) == w && out->size(3) == c, "prv2::", what,
              ": out has shape ", out->sizes(), ", expected [", n, ", ", h, ", ", w, ", ", c, "]");
  return *out;
}

// ---------------------------------------------------------------------------------------------------------------------
Tensor pack_conv_weight(const Tensor& w, const optional<Tensor>& bn_scale, int64_t convt_k, int64_t prec) {
  dev_f32(w, "weight");
  TORCH_CHECK(w.dim() == 4 || w.dim() == 2, "prv2::pack_conv_weight: weight must be [cout, cin, kh, kw] or [cout, cin]");
  Tensor wc = w.contiguous();
  const int64_t kh = w.dim() == 4 ? w.size(2) : 1, kw = w.dim() == 4 ? w.size(3) : 1;
  const int64_t cout = convt_k ? w.size(1) : w.size(0), cin = convt_k ? w.size(0) : w.size(1);
  const int64_t bytes = prv2_packed_weight_bytes((int)cout, (int)cin, (int)kh, (int)kw, (int)convt_k, (int)prec);
  TORCH_CHECK(bytes > 0, "prv2::pack_conv_weight: ", prv2_last_error());
  Tensor packed = at::empty({bytes / 4}, w.options());
  Launch L(w);
  ok(prv2_pack_conv_weight(wc.data_ptr<float>(), opt_ptr(bn_scale, "bn_scale", cout), packed.data_ptr(), (int)cout, (int)cin, (int)kh, (int)kw,
                           (int)convt_k, (int)prec, L.stream), "pack_conv_weight");
  return packed;
}

// nn.Conv2d / nn.Linear (kh = kw = 1 over an [1, M, 1, K] view) / nn.ConvTranspose2d(k == stride) with the fused epilogue of
// include/prv2.h::prv2_conv2d
Tensor conv2d(const Tensor& x_in, const Tensor& w_packed, const optional<Tensor>& bias, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
              int64_t pad, int64_t act, bool relu_in, const optional<Tensor>& ln_weight, const optional<Tensor>& ln_bias,
              const optional<Tensor>& gamma, const optional<Tensor>& mul, const optional<Tensor>& res, const optional<Tensor>& res2,
              int64_t convt_k, int64_t prec, double ln_eps, bool same_pad, const optional<Tensor>& out, int64_t fmt, bool force_generic,
              int64_t part) {
  Tensor x = x_in;
  if (x.is_cuda() && x.dim() == 4 && x.scalar_type() == at::kFloat && (x.stride(2) % 4 != 0 || (reinterpret_cast<uintptr_t>(x.data_ptr()) & 15))) {
    // the kernels read 16-byte groups: a dense tensor whose channel count is not a multiple of 4 (34, 66, 98 ... channels) is
    // copied once into a buffer with zeroed pad channels (the host mirror allocates its concat buffers padded from the start)
    Tensor padded = alloc_nhwc(x, x.size(0), x.size(1), x.size(2), x.size(3));
    padded.copy_(x);
    x = padded;
  }
  const int64_t ldx = nhwc_ld(x, "x", true);  // (an image stride of its own: token maps read as NHWC images, DPT head)
  dev_f32(w_packed, "w_packed");
  if (convt_k) TORCH_CHECK(kh == convt_k && kw == convt_k && pad == 0, "prv2::conv2d: convt_k needs kh == kw == convt_k and pad 0");
  const int64_t n = x.size(0), h = x.size(1), w = x.size(2), cin = x.size(3);
  int64_t oh, ow;
  if (convt_k) { oh = h * convt_k; ow = w * convt_k; }
  else if (same_pad) { oh = (h + stride - 1) / stride; ow = (w + stride - 1) / stride; }
  else { oh = (h + 2 * pad - kh) / stride + 1; ow = (w + 2 * pad - kw) / stride + 1; }
  TORCH_CHECK(oh > 0 && ow > 0, "prv2::conv2d: empty output");
  TORCH_CHECK(w_packed.numel() * 4 == prv2_packed_weight_bytes((int)cout, (int)cin, (int)kh, (int)kw, (int)convt_k, (int)prec),
              "prv2::conv2d: w_packed does not match (cout, cin, kh, kw, convt_k, prec) = (", cout, ", ", cin, ", ", kh, ", ", kw, ", ", convt_k, ", ", prec, ")");
  TORCH_CHECK(ln_weight.has_value() == ln_bias.has_value(), "prv2::conv2d: ln_weight and ln_bias go together");
  Tensor y = out_or_alloc(out, x, n, oh, ow, cout, "conv2d");
  prv2_conv_desc d = {};
  d.n = (int)n; d.h = (int)h; d.w = (int)w; d.cin = (int)cin; d.cout = (int)cout; d.kh = (int)kh; d.kw = (int)kw;
  d.stride = (int)(convt_k ? convt_k : stride); d.pad = (int)pad; d.ldx = (int)ldx; d.ldy = (int)nhwc_ld(y, "out");
  d.relu_in = relu_in; d.act = (int)act; d.convt_k = (int)convt_k; d.prec = (int)prec; d.ln_eps = (float)ln_eps; d.same_pad = same_pad;
  d.fmt = (int)fmt; d.force_generic = force_generic; d.part = (int)part;
  // (also for one image: which kernel runs a layer depends on the layer's layout, never on the batch -- a token map read through an
  // image stride takes the same kernel for one tile as for forty)
  if (x.stride(0) != h * w * ldx) d.x_bstride = x.stride(0);
  auto aux = [&](const optional<Tensor>& t, const char* name, int32_t& ld) -> const float* {
    if (!t.has_value()) return nullptr;
    ld = (int32_t)nhwc_ld(*t, name);
    TORCH_CHECK(t->sizes() == y.sizes(), "prv2::conv2d: ", name, " must have the output's shape");
    return t->data_ptr<float>();
  };
  const float* pm = aux(mul, "mul", d.ld_mul);
  const float* pr = aux(res, "res", d.ld_res);
  const float* pr2 = aux(res2, "res2", d.ld_res2);
  Launch L(x);
  ok(prv2_conv2d(&d, x.data_ptr<float>(), w_packed.data_ptr(), opt_ptr(bias, "bias", cout), opt_ptr(ln_weight, "ln_weight", cout),
                 opt_ptr(ln_bias, "ln_bias", cout), opt_ptr(gamma, "gamma", cout), pm, pr, pr2, y.data_ptr<float>(), L.stream), "conv2d");
  return y;
}

// include/prv2.h::prv2_conv2d_ups: 3x3 conv over the virtual concat [bilinear_align_corners(u -> oh x ow) | x[..., u.c:]] -- the upsample
// of UpSample.forward_hardcode (fusion_model.py:15-24) / of output_conv1's input (bi_directional_fusion_model.py:139-142,201) formed
// inside the conv's tile loader.  x: NHWC [n, oh, ow, cin] whose first u.size(3) channels are not read, or None when every input
// channel comes from u (oh, ow give the output size either way).
Tensor conv3x3_ups(const optional<Tensor>& x_in, const Tensor& u, const Tensor& w_packed, const optional<Tensor>& bias, int64_t cout,
                   int64_t oh, int64_t ow, int64_t act, const optional<Tensor>& ln_weight, const optional<Tensor>& ln_bias,
                   const optional<Tensor>& res, int64_t prec, double ln_eps, const optional<Tensor>& out) {
  const int64_t ldu = nhwc_ld(u, "u");
  const int64_t n = u.size(0), cu = u.size(3);
  const Tensor& x = x_in.has_value() ? *x_in : u;
  const int64_t ldx = x_in.has_value() ? nhwc_ld(x, "x") : ldu;
  const int64_t cin = x_in.has_value() ? x.size(3) : cu;
  if (x_in.has_value()) TORCH_CHECK(x.size(0) == n && x.size(1) == oh && x.size(2) == ow, "prv2::conv3x3_ups: x must be [n, oh, ow, cin]");
  dev_f32(w_packed, "w_packed");
  TORCH_CHECK(w_packed.numel() * 4 == prv2_packed_weight_bytes((int)cout, (int)cin, 3, 3, 0, (int)prec), "prv2::conv3x3_ups: w_packed does not match (cout, cin, prec)");
  TORCH_CHECK(ln_weight.has_value() == ln_bias.has_value(), "prv2::conv3x3_ups: ln_weight and ln_bias go together");
  Tensor y = out_or_alloc(out, u, n, oh, ow, cout, "conv3x3_ups");
  prv2_conv_desc d = {};
  d.n = (int)n; d.h = (int)oh; d.w = (int)ow; d.cin = (int)cin; d.cout = (int)cout; d.kh = 3; d.kw = 3; d.stride = 1; d.pad = 1;
  d.ldx = (int)ldx; d.ldy = (int)nhwc_ld(y, "out"); d.act = (int)act; d.prec = (int)prec; d.ln_eps = (float)ln_eps;
  const float* pr = nullptr;
  if (res.has_value()) {
    d.ld_res = (int32_t)nhwc_ld(*res, "res");
    TORCH_CHECK(res->sizes() == y.sizes(), "prv2::conv3x3_ups: res must have the output's shape");
    pr = res->data_ptr<float>();
  }
  prv2_ups_src us = {};
  us.x = u.data_ptr<float>(); us.h = (int)u.size(1); us.w = (int)u.size(2); us.ld = (int)ldu; us.channels = (int)cu; us.bstride = 0;
  TORCH_CHECK(prv2_conv2d_ups_supported(&d, &us), "prv2::conv3x3_ups: layer not covered (bf16 modes, cout > 64, width >= 24, u channels % 32 == 0, 16-byte aligned rows)");
  Launch L(u);
  ok(prv2_conv2d_ups(&d, x.data_ptr<float>(), &us, w_packed.data_ptr(), opt_ptr(bias, "bias", cout), opt_ptr(ln_weight, "ln_weight", cout),
                     opt_ptr(ln_bias, "ln_bias", cout), pr, y.data_ptr<float>(), L.stream), "conv3x3_ups");
  return y;
}

// include/prv2.h::prv2_upconv3x3: act(conv3x3(interpolate(u, (oh, ow), bilinear, align_corners=True)) + bias) computed at u's resolution
// (tap GEMMs on the low-resolution grid + a gather): output_conv1 (bi_directional_fusion_model.py:139-142,201), the x1 part of
// UpSample.forward_hardcode's first conv (fusion_model.py:15-24)
Tensor upconv3x3(const Tensor& u, const Tensor& w_packed, const optional<Tensor>& bias, int64_t cout, int64_t oh, int64_t ow, int64_t act, int64_t prec,
                 const optional<Tensor>& out, const optional<Tensor>& add) {
  const int64_t ldu = nhwc_ld(u, "u");
  const int64_t n = u.size(0), cu = u.size(3);
  dev_f32(w_packed, "w_packed");
  TORCH_CHECK(w_packed.numel() * 4 == prv2_packed_weight_bytes((int)cout, (int)cu, 3, 3, 0, (int)prec), "prv2::upconv3x3: w_packed does not match (cout, u channels, prec)");
  Tensor y = out_or_alloc(out, u, n, oh, ow, cout, "upconv3x3");
  prv2_ups_src us = {};
  us.x = u.data_ptr<float>(); us.h = (int)u.size(1); us.w = (int)u.size(2); us.ld = (int)ldu; us.channels = (int)cu; us.bstride = 0;
  TORCH_CHECK(prv2_upconv3x3_supported(&us, (int)n, (int)oh, (int)ow, (int)cout, (int)prec),
              "prv2::upconv3x3: layer not covered (bf16 modes, u channels % 32 == 0, output at least 2h-1 x 2w-1 of u)");
  const float* pa = nullptr;
  int ld_add = 0;
  if (add.has_value()) {
    TORCH_CHECK(add->sizes() == y.sizes(), "prv2::upconv3x3: add must have the output's shape");
    pa = add->data_ptr<float>();
    ld_add = (int)nhwc_ld(*add, "add");
  }
  Launch L(u);
  ok(prv2_upconv3x3(&us, w_packed.data_ptr(), opt_ptr(bias, "bias", cout), pa, ld_add, (int)n, (int)oh, (int)ow, (int)cout, (int)act, (int)prec,
                    y.data_ptr<float>(), (int)nhwc_ld(y, "out"), 0, L.stream), "upconv3x3");
  return y;
}

// include/prv2.h::prv2_pack_gate_weight / prv2_conv3x3_ln_gate: the GatedConvUnit tail (bi_directional_fusion_model.py:44-51,70-80)
Tensor pack_gate_weight(const Tensor& w) {
  dev_f32(w, "weight");
  const int64_t c = w.size(0);
  TORCH_CHECK(w.numel() == c * c && prv2_gate_weight_bytes((int)c) > 0, "prv2::pack_gate_weight: the fused gate is a C -> C 1x1 conv, C = 32, 128 or 256");
  Tensor wc = w.contiguous();
  Tensor packed = at::empty({prv2_gate_weight_bytes((int)c) / 4}, w.options());
  Launch L(w);
  ok(prv2_pack_gate_weight(wc.data_ptr<float>(), packed.data_ptr(), (int)c, (int)c, L.stream), "pack_gate_weight");
  return packed;
}

Tensor conv3x3_ln_gate(const Tensor& x, const Tensor& w_packed, const optional<Tensor>& bias, const Tensor& ln_weight, const Tensor& ln_bias,
                       const optional<Tensor>& gate_w_packed, const optional<Tensor>& gate_bias, const optional<Tensor>& mul,
                       const optional<Tensor>& res, int64_t act, bool relu_in, int64_t prec, double ln_eps, const optional<Tensor>& out,
                       const optional<Tensor>& pre, int64_t fmt) {
  const int64_t ldx = nhwc_ld(x, "x");
  dev_f32(w_packed, "w_packed");
  const int64_t n = x.size(0), h = x.size(1), w = x.size(2), cin = x.size(3), cout = ln_weight.numel();
  TORCH_CHECK(w_packed.numel() * 4 == prv2_packed_weight_bytes((int)cout, (int)cin, 3, 3, 0, (int)prec), "prv2::conv3x3_ln_gate: w_packed does not match a 3x3 ",
              cin, " -> ", cout, " conv in mode ", prec);
  Tensor y = out_or_alloc(out, x, n, h, w, cout, "conv3x3_ln_gate");
  prv2_conv_desc d = {};
  d.n = (int)n; d.h = (int)h; d.w = (int)w; d.cin = (int)cin; d.cout = (int)cout; d.kh = 3; d.kw = 3; d.stride = 1; d.pad = 1;
  d.ldx = (int)ldx; d.ldy = (int)nhwc_ld(y, "out"); d.relu_in = relu_in; d.act = (int)act; d.prec = (int)prec; d.ln_eps = (float)ln_eps;
  d.fmt = (int)fmt;
  TORCH_CHECK(prv2_conv3x3_ln_gate_supported(&d), "prv2::conv3x3_ln_gate: shape not covered (3x3 s1 p1, cout 256 with width >= 16 or 128 / 32 with width >= 24 and height >= 4, cin % 32 == 0, bf16 modes)");
  auto aux = [&](const optional<Tensor>& t, const char* name, int32_t& ld) -> const float* {
    if (!t.has_value()) return nullptr;
    ld = (int32_t)nhwc_ld(*t, name);
    TORCH_CHECK(t->sizes() == y.sizes(), "prv2::conv3x3_ln_gate: ", name, " must have the output's shape");
    return t->data_ptr<float>();
  };
  const float* pm = aux(mul, "mul", d.ld_mul);
  const float* pr = aux(res, "res", d.ld_res);
  if (gate_w_packed.has_value()) {
    dev_f32(*gate_w_packed, "gate_w_packed");
    TORCH_CHECK(gate_w_packed->numel() * 4 == prv2_gate_weight_bytes((int)cout), "prv2::conv3x3_ln_gate: gate_w_packed is not a pack_gate_weight image");
  }
  int32_t ld_pre = 0;
  const float* pp = aux(pre, "pre", ld_pre);
  Launch L(x);
  ok(prv2_conv3x3_ln_gate_pre(&d, x.data_ptr<float>(), w_packed.data_ptr(), opt_ptr(bias, "bias", cout), pp, ld_pre, opt_ptr(ln_weight, "ln_weight", cout),
                              opt_ptr(ln_bias, "ln_bias", cout), gate_w_packed.has_value() ? gate_w_packed->data_ptr() : nullptr,
                              opt_ptr(gate_bias, "gate_bias", cout), pm, pr, y.data_ptr<float>(), L.stream), "conv3x3_ln_gate");
  return y;
}

// shared by conv3x3_tail / conv3x3_pre: the descriptor of a 3x3 s1 p1 conv over x -> y
prv2_conv_desc desc3x3(const Tensor& x, const Tensor& y, int64_t cout, int64_t act, int64_t prec, double ln_eps) {
  prv2_conv_desc d = {};
  d.n = (int)x.size(0); d.h = (int)x.size(1); d.w = (int)x.size(2); d.cin = (int)x.size(3); d.cout = (int)cout; d.kh = 3; d.kw = 3; d.stride = 1; d.pad = 1;
  d.ldx = (int)nhwc_ld(x, "x"); d.ldy = (int)nhwc_ld(y, "out"); d.act = (int)act; d.prec = (int)prec; d.ln_eps = (float)ln_eps;
  return d;
}

// include/prv2.h::prv2_conv2d_tail: the conv that fills a fusion level's features also writes the [pred1 | pred2 | 0 | 0] tail behind them
// (fusion_model.py:91-118).  out: the [n, h, w, cout] slice of a buffer with at least 4 more channels per pixel.
void conv3x3_tail(const Tensor& x, const Tensor& w_packed, const optional<Tensor>& bias, int64_t cout, int64_t act, const optional<Tensor>& ln_weight,
                  const optional<Tensor>& ln_bias, const optional<Tensor>& res, const Tensor& p1, const Tensor& p2, int64_t prec, double ln_eps, Tensor out) {
  dev_f32(w_packed, "w_packed"); dev_f32(p1, "p1"); dev_f32(p2, "p2");
  TORCH_CHECK(p1.is_contiguous() && p2.is_contiguous() && p1.sizes() == p2.sizes() && p1.size(0) == x.size(0), "prv2::conv3x3_tail: p1 / p2 are dense [n, ph, pw] maps");
  prv2_conv_desc d = desc3x3(x, out, cout, act, prec, ln_eps);
  const float* pr = nullptr;
  if (res.has_value()) { d.ld_res = (int32_t)nhwc_ld(*res, "res"); pr = res->data_ptr<float>(); }
  TORCH_CHECK(prv2_conv2d_tail_supported(&d), "prv2::conv3x3_tail: layer not covered");
  Launch L(x);
  ok(prv2_conv2d_tail(&d, x.data_ptr<float>(), w_packed.data_ptr(), opt_ptr(bias, "bias", cout), opt_ptr(ln_weight, "ln_weight", cout), opt_ptr(ln_bias, "ln_bias", cout),
                      pr, p1.data_ptr<float>(), p2.data_ptr<float>(), (int)p1.size(-2), (int)p1.size(-1), out.data_ptr<float>(), L.stream), "conv3x3_tail");
}

// include/prv2.h::prv2_conv2d_pre: y = act([LN](conv3x3(x) + pre + bias)) (+ res) -- fusion_layers_1[l](cat([c, f])) over f with the coarse
// half of the conv (coarse_tap_gather) as the addend (bi_directional_fusion_model.py:424-426)
Tensor conv3x3_pre(const Tensor& x, const Tensor& w_packed, const optional<Tensor>& bias, const Tensor& pre, int64_t cout, int64_t act,
                   const optional<Tensor>& ln_weight, const optional<Tensor>& ln_bias, const optional<Tensor>& res, int64_t prec, double ln_eps,
                   const optional<Tensor>& out) {
  dev_f32(w_packed, "w_packed");
  Tensor y = out_or_alloc(out, x, x.size(0), x.size(1), x.size(2), cout, "conv3x3_pre");
  prv2_conv_desc d = desc3x3(x, y, cout, act, prec, ln_eps);
  TORCH_CHECK(pre.sizes() == y.sizes(), "prv2::conv3x3_pre: pre must have the output's shape");
  const float* pr = nullptr;
  if (res.has_value()) { d.ld_res = (int32_t)nhwc_ld(*res, "res"); pr = res->data_ptr<float>(); }
  Launch L(x);
  ok(prv2_conv2d_pre(&d, x.data_ptr<float>(), w_packed.data_ptr(), opt_ptr(bias, "bias", cout), pre.data_ptr<float>(), (int)nhwc_ld(pre, "pre"),
                     opt_ptr(ln_weight, "ln_weight", cout), opt_ptr(ln_bias, "ln_bias", cout), pr, y.data_ptr<float>(), L.stream), "conv3x3_pre");
  return y;
}

// include/prv2.h::prv2_pack_conv3x3_f6_weight / prv2_conv3x3_f6: the 256-column 3x3 conv of a GatedConvUnit in the fp16 + fp6 arithmetic
// (csrc/conv3x3_f6.hip; bi_directional_fusion_model.py:40-43, 58-64).  range: int32[1] (float bits of the largest |relu(x) x_scale| seen)
void pack_conv3x3_f6_weight(const Tensor& w, double w_scale, Tensor packed) {
  dev_f32(w, "weight"); dev_f32(packed, "packed");
  TORCH_CHECK(w.dim() == 4 && w.size(2) == 3 && w.size(3) == 3 && w.is_contiguous(), "prv2::pack_conv3x3_f6_weight: weight must be a contiguous [256, cin, 3, 3]");
  TORCH_CHECK(packed.numel() * 4 == prv2_conv3x3_f6_weight_bytes((int)w.size(0), (int)w.size(1)) && packed.numel() > 0, "prv2::pack_conv3x3_f6_weight: packed has the wrong size");
  Launch L(w);
  ok(prv2_pack_conv3x3_f6_weight(w.data_ptr<float>(), (float)w_scale, packed.data_ptr(), (int)w.size(0), (int)w.size(1), L.stream), "pack_conv3x3_f6_weight");
}

void conv3x3_f6(const Tensor& x, const Tensor& w_packed, const optional<Tensor>& bias, const optional<Tensor>& res, bool relu_in, double x_scale, double out_scale,
                const optional<Tensor>& range, Tensor out, int64_t fmt) {
  dev_f32(w_packed, "w_packed");
  const int64_t cout = out.size(3);
  TORCH_CHECK(w_packed.numel() * 4 == prv2_conv3x3_f6_weight_bytes((int)cout, (int)x.size(3)), "prv2::conv3x3_f6: w_packed does not match a 3x3 ", x.size(3), " -> ", cout, " conv");
  prv2_conv_desc d = desc3x3(x, out, cout, PRV2_ACT_NONE, PRV2_PREC_F16F6, 1e-6);
  d.relu_in = relu_in ? 1 : 0;
  d.fmt = (int32_t)fmt;
  TORCH_CHECK(out.size(0) == x.size(0) && out.size(1) == x.size(1) && out.size(2) == x.size(2), "prv2::conv3x3_f6: out must have x's batch and size");
  const float* pr = nullptr;
  if (res.has_value()) {
    TORCH_CHECK(res->sizes() == out.sizes(), "prv2::conv3x3_f6: res must have the output's shape");
    d.ld_res = (int32_t)nhwc_ld(*res, "res");
    pr = res->data_ptr<float>();
  }
  uint32_t* rw = nullptr;
  if (range.has_value()) {
    TORCH_CHECK(range->is_cuda() && range->scalar_type() == at::kInt && range->numel() == 1, "prv2::conv3x3_f6: range must be a GPU int32[1]");
    rw = reinterpret_cast<uint32_t*>(range->data_ptr<int32_t>());
  }
  TORCH_CHECK(prv2_conv3x3_f6_supported(&d), "prv2::conv3x3_f6: layer not covered (3x3 s1 p1, cout 256, cin % 64 == 0, width >= 16)");
  Launch L(x);
  ok(prv2_conv3x3_f6(&d, x.data_ptr<float>(), w_packed.data_ptr(), opt_ptr(bias, "bias", cout), pr, (float)x_scale, (float)out_scale, rw, out.data_ptr<float>(), L.stream),
     "conv3x3_f6");
}

// include/prv2.h::prv2_conv3x3_ln_gate_f6: the GatedConvUnit tail with its 3x3 conv in the fp16 + fp6 arithmetic; x / mul: the unit's pre-split ``out`` (raw buffers)
void conv3x3_ln_gate_f6(const Tensor& x, const Tensor& w_packed, const optional<Tensor>& bias, const optional<Tensor>& pre, const Tensor& ln_weight,
                        const Tensor& ln_bias, const Tensor& gate_w_packed, const optional<Tensor>& gate_bias, const optional<Tensor>& mul,
                        const optional<Tensor>& res, int64_t act, double ln_eps, double x_scale, double out_scale, const optional<Tensor>& range, Tensor out,
                        int64_t fmt) {
  dev_f32(w_packed, "w_packed");
  const int64_t cout = ln_weight.numel();
  TORCH_CHECK(w_packed.numel() * 4 == prv2_conv3x3_f6_weight_bytes((int)cout, (int)x.size(3)), "prv2::conv3x3_ln_gate_f6: w_packed does not match a 3x3 ", x.size(3), " -> ", cout, " conv");
  TORCH_CHECK(gate_w_packed.numel() * 4 == prv2_gate_weight_bytes((int)cout), "prv2::conv3x3_ln_gate_f6: gate_w_packed is not a pack_gate_weight image");
  prv2_conv_desc d = desc3x3(x, out, cout, act, PRV2_PREC_F16F6, ln_eps);
  d.fmt = (int32_t)fmt;
  const float *pm = nullptr, *pr = nullptr, *pp = nullptr;
  int32_t ld_pre = 0;
  if (mul.has_value()) { TORCH_CHECK(mul->sizes() == out.sizes(), "prv2::conv3x3_ln_gate_f6: mul must have the output's shape"); d.ld_mul = (int32_t)nhwc_ld(*mul, "mul"); pm = mul->data_ptr<float>(); }
  if (res.has_value()) { TORCH_CHECK(res->sizes() == out.sizes(), "prv2::conv3x3_ln_gate_f6: res must have the output's shape"); d.ld_res = (int32_t)nhwc_ld(*res, "res"); pr = res->data_ptr<float>(); }
  if (pre.has_value()) { TORCH_CHECK(pre->sizes() == out.sizes(), "prv2::conv3x3_ln_gate_f6: pre must have the output's shape"); ld_pre = (int32_t)nhwc_ld(*pre, "pre"); pp = pre->data_ptr<float>(); }
  uint32_t* rw = nullptr;
  if (range.has_value()) {
    TORCH_CHECK(range->is_cuda() && range->scalar_type() == at::kInt && range->numel() == 1, "prv2::conv3x3_ln_gate_f6: range must be a GPU int32[1]");
    rw = reinterpret_cast<uint32_t*>(range->data_ptr<int32_t>());
  }
  TORCH_CHECK(prv2_conv3x3_f6_supported(&d), "prv2::conv3x3_ln_gate_f6: layer not covered (3x3 s1 p1, cout 256, cin % 64 == 0, width >= 16)");
  Launch L(x);
  ok(prv2_conv3x3_ln_gate_f6(&d, x.data_ptr<float>(), w_packed.data_ptr(), opt_ptr(bias, "bias", cout), pp, ld_pre, opt_ptr(ln_weight, "ln_weight", cout),
                             opt_ptr(ln_bias, "ln_bias", cout), gate_w_packed.data_ptr(), opt_ptr(gate_bias, "gate_bias", cout), pm, pr, (float)x_scale,
                             (float)out_scale, rw, out.data_ptr<float>(), L.stream), "conv3x3_ln_gate_f6");
}

// include/prv2.h::prv2_upconv5x5*: output_conv2[0] o output_conv1 o interpolate as one 5x5 conv at u's resolution (csrc/upconv5.hip)
Tensor upconv5x5(const Tensor& u, const Tensor& w_packed, const Tensor& bias_map, int64_t cout, int64_t oh, int64_t ow, int64_t act, int64_t prec,
                 const optional<Tensor>& out) {
  const int64_t ldu = nhwc_ld(u, "u"), n = u.size(0), cu = u.size(3);
  TORCH_CHECK(w_packed.numel() * 4 == prv2_packed_weight_bytes((int)cout, (int)cu, 5, 5, 0, (int)prec), "prv2::upconv5x5: w_packed does not match (cout, u channels, 5x5, prec)");
  Tensor y = out_or_alloc(out, u, n, oh, ow, cout, "upconv5x5");
  prv2_ups_src us = {u.data_ptr<float>(), (int)u.size(1), (int)u.size(2), (int)ldu, (int)cu, 0};
  TORCH_CHECK(prv2_upconv5x5_supported(&us, (int)n, (int)oh, (int)ow, (int)cout, (int)prec), "prv2::upconv5x5: layer not covered (see include/prv2.h)");
  Launch L(u);
  ok(prv2_upconv5x5(&us, w_packed.data_ptr(), opt_ptr(bias_map, "bias_map", 25 * cout), (int)n, (int)oh, (int)ow, (int)cout, (int)act, (int)prec,
                    y.data_ptr<float>(), (int)nhwc_ld(y, "out"), 0, L.stream), "upconv5x5");
  return y;
}
Tensor upconv5x5_lines(const Tensor& u, int64_t oh, int64_t ow) {
  const int64_t ldu = nhwc_ld(u, "u"), n = u.size(0), cu = u.size(3);
  Tensor lines = at::empty({n, 1, 2 * u.size(2) + 2 * u.size(1), cu}, u.options());
  prv2_ups_src us = {u.data_ptr<float>(), (int)u.size(1), (int)u.size(2), (int)ldu, (int)cu, 0};
  Launch L(u);
  ok(prv2_upconv5x5_lines(&us, (int)n, (int)oh, (int)ow, lines.data_ptr<float>(), L.stream), "upconv5x5_lines");
  return lines;
}
void upconv5x5_ring_(Tensor y, const Tensor& g_edges, int64_t uh, int64_t uw, int64_t act) {
  const int64_t ldy = nhwc_ld(y, "y"), ldg = nhwc_ld(g_edges, "g_edges");
  TORCH_CHECK(g_edges.size(0) == y.size(0) && g_edges.size(1) == 1 && g_edges.size(2) == 2 * uw + 2 * uh && g_edges.size(3) == 28 * y.size(3),
              "prv2::upconv5x5_ring_: g_edges is [n, 1, 2 uw + 2 uh, 28 * cout]");
  Launch L(y);
  ok(prv2_upconv5x5_ring(y.data_ptr<float>(), (int)ldy, 0, (int)y.size(0), (int)y.size(1), (int)y.size(2), (int)y.size(3), g_edges.data_ptr<float>(), (int)ldg,
                         (int)uh, (int)uw, (int)act, L.stream), "upconv5x5_ring");
}

// include/prv2.h::prv2_chain32_*: the 32-channel full-resolution tail of BiDirectionalFusion, two 3x3 convs per kernel (csrc/chain32.hip)
Tensor pack_chain32_weight(const Tensor& w, int64_t kind) {
  dev_f32(w, "weight");
  TORCH_CHECK((w.dim() == 4 || w.dim() == 2) && w.size(0) == 32, "prv2::pack_chain32_weight: weight must be [32, cin, k, k] or [32, cin]");
  const int64_t taps = w.dim() == 4 ? w.size(2) * w.size(3) : 1;
  Tensor wc = w.contiguous();
  Tensor packed = at::empty({prv2_chain32_weight_bytes((int)kind, (int)taps) / 4}, w.options());
  Launch L(w);
  ok(prv2_pack_chain32_weight(wc.data_ptr<float>(), (int)w.size(1), (int)taps, (int)kind, packed.data_ptr(), L.stream), "pack_chain32_weight");
  return packed;
}
static prv2_chain32_desc chain32_desc(const Tensor& x, const Tensor& w1, const Tensor& w2, const Tensor& consts, const optional<Tensor>& pre, double ln_eps,
                                      const Tensor& y, const char* what) {
  prv2_chain32_desc d = {};
  d.ldx = (int)nhwc_ld(x, "x");
  d.ldy = (int)nhwc_ld(y, "y");
  TORCH_CHECK(x.size(3) == 32 && y.size(3) == 32 && y.size(0) == x.size(0) && y.size(1) == x.size(1) && y.size(2) == x.size(2), "prv2::", what,
              ": x and y are [n, h, w, 32]");
  TORCH_CHECK(w1.numel() * 4 == prv2_chain32_weight_bytes(0, 9) && w2.numel() * 4 == prv2_chain32_weight_bytes(1, 9), "prv2::", what,
              ": w1 / w2 are pack_chain32_weight images of 3x3 convs (kind 0 / 1)");
  d.x = x.data_ptr<float>(); d.y = y.data_ptr<float>(); d.w1 = w1.data_ptr(); d.w2 = w2.data_ptr();
  d.consts = opt_ptr(consts, "consts", 9 * 32);
  d.n = (int)x.size(0); d.h = (int)x.size(1); d.w = (int)x.size(2); d.ln_eps = (float)ln_eps;
  if (pre.has_value()) {
    d.ld_pre = (int)nhwc_ld(*pre, "pre");
    TORCH_CHECK(pre->size(0) == x.size(0) && pre->size(1) == x.size(1) && pre->size(2) == x.size(2) && pre->size(3) == 32, "prv2::", what, ": pre is [n, h, w, 32]");
    d.pre = pre->data_ptr<float>();
  }
  return d;
}
void chain32_c2f(const Tensor& x, const Tensor& w1, const Tensor& w2, const Tensor& wg, const Tensor& wo, const Tensor& consts, double b3,
                 const optional<Tensor>& pre, double ln_eps, Tensor y, optional<Tensor> depth) {
  prv2_chain32_desc d = chain32_desc(x, w1, w2, consts, pre, ln_eps, y, "chain32_c2f");
  TORCH_CHECK(wg.numel() * 4 == prv2_chain32_weight_bytes(1, 1) && wo.numel() * 4 == prv2_chain32_weight_bytes(1, 1),
              "prv2::chain32_c2f: wg / wo are pack_chain32_weight images of 1x1 convs (kind 1)");
  d.wg = wg.data_ptr(); d.wo = wo.data_ptr(); d.b3 = (float)b3;
  if (depth.has_value()) {
    dev_f32(*depth, "depth");
    TORCH_CHECK(depth->is_contiguous() && depth->numel() == x.size(0) * x.size(1) * x.size(2), "prv2::chain32_c2f: depth is dense [n, (1,) h, w]");
    d.depth = depth->data_ptr<float>();
  }
  Launch L(x);
  ok(prv2_chain32_c2f(&d, L.stream), "chain32_c2f");
}
void chain32_enc(const Tensor& x, const Tensor& w1, const Tensor& w2, const Tensor& wt, const Tensor& consts, const Tensor& pre, const Tensor& p1,
                 const Tensor& p2, double ln_eps, Tensor y) {
  prv2_chain32_desc d = chain32_desc(x, w1, w2, consts, pre, ln_eps, y, "chain32_enc");
  TORCH_CHECK(wt.numel() * 4 == prv2_chain32_weight_bytes(2, 9), "prv2::chain32_enc: wt is the kind-2 pack_chain32_weight image of the second conv");
  dev_f32(p1, "p1");
  dev_f32(p2, "p2");
  const int64_t px = x.size(0) * x.size(1) * x.size(2);
  TORCH_CHECK(p1.is_contiguous() && p2.is_contiguous() && p1.numel() == px && p2.numel() == px, "prv2::chain32_enc: p1 / p2 are dense [n, (1,) h, w] at x's size");
  d.wg = wt.data_ptr(); d.p1 = p1.data_ptr<float>(); d.p2 = p2.data_ptr<float>();
  Launch L(x);
  ok(prv2_chain32_enc(&d, L.stream), "chain32_enc");
}

// the per-frame coarse half of the cat([fine, coarse_roi]) convs (include/prv2.h::prv2_coarse_tap_knots / prv2_coarse_tap_gather)
Tensor coarse_tap_knots(const Tensor& g, int64_t cout, double knot_bh, double knot_bw) {
  const int64_t ldg = nhwc_ld(g, "g");
  TORCH_CHECK(g.size(0) == 1 && g.size(3) == 9 * cout, "prv2::coarse_tap_knots: g is [1, h, w, 9 * cout] (tap-major)");
  Tensor v = at::empty({1, 3 * g.size(1), 3 * g.size(2), cout}, g.options());
  Launch L(g);
  ok(prv2_coarse_tap_knots(g.data_ptr<float>(), (int)g.size(1), (int)g.size(2), (int)cout, (int)ldg, (float)knot_bh, (float)knot_bw, v.data_ptr<float>(), (int)cout,
                           L.stream), "coarse_tap_knots");
  return v;
}
Tensor coarse_tap_gather(const Tensor& v, const Tensor& g, double knot_bh, double knot_bw, const Tensor& boxes, double spatial_scale, int64_t oh, int64_t ow,
                         const optional<Tensor>& out) {
  const int64_t ldv = nhwc_ld(v, "v"), ldg = nhwc_ld(g, "g"), cout = v.size(3);
  dev_f32(boxes, "boxes");
  TORCH_CHECK(boxes.dim() == 2 && boxes.size(1) == 4 && boxes.is_contiguous() && g.size(3) == 9 * cout && v.size(1) == 3 * g.size(1) && v.size(2) == 3 * g.size(2),
              "prv2::coarse_tap_gather: v [1, 3h, 3w, cout], g [1, h, w, 9 cout], boxes [k, 4]");
  Tensor y = out_or_alloc(out, v, boxes.size(0), oh, ow, cout, "coarse_tap_gather");
  Launch L(v);
  ok(prv2_coarse_tap_gather(v.data_ptr<float>(), g.data_ptr<float>(), (int)g.size(1), (int)g.size(2), (int)cout, (int)ldv, (int)ldg, (float)knot_bh, (float)knot_bw,
                            boxes.data_ptr<float>(), (int)boxes.size(0), (float)spatial_scale, (int)oh, (int)ow, y.data_ptr<float>(), (int)nhwc_ld(y, "out"), L.stream),
     "coarse_tap_gather");
  return y;
}

// single-output-channel convs (final_conv 3x3 + clamp(update_base + offset, 0), output heads): include/prv2.h::prv2_conv2d_cout1
Tensor conv_cout1(const Tensor& x, const Tensor& weight, const optional<Tensor>& bias, int64_t k, int64_t act, double scale, const optional<Tensor>& res,
                  bool clamp0, const optional<Tensor>& out) {
  const int64_t ldx = nhwc_ld(x, "x");
  dev_f32(weight, "weight");
  Tensor y = out.has_value() ? *out : at::empty({x.size(0), 1, x.size(1), x.size(2)}, x.options());
  TORCH_CHECK(y.is_contiguous() && y.numel() == x.size(0) * x.size(1) * x.size(2), "prv2::conv_cout1: out must be a dense [n, 1, h, w] map");
  Launch L(x);
  ok(prv2_conv2d_cout1(x.data_ptr<float>(), (int)x.size(0), (int)x.size(1), (int)x.size(2), (int)x.size(3), (int)ldx, weight.data_ptr<float>(), (int)k,
                       opt_ptr(bias, "bias", 1), (int)act, (float)scale, res.has_value() ? res->data_ptr<float>() : nullptr, clamp0, y.data_ptr<float>(), L.stream),
     "conv_cout1");
  return y;
}

// depthwise k x k of the refiner encoders (include/prv2.h::prv2_dwconv2d_ex); weights tap-major [k*k, c]
Tensor dwconv2d(const Tensor& x, const Tensor& w_tapmajor, const optional<Tensor>& bias, int64_t k, int64_t stride, int64_t act, bool same_pad) {
  const int64_t ldx = nhwc_ld(x, "x"), c = x.size(3);
  dev_f32(w_tapmajor, "weight");
  const int64_t oh = same_pad ? (x.size(1) + stride - 1) / stride : (x.size(1) + 2 * (k / 2) - k) / stride + 1;
  const int64_t ow = same_pad ? (x.size(2) + stride - 1) / stride : (x.size(2) + 2 * (k / 2) - k) / stride + 1;
  Tensor y = alloc_nhwc(x, x.size(0), oh, ow, c);
  Launch L(x);
  ok(prv2_dwconv2d_ex(x.data_ptr<float>(), (int)x.size(0), (int)x.size(1), (int)x.size(2), (int)c, (int)ldx, w_tapmajor.data_ptr<float>(), opt_ptr(bias, "bias", c),
                      (int)k, (int)stride, (int)act, same_pad, y.data_ptr<float>(), (int)nhwc_ld(y, "out"), L.stream), "dwconv2d");
  return y;
}

// SqueezeExcite pieces (include/prv2.h::prv2_global_avgpool / prv2_se_gate / prv2_channel_scale)
Tensor global_avgpool(const Tensor& x) {
  const int64_t ldx = nhwc_ld(x, "x"), n = x.size(0), hw = x.size(1) * x.size(2), c = x.size(3);
  Tensor out = at::empty({n, c}, x.options());
  Tensor ws = at::empty({std::max<int64_t>(prv2_global_avgpool_workspace_floats((int)n, hw, (int)c), 1)}, x.options());
  Launch L(x);
  ok(prv2_global_avgpool(x.data_ptr<float>(), (int)n, hw, (int)c, (int)ldx, out.data_ptr<float>(), ws.data_ptr<float>(), L.stream), "global_avgpool");
  return out;
}
Tensor se_gate(const Tensor& mean, const Tensor& w1, const optional<Tensor>& b1, const Tensor& w2t, const optional<Tensor>& b2) {
  dev_f32(mean, "mean"); dev_f32(w1, "w1"); dev_f32(w2t, "w2t");
  const int64_t n = mean.size(0), c = mean.size(1), cse = w1.size(0);
  TORCH_CHECK(mean.is_contiguous() && w1.is_contiguous() && w2t.is_contiguous() && w1.size(1) == c && w2t.size(0) == cse && w2t.size(1) == c, "prv2::se_gate: shapes");
  Tensor g = at::empty({n, c}, mean.options()), ws = at::empty({n, cse}, mean.options());
  Launch L(mean);
  ok(prv2_se_gate(mean.data_ptr<float>(), (int)n, (int)c, w1.data_ptr<float>(), opt_ptr(b1, "b1", cse), (int)cse, w2t.data_ptr<float>(), opt_ptr(b2, "b2", c),
                  g.data_ptr<float>(), ws.data_ptr<float>(), L.stream), "se_gate");
  return g;
}
void channel_scale_(Tensor x, const Tensor& s) {
  const int64_t ldx = nhwc_ld(x, "x");
  dev_f32(s, "s");
  TORCH_CHECK(s.is_contiguous() && s.size(0) == x.size(0) && s.size(1) == x.size(3), "prv2::channel_scale_: s is [n, c]");
  Launch L(x);
  ok(prv2_channel_scale(x.data_ptr<float>(), (int)x.size(0), x.size(1) * x.size(2), (int)x.size(3), (int)ldx, s.data_ptr<float>(), L.stream), "channel_scale_");
}

// ViT token path (include/prv2.h: patchify, assemble_tokens, split-swizzled operands of the large linears)
Tensor patchify(const Tensor& img, int64_t p, int64_t ldo) {
  const int64_t ldi = nhwc_ld(img, "img"), gh = img.size(1) / p, gw = img.size(2) / p;
  Tensor rows = at::empty({img.size(0) * gh * gw, ldo}, img.options());
  Launch L(img);
  ok(prv2_patchify(img.data_ptr<float>(), (int)img.size(0), (int)gh, (int)gw, (int)p, (int)ldi, rows.data_ptr<float>(), (int)ldo, L.stream), "patchify");
  return rows;
}
Tensor assemble_tokens(const Tensor& emb, const Tensor& cls, const Tensor& pos, int64_t b, int64_t np_, int64_t dim) {
  dev_f32(emb, "emb"); dev_f32(cls, "cls"); dev_f32(pos, "pos");
  Tensor tok = at::empty({b, np_ + 1, dim}, emb.options());
  Launch L(emb);
  ok(prv2_assemble_tokens(emb.data_ptr<float>(), cls.data_ptr<float>(), pos.data_ptr<float>(), (int)b, (int)np_, (int)dim, tok.data_ptr<float>(), L.stream), "assemble_tokens");
  return tok;
}
Tensor split_ss(const Tensor& x) {
  dev_f32(x, "x");
  TORCH_CHECK(x.dim() == 2 && x.stride(1) == 1, "prv2::split_ss: x is [rows, c] with unit column stride");
  Tensor y = at::empty({x.size(0), x.size(1)}, x.options());
  Launch L(x);
  ok(prv2_split_ss(x.data_ptr<float>(), x.size(0), (int)x.size(1), (int)x.stride(0), y.data_ptr(), L.stream), "split_ss");
  return y;
}
void layernorm_ss(const Tensor& x, const Tensor& weight, const Tensor& bias, double eps, Tensor y_ss) {
  dev_f32(x, "x"); dev_f32(y_ss, "y_ss");
  TORCH_CHECK(x.dim() == 2 && x.stride(1) == 1 && y_ss.is_contiguous() && y_ss.numel() == x.numel(), "prv2::layernorm_ss: x [rows, c] (unit column stride), y_ss dense");
  const int64_t c = x.size(1);
  Launch L(x);
  ok(prv2_layernorm_ss(x.data_ptr<float>(), x.size(0), (int)c, (int)x.stride(0), opt_ptr(weight, "weight", c), opt_ptr(bias, "bias", c), (float)eps, y_ss.data_ptr(),
                       L.stream), "layernorm_ss");
}
Tensor gemm_ss(const Tensor& a_ss, const Tensor& w_packed, int64_t cout, const optional<Tensor>& bias, const optional<Tensor>& gamma, const optional<Tensor>& res,
               int64_t act, bool out_ss, const optional<Tensor>& out) {
  dev_f32(a_ss, "a_ss"); dev_f32(w_packed, "w_packed");
  TORCH_CHECK(a_ss.dim() == 2 && a_ss.is_contiguous(), "prv2::gemm_ss: a_ss is a dense [rows, k] container");
  const int64_t m = a_ss.size(0), k = a_ss.size(1);
  Tensor y = out.has_value() ? *out : at::empty({m, cout}, a_ss.options());
  TORCH_CHECK(y.dim() == 2 && y.size(0) == m && y.size(1) == cout && y.stride(1) == 1, "prv2::gemm_ss: out is [rows, cout]");
  const float* pr = nullptr;
  int ld_res = 0;
  if (res.has_value()) { dev_f32(*res, "res"); TORCH_CHECK(res->dim() == 2 && res->stride(1) == 1, "prv2::gemm_ss: res rows"); pr = res->data_ptr<float>(); ld_res = (int)res->stride(0); }
  Launch L(a_ss);
  ok(prv2_gemm_ss(a_ss.data_ptr(), m, (int)k, w_packed.data_ptr(), (int)cout, opt_ptr(bias, "bias", cout), opt_ptr(gamma, "gamma", cout), pr, ld_res, (int)act,
                  out_ss ? nullptr : y.data_ptr<float>(), (int)y.stride(0), out_ss ? y.data_ptr() : nullptr, L.stream), "gemm_ss");
  return y;
}
Tensor attention_ss(const Tensor& qkv, int64_t b, int64_t ntok, int64_t heads, const optional<Tensor>& bias, bool bias_image) {
  dev_f32(qkv, "qkv");
  TORCH_CHECK(qkv.is_contiguous() && qkv.numel() == b * ntok * 3 * heads * 64, "prv2::attention_ss: qkv must be contiguous [b * ntok, 3 * heads * 64]");
  if (bias.has_value()) {
    dev_f32(*bias, "bias");
    TORCH_CHECK(bias->is_contiguous() && (bias_image ? bias->numel() * 4 == prv2_attention_bias_image_bytes((int)heads, (int)ntok) : bias->dim() == 3),
                "prv2::attention_ss: bias is [heads, ntok, ld] or a pack_attention_bias image");
  }
  Tensor out = at::empty({b * ntok, heads * 64}, qkv.options());
  const int64_t wsb = prv2_attention_workspace_bytes((int)b, (int)ntok, (int)heads, PRV2_PREC_BF16X3);
  Tensor ws = at::empty({wsb > 0 ? wsb : 1}, qkv.options().dtype(at::kByte));
  Launch L(qkv);
  ok(prv2_attention_ss(qkv.data_ptr<float>(), (int)b, (int)ntok, (int)heads, 64, bias.has_value() ? bias->data_ptr<float>() : nullptr,
                       bias.has_value() ? (bias_image ? PRV2_ATTENTION_BIAS_IMAGE : (int)bias->size(2)) : 0, out.data_ptr(), wsb > 0 ? ws.data_ptr() : nullptr, wsb, L.stream), "attention_ss");
  return out;
}

Tensor gemm_ss_qkv(const Tensor& a_ss, const Tensor& w_packed, int64_t cout, const optional<Tensor>& bias, int64_t q_cols, double q_scale) {
  dev_f32(a_ss, "a_ss"); dev_f32(w_packed, "w_packed");
  TORCH_CHECK(a_ss.dim() == 2 && a_ss.is_contiguous(), "prv2::gemm_ss_qkv: a_ss is a dense [rows, k] container");
  Tensor y = at::empty({a_ss.size(0), cout}, a_ss.options());
  Launch L(a_ss);
  ok(prv2_gemm_ss_qkv(a_ss.data_ptr(), a_ss.size(0), (int)a_ss.size(1), w_packed.data_ptr(), (int)cout, opt_ptr(bias, "bias", cout), (int)q_cols, (float)q_scale, y.data_ptr(),
                      L.stream), "gemm_ss_qkv");
  return y;
}
Tensor attention_qkv_ss(const Tensor& qkv_ss, int64_t b, int64_t ntok, int64_t heads, const optional<Tensor>& bias, bool bias_image, bool out_ss) {
  dev_f32(qkv_ss, "qkv_ss");
  TORCH_CHECK(qkv_ss.is_contiguous() && qkv_ss.numel() == b * ntok * 3 * heads * 64, "prv2::attention_qkv_ss: qkv_ss must be a dense [b * ntok, 3 * heads * 64] container");
  if (bias.has_value()) {
    dev_f32(*bias, "bias");
    TORCH_CHECK(bias->is_contiguous() && (bias_image ? bias->numel() * 4 == prv2_attention_bias_image_bytes((int)heads, (int)ntok) : bias->dim() == 3),
                "prv2::attention_qkv_ss: bias is [heads, ntok, ld] or a pack_attention_bias image");
  }
  Tensor out = at::empty({b * ntok, heads * 64}, qkv_ss.options());
  Launch L(qkv_ss);
  ok(prv2_attention_qkv_ss(qkv_ss.data_ptr(), (int)b, (int)ntok, (int)heads, 64, bias.has_value() ? bias->data_ptr<float>() : nullptr,
                           bias.has_value() ? (bias_image ? PRV2_ATTENTION_BIAS_IMAGE : (int)bias->size(2)) : 0, out_ss ? nullptr : out.data_ptr<float>(),
                           out_ss ? out.data_ptr() : nullptr, L.stream), "attention_qkv_ss");
  return out;
}

// input stage and small placement kernels
Tensor bicubic_resize(const Tensor& img_hwc, int64_t oh, int64_t ow) {
  TORCH_CHECK(img_hwc.is_cuda() && (img_hwc.scalar_type() == at::kByte || img_hwc.scalar_type() == at::kFloat) && img_hwc.dim() == 3 && img_hwc.size(2) == 3 &&
                  img_hwc.is_contiguous(), "prv2::bicubic_resize: a contiguous [h, w, 3] uint8 or float32 GPU image");
  Tensor out = at::empty({3, oh, ow}, img_hwc.options().dtype(at::kFloat));
  Launch L(img_hwc);
  ok(prv2_bicubic_resize(img_hwc.data_ptr(), img_hwc.scalar_type() == at::kByte, (int)img_hwc.size(0), (int)img_hwc.size(1), out.data_ptr<float>(), (int)oh, (int)ow,
                         L.stream), "bicubic_resize");
  return out;
}
void depth_pair_fill(const Tensor& p1, const Tensor& p2, Tensor tail) {
  // tail: the [n, oh, ow, 4] channel slice (c0 .. c0 + 3) of the concat buffer that receives (p1, p2, 0, 0)
  dev_f32(p1, "p1"); dev_f32(p2, "p2");
  TORCH_CHECK(p1.is_contiguous() && p2.is_contiguous() && p1.sizes() == p2.sizes() && tail.size(3) == 4 && tail.size(0) == p1.size(0), "prv2::depth_pair_fill: shapes");
  Launch L(tail);
  ok(prv2_depth_pair_fill(p1.data_ptr<float>(), p2.data_ptr<float>(), (int)p1.size(0), (int)p1.size(-2), (int)p1.size(-1), (int)tail.size(1), (int)tail.size(2),
                          tail.data_ptr<float>(), (int)nhwc_ld(tail, "tail"), L.stream), "depth_pair_fill");
}
void conv_border_bias_(Tensor y, const Tensor& tap_bias) {
  dev_f32(tap_bias, "tap_bias");
  TORCH_CHECK(tap_bias.is_contiguous() && tap_bias.size(0) == 9 && tap_bias.size(1) == y.size(3), "prv2::conv_border_bias_: tap_bias is [9, c]");
  Launch L(y);
  ok(prv2_conv_border_bias(y.data_ptr<float>(), (int)y.size(0), (int)y.size(1), (int)y.size(2), (int)y.size(3), (int)nhwc_ld(y, "y"), tap_bias.data_ptr<float>(), L.stream),
     "conv_border_bias_");
}
Tensor add_nhwc(const Tensor& a, const Tensor& b, const optional<Tensor>& out) {
  TORCH_CHECK(a.sizes() == b.sizes(), "prv2::add_nhwc: shapes differ");
  Tensor y = out_or_alloc(out, a, a.size(0), a.size(1), a.size(2), a.size(3), "add_nhwc");
  Launch L(a);
  ok(prv2_add(a.data_ptr<float>(), (int)nhwc_ld(a, "a"), b.data_ptr<float>(), (int)nhwc_ld(b, "b"), a.size(0) * a.size(1) * a.size(2), (int)a.size(3), y.data_ptr<float>(),
              (int)nhwc_ld(y, "out"), L.stream), "add_nhwc");
  return y;
}
void zero_pad_channels_(Tensor buf, int64_t c) {
  dev_f32(buf, "buf");
  TORCH_CHECK(buf.dim() == 4 && buf.is_contiguous() && c <= buf.size(3), "prv2::zero_pad_channels_: buf is a dense NHWC buffer [n, h, w, ld], c <= ld");
  Launch L(buf);
  ok(prv2_zero_pad_channels(buf.data_ptr<float>(), buf.size(0) * buf.size(1) * buf.size(2), (int)c, (int)buf.size(3), L.stream), "zero_pad_channels_");
}

// nn.LayerNorm over the last dimension of [rows, c] (tokens) or of an NHWC map (the reference's channels-first LayerNorm)
Tensor layernorm(const Tensor& x, const Tensor& weight, const Tensor& bias, double eps, int64_t act, const optional<Tensor>& out) {
  dev_f32(x, "x");
  if (out.has_value()) {  // rows [M, c] with a row stride (in place: out may be x itself)
    TORCH_CHECK(x.dim() == 2 && out->dim() == 2 && out->sizes() == x.sizes() && x.stride(1) == 1 && out->stride(1) == 1,
                "prv2::layernorm: with out=, x and out are [rows, c] tensors with unit column stride");
    dev_f32(*out, "out");
    const int64_t rows = x.size(0), c = x.size(1);
    Launch L(x);
    ok(prv2_layernorm(x.data_ptr<float>(), rows, (int)c, (int)(rows == 1 ? c : x.stride(0)), opt_ptr(weight, "weight", c), opt_ptr(bias, "bias", c),
                      (float)eps, (int)act, out->data_ptr<float>(), (int)(rows == 1 ? c : out->stride(0)), L.stream), "layernorm");
    return *out;
  }
  TORCH_CHECK(x.dim() >= 2 && x.stride(-1) == 1, "prv2::layernorm: x must have unit stride along the normalised dimension");
  const int64_t c = x.size(-1);
  Tensor xr = x.dim() == 2 ? x : (x.is_contiguous() ? x.view({-1, c}) : x.contiguous().view({-1, c}));
  TORCH_CHECK(xr.size(0) == 1 || xr.stride(0) >= c, "prv2::layernorm: unsupported strides");
  Tensor y = at::empty({xr.size(0), c}, x.options());
  Launch L(x);
  ok(prv2_layernorm(xr.data_ptr<float>(), xr.size(0), (int)c, (int)(xr.size(0) == 1 ? c : xr.stride(0)), opt_ptr(weight, "weight", c),
                    opt_ptr(bias, "bias", c), (float)eps, (int)act, y.data_ptr<float>(), (int)c, L.stream), "layernorm");
  return y.view(x.sizes());
}

// softmax((q * scale) k^T) v, head_dim 64; qkv rows [3][heads][64] as nn.Linear(dim, 3 * dim) leaves them (attention.py:49-62)
// the score bias of a (model, resolution) re-ordered once for the bf16x3 kernel (include/prv2.h::prv2_pack_attention_bias)
Tensor pack_attention_bias(const Tensor& bias, int64_t ntok) {
  dev_f32(bias, "bias");
  TORCH_CHECK(bias.is_contiguous() && bias.dim() == 3 && bias.size(1) == ntok, "prv2::pack_attention_bias: bias is contiguous [heads, ntok, ld]");
  Tensor img = at::empty({prv2_attention_bias_image_bytes((int)bias.size(0), (int)ntok) / 4}, bias.options());
  Launch L(bias);
  ok(prv2_pack_attention_bias(bias.data_ptr<float>(), (int)bias.size(0), (int)ntok, (int)bias.size(2), img.data_ptr<float>(), L.stream), "pack_attention_bias");
  return img;
}

Tensor attention_fwd(const Tensor& qkv, int64_t b, int64_t ntok, int64_t heads, int64_t prec, const optional<Tensor>& bias, bool bias_image) {
  dev_f32(qkv, "qkv");
  if (bias.has_value() && bias_image) {
    dev_f32(*bias, "bias");
    TORCH_CHECK(bias->numel() * 4 == prv2_attention_bias_image_bytes((int)heads, (int)ntok), "prv2::attention_fwd: bias is not a pack_attention_bias image of (heads, ntok)");
  } else if (bias.has_value()) {  // additive score bias [heads, ntok, ld >= roundup(ntok, 64)] shared by the batch (BEiT relative position bias)
    dev_f32(*bias, "bias");
    TORCH_CHECK(bias->is_contiguous() && bias->dim() == 3 && bias->size(0) == heads && bias->size(1) == ntok && bias->size(2) >= (ntok + 63) / 64 * 64,
                "prv2::attention_fwd: bias must be contiguous [heads, ntok, ld] with ld >= ntok rounded up to 64");
  }
  TORCH_CHECK(qkv.is_contiguous() && qkv.numel() == b * ntok * 3 * heads * 64, "prv2::attention_fwd: qkv must be contiguous [b * ntok, 3 * heads * 64]");
  Tensor out = at::empty({b * ntok, heads * 64}, qkv.options());
  const int64_t wsb = prv2_attention_workspace_bytes((int)b, (int)ntok, (int)heads, (int)prec);
  Tensor ws = at::empty({wsb > 0 ? wsb : 1}, qkv.options().dtype(at::kByte));
  Launch L(qkv);
  ok(prv2_attention_bias(qkv.data_ptr<float>(), (int)b, (int)ntok, (int)heads, 64, bias.has_value() ? bias->data_ptr<float>() : nullptr,
                         bias.has_value() ? (bias_image ? PRV2_ATTENTION_BIAS_IMAGE : (int)bias->size(2)) : 0, out.data_ptr<float>(), (int)prec,
                         wsb > 0 ? ws.data_ptr() : nullptr, wsb, L.stream),
     "attention_fwd");
  return out;
}

// crop + bilinear(align_corners=True) resize of k tiles of a CHW image -> NHWC [k, oh, ow, 3] (+ (v - mean) / std)
Tensor crop_resize_bilinear(const Tensor& img_chw, const Tensor& tiles, int64_t ch, int64_t cw, int64_t oh, int64_t ow,
                            optional<at::ArrayRef<double>> mean, optional<at::ArrayRef<double>> std_, const optional<Tensor>& out) {
  dev_f32(img_chw, "img_chw");
  TORCH_CHECK(img_chw.dim() == 3 && img_chw.size(0) == 3 && img_chw.is_contiguous(), "prv2::crop_resize_bilinear: img must be contiguous [3, H, W]");
  TORCH_CHECK(tiles.is_cuda() && tiles.scalar_type() == at::kInt && tiles.dim() == 2 && tiles.size(1) == 2 && tiles.is_contiguous(),
              "prv2::crop_resize_bilinear: tiles must be an int32 GPU tensor [k, 2] = (h_start, w_start)");
  TORCH_CHECK(mean.has_value() == std_.has_value() && (!mean.has_value() || (mean->size() == 3 && std_->size() == 3)), "prv2::crop_resize_bilinear: mean / std are 3 + 3 floats");
  const int64_t k = tiles.size(0);
  Tensor y = out_or_alloc(out, img_chw, k, oh, ow, 3, "crop_resize_bilinear");
  float m[3], s[3];
  if (mean.has_value()) for (int i = 0; i < 3; ++i) { m[i] = (float)(*mean)[i]; s[i] = (float)(*std_)[i]; }
  Launch L(img_chw);
  ok(prv2_crop_resize(img_chw.data_ptr<float>(), (int)img_chw.size(1), (int)img_chw.size(2), tiles.data_ptr<int32_t>(), (int)k, (int)ch, (int)cw, (int)oh,
                      (int)ow, mean.has_value() ? m : nullptr, mean.has_value() ? s : nullptr, y.data_ptr<float>(), (int)nhwc_ld(y, "out"), L.stream),
     "crop_resize_bilinear");
  return y;
}

// torchvision.ops.roi_align(feat.repeat(K), boxes, feat.shape[-2:], h / ph, aligned=True) for every level of the coarse pyramid
// (patchrefinerplus.py:263-283) without the K copies.  feats: NHWC [1, h, w, c]; boxes [k, 4] = (x1, y1, x2, y2) in lr pixels.
std::vector<Tensor> roi_gather_pyramid(at::TensorList feats, const Tensor& boxes, int64_t ph) {
  dev_f32(boxes, "boxes");
  TORCH_CHECK(boxes.dim() == 2 && boxes.size(1) == 4 && boxes.is_contiguous(), "prv2::roi_gather_pyramid: boxes must be contiguous [k, 4]");
  std::vector<Tensor> outs;
  for (const Tensor& f : feats) {
    const int64_t ld = nhwc_ld(f, "feat");
    TORCH_CHECK(f.size(0) == 1, "prv2::roi_gather_pyramid: one coarse pyramid per frame (batch 1, patchrefinerplus.py:477)");
    Tensor y = alloc_nhwc(f, boxes.size(0), f.size(1), f.size(2), f.size(3));
    Launch L(f);
    ok(prv2_roi_align(f.data_ptr<float>(), (int)f.size(1), (int)f.size(2), (int)f.size(3), (int)ld, boxes.data_ptr<float>(), (int)boxes.size(0),
                      (float)((double)f.size(1) / (double)ph), (int)f.size(1), (int)f.size(2), y.data_ptr<float>(), (int)nhwc_ld(y, "out"), L.stream),
       "roi_gather_pyramid");
    outs.push_back(y);
  }
  return outs;
}

// one level of the above with explicit scale / output size: torchvision.ops.roi_align(feat.repeat(K), boxes, (oh, ow), scale, aligned=True)
Tensor roi_align(const Tensor& feat, const Tensor& boxes, double spatial_scale, int64_t oh, int64_t ow, const optional<Tensor>& out, bool x2) {
  dev_f32(boxes, "boxes");
  TORCH_CHECK(boxes.dim() == 2 && boxes.size(1) == 4 && boxes.is_contiguous(), "prv2::roi_align: boxes must be contiguous [k, 4]");
  const int64_t ld = nhwc_ld(feat, "feat");
  TORCH_CHECK(feat.size(0) == 1, "prv2::roi_align: one feature map (batch 1)");
  Tensor y = out_or_alloc(out, feat, boxes.size(0), oh, ow, feat.size(3), "roi_align");
  Launch L(feat);
  ok((x2 ? prv2_roi_align_x2 : prv2_roi_align)(feat.data_ptr<float>(), (int)feat.size(1), (int)feat.size(2), (int)feat.size(3), (int)ld, boxes.data_ptr<float>(),
                                                (int)boxes.size(0), (float)spatial_scale, (int)oh, (int)ow, y.data_ptr<float>(), (int)nhwc_ld(y, "out"), L.stream),
     "roi_align");
  return y;
}

// F.interpolate(mode='bilinear', align_corners=True) on NHWC
Tensor upsample_bilinear_ac(const Tensor& x, int64_t oh, int64_t ow, const optional<Tensor>& out) {
  const int64_t ld = nhwc_ld(x, "x");
  Tensor y = out_or_alloc(out, x, x.size(0), oh, ow, x.size(3), "upsample_bilinear_ac");
  Launch L(x);
  ok(prv2_upsample_bilinear(x.data_ptr<float>(), (int)x.size(0), (int)x.size(1), (int)x.size(2), (int)x.size(3), (int)ld, (int)oh, (int)ow, y.data_ptr<float>(),
                            (int)nhwc_ld(y, "out"), L.stream), "upsample_bilinear_ac");
  return y;
}

// RunningAverageMap on the device (estimator/models/utils.py:22-49): init = paste of the first pass, update = weighted
// running mean in tile order, resize = avg nearest / count bilinear(align_corners)
void blend_args(const Tensor& avg, const Tensor& cnt, const Tensor& pred, const Tensor& mask, const Tensor& tiles, int64_t th, int64_t tw) {
  dev_f32(avg, "avg"); dev_f32(cnt, "cnt"); dev_f32(pred, "pred"); dev_f32(mask, "mask");
  TORCH_CHECK(avg.dim() == 2 && avg.is_contiguous() && cnt.sizes() == avg.sizes() && cnt.is_contiguous(), "prv2::blend: avg / cnt must be contiguous [H, W] maps");
  TORCH_CHECK(pred.is_contiguous() && pred.dim() >= 3, "prv2::blend: pred must be contiguous [k, (1,) ph, pw]");
  TORCH_CHECK(tiles.is_cuda() && tiles.scalar_type() == at::kInt && tiles.is_contiguous() && tiles.dim() == 2 && tiles.size(1) == 2 && tiles.size(0) == pred.size(0),
              "prv2::blend: tiles must be an int32 GPU tensor [k, 2], one row per prediction");
  TORCH_CHECK(mask.is_contiguous() && mask.dim() == 2 && mask.size(0) == th && mask.size(1) == tw, "prv2::blend: mask must be [th, tw]");
}
void blend_init(Tensor avg, Tensor cnt, const Tensor& pred, const Tensor& mask, const Tensor& tiles, int64_t th, int64_t tw) {
  blend_args(avg, cnt, pred, mask, tiles, th, tw);
  Launch L(avg);
  ok(prv2_blend_paste(avg.data_ptr<float>(), cnt.data_ptr<float>(), (int)avg.size(0), (int)avg.size(1), pred.data_ptr<float>(), (int)pred.size(-2), (int)pred.size(-1),
                      mask.data_ptr<float>(), tiles.data_ptr<int32_t>(), (int)tiles.size(0), (int)th, (int)tw, L.stream), "blend_init");
}
void blend_update(Tensor avg, Tensor cnt, const Tensor& pred, const Tensor& mask, const Tensor& tiles, int64_t th, int64_t tw) {
  blend_args(avg, cnt, pred, mask, tiles, th, tw);
  Launch L(avg);
  ok(prv2_blend_update(avg.data_ptr<float>(), cnt.data_ptr<float>(), (int)avg.size(0), (int)avg.size(1), pred.data_ptr<float>(), (int)pred.size(-2), (int)pred.size(-1),
                       mask.data_ptr<float>(), tiles.data_ptr<int32_t>(), (int)tiles.size(0), (int)th, (int)tw, L.stream), "blend_update");
}
std::tuple<Tensor, Tensor> blend_resize(const Tensor& avg, const Tensor& cnt, int64_t oh, int64_t ow) {
  dev_f32(avg, "avg"); dev_f32(cnt, "cnt");
  TORCH_CHECK(avg.dim() == 2 && avg.is_contiguous() && cnt.sizes() == avg.sizes() && cnt.is_contiguous(), "prv2::blend_resize: avg / cnt must be contiguous [H, W] maps");
  Tensor a = at::empty({oh, ow}, avg.options()), c = at::empty({oh, ow}, avg.options());
  Launch L(avg);
  ok(prv2_blend_resize(avg.data_ptr<float>(), cnt.data_ptr<float>(), (int)avg.size(0), (int)avg.size(1), a.data_ptr<float>(), c.data_ptr<float>(), (int)oh, (int)ow, L.stream),
     "blend_resize");
  return {a, c};
}

// ZoeDepth metric-bins head, elementwise parts (attractor.py:45-57,186-206; dist_layers.py:29-69,100-116; zoedepth_v1.py:219)
Tensor zoe_attractor(const Tensor& attr, const Tensor& bins, double alpha) {
  const int64_t lda = nhwc_ld(attr, "attr"), ldb = nhwc_ld(bins, "bins");
  TORCH_CHECK(attr.size(0) == bins.size(0) && attr.size(1) == bins.size(1) && attr.size(2) == bins.size(2), "prv2::zoe_attractor: attr / bins maps differ in size");
  Tensor y = alloc_nhwc(bins, bins.size(0), bins.size(1), bins.size(2), bins.size(3));
  Launch L(bins);
  ok(prv2_zoe_attractor(attr.data_ptr<float>(), (int)lda, (int)attr.size(3), bins.data_ptr<float>(), (int)ldb, (int)bins.size(3), (float)alpha,
                        bins.size(0) * bins.size(1) * bins.size(2), y.data_ptr<float>(), (int)nhwc_ld(y, "out"), L.stream), "zoe_attractor");
  return y;
}
Tensor zoe_bins_head(const Tensor& pt, const Tensor& centers, double min_temp, double max_temp) {
  const int64_t ldp = nhwc_ld(pt, "pt"), ldc = nhwc_ld(centers, "centers");
  TORCH_CHECK(pt.size(3) == 4 && pt.size(0) == centers.size(0) && pt.size(1) == centers.size(1) && pt.size(2) == centers.size(2),
              "prv2::zoe_bins_head: pt must be [n, h, w, 4] over the same pixels as centers");
  Tensor depth = at::empty({pt.size(0), 1, pt.size(1), pt.size(2)}, pt.options());
  Launch L(pt);
  ok(prv2_zoe_logbinom_depth(pt.data_ptr<float>(), (int)ldp, centers.data_ptr<float>(), (int)ldc, (int)centers.size(3), (float)min_temp, (float)max_temp,
                             pt.size(0) * pt.size(1) * pt.size(2), depth.data_ptr<float>(), L.stream), "zoe_bins_head");
  return depth;
}

// layout changes at the boundary (image_lr in, features out)
Tensor nchw_to_nhwc(const Tensor& x) {
  dev_f32(x, "x");
  TORCH_CHECK(x.dim() == 4, "prv2::nchw_to_nhwc: x must be [n, c, h, w]");
  Tensor xc = x.contiguous();
  Tensor y = alloc_nhwc(x, x.size(0), x.size(2), x.size(3), x.size(1));
  Launch L(x);
  ok(prv2_nchw_to_nhwc(xc.data_ptr<float>(), (int)x.size(0), (int)x.size(1), (int)x.size(2), (int)x.size(3), y.data_ptr<float>(), (int)nhwc_ld(y, "out"), L.stream), "nchw_to_nhwc");
  return y;
}
Tensor nhwc_to_nchw(const Tensor& x) {
  const int64_t ld = nhwc_ld(x, "x");
  Tensor y = at::empty({x.size(0), x.size(3), x.size(1), x.size(2)}, x.options());
  Launch L(x);
  ok(prv2_nhwc_to_nchw(x.data_ptr<float>(), (int)x.size(0), (int)x.size(3), (int)x.size(1), (int)x.size(2), (int)ld, y.data_ptr<float>(), L.stream), "nhwc_to_nchw");
  return y;
}

int64_t abi_version() { return prv2_abi_version(); }

}  // namespace

TORCH_LIBRARY(prv2, m) {
  m.def("abi_version() -> int", &abi_version);
  m.def("pack_conv_weight(Tensor weight, Tensor? bn_scale=None, int convt_k=0, int prec=0) -> Tensor");
  m.def("conv2d(Tensor x, Tensor w_packed, Tensor? bias, int cout, int kh, int kw, int stride=1, int pad=0, int act=0, bool relu_in=False, "
        "Tensor? ln_weight=None, Tensor? ln_bias=None, Tensor? gamma=None, Tensor? mul=None, Tensor? res=None, Tensor? res2=None, int convt_k=0, "
        "int prec=0, float ln_eps=1e-06, bool same_pad=False, Tensor(a!)? out=None, int fmt=0, bool force_generic=False, int part=0) -> Tensor");
  m.def("conv3x3_ups(Tensor? x, Tensor u, Tensor w_packed, Tensor? bias, int cout, int oh, int ow, int act=0, Tensor? ln_weight=None, "
        "Tensor? ln_bias=None, Tensor? res=None, int prec=1, float ln_eps=1e-06, Tensor(a!)? out=None) -> Tensor");
  // (``add`` may BE ``out`` -- every output element is read by the thread that writes it --: declared as a member of the same alias set)
  m.def("upconv3x3(Tensor u, Tensor w_packed, Tensor? bias, int cout, int oh, int ow, int act=0, int prec=1, Tensor(a!)? out=None, Tensor(a)? add=None) -> Tensor");
  m.def("pack_gate_weight(Tensor weight) -> Tensor");
  m.def("conv3x3_ln_gate(Tensor x, Tensor w_packed, Tensor? bias, Tensor ln_weight, Tensor ln_bias, Tensor? gate_w_packed=None, Tensor? gate_bias=None, "
        "Tensor? mul=None, Tensor? res=None, int act=1, bool relu_in=False, int prec=1, float ln_eps=1e-06, Tensor(a!)? out=None, Tensor? pre=None, "
        "int fmt=0) -> Tensor");
  m.def("conv3x3_tail(Tensor x, Tensor w_packed, Tensor? bias, int cout, int act, Tensor? ln_weight, Tensor? ln_bias, Tensor? res, Tensor p1, Tensor p2, "
        "int prec, float ln_eps, Tensor(a!) out) -> ()");
  m.def("conv3x3_pre(Tensor x, Tensor w_packed, Tensor? bias, Tensor pre, int cout, int act=0, Tensor? ln_weight=None, Tensor? ln_bias=None, Tensor? res=None, "
        "int prec=1, float ln_eps=1e-06, Tensor(a!)? out=None) -> Tensor");
  m.def("pack_conv3x3_f6_weight(Tensor weight, float w_scale, Tensor(a!) packed) -> ()");
  m.def("conv3x3_f6(Tensor x, Tensor w_packed, Tensor? bias, Tensor? res, bool relu_in, float x_scale, float out_scale, Tensor(a!)? range, Tensor(b!) out, "
        "int fmt=0) -> ()");
  m.def("conv3x3_ln_gate_f6(Tensor x, Tensor w_packed, Tensor? bias, Tensor? pre, Tensor ln_weight, Tensor ln_bias, Tensor gate_w_packed, Tensor? gate_bias, "
        "Tensor? mul, Tensor? res, int act, float ln_eps, float x_scale, float out_scale, Tensor(a!)? range, Tensor(b!) out, int fmt=3) -> ()");
  m.def("upconv5x5(Tensor u, Tensor w_packed, Tensor bias_map, int cout, int oh, int ow, int act=0, int prec=1, Tensor(a!)? out=None) -> Tensor");
  m.def("upconv5x5_lines(Tensor u, int oh, int ow) -> Tensor");
  m.def("upconv5x5_ring_(Tensor(a!) y, Tensor g_edges, int uh, int uw, int act=0) -> ()");
  m.def("pack_chain32_weight(Tensor weight, int kind) -> Tensor");
  m.def("chain32_c2f(Tensor x, Tensor w1, Tensor w2, Tensor wg, Tensor wo, Tensor consts, float b3, Tensor? pre, float ln_eps, Tensor(a!) y, "
        "Tensor(b!)? depth=None) -> ()");
  m.def("chain32_enc(Tensor x, Tensor w1, Tensor w2, Tensor wt, Tensor consts, Tensor pre, Tensor p1, Tensor p2, float ln_eps, Tensor(a!) y) -> ()");
  m.def("coarse_tap_knots(Tensor g, int cout, float knot_bh, float knot_bw) -> Tensor");
  m.def("coarse_tap_gather(Tensor v, Tensor g, float knot_bh, float knot_bw, Tensor boxes, float spatial_scale, int oh, int ow, Tensor(a!)? out=None) -> Tensor");
  m.def("conv_cout1(Tensor x, Tensor weight, Tensor? bias, int k, int act=0, float scale=1.0, Tensor? res=None, bool clamp0=False, Tensor(a!)? out=None) -> Tensor");
  m.def("dwconv2d(Tensor x, Tensor w_tapmajor, Tensor? bias, int k, int stride, int act=0, bool same_pad=False) -> Tensor");
  m.def("global_avgpool(Tensor x) -> Tensor");
  m.def("se_gate(Tensor mean, Tensor w1, Tensor? b1, Tensor w2t, Tensor? b2) -> Tensor");
  m.def("channel_scale_(Tensor(a!) x, Tensor s) -> ()");
  m.def("patchify(Tensor img, int p, int ldo) -> Tensor");
  m.def("assemble_tokens(Tensor emb, Tensor cls, Tensor pos, int b, int np, int dim) -> Tensor");
  m.def("split_ss(Tensor x) -> Tensor");
  m.def("layernorm_ss(Tensor x, Tensor weight, Tensor bias, float eps, Tensor(a!) y_ss) -> ()");
  m.def("gemm_ss(Tensor a_ss, Tensor w_packed, int cout, Tensor? bias=None, Tensor? gamma=None, Tensor? res=None, int act=0, bool out_ss=False, "
        "Tensor(a!)? out=None) -> Tensor");
  m.def("attention_ss(Tensor qkv, int b, int ntok, int heads, Tensor? bias=None, bool bias_image=False) -> Tensor");
  m.def("gemm_ss_qkv(Tensor a_ss, Tensor w_packed, int cout, Tensor? bias, int q_cols, float q_scale) -> Tensor");
  m.def("attention_qkv_ss(Tensor qkv_ss, int b, int ntok, int heads, Tensor? bias=None, bool bias_image=False, bool out_ss=True) -> Tensor");
  m.def("bicubic_resize(Tensor img_hwc, int oh, int ow) -> Tensor");
  m.def("depth_pair_fill(Tensor p1, Tensor p2, Tensor(a!) tail) -> ()");
  m.def("conv_border_bias_(Tensor(a!) y, Tensor tap_bias) -> ()");
  m.def("add_nhwc(Tensor a, Tensor b, Tensor(a!)? out=None) -> Tensor");
  m.def("zero_pad_channels_(Tensor(a!) buf, int c) -> ()");
  m.def("layernorm(Tensor x, Tensor weight, Tensor bias, float eps=1e-06, int act=0, Tensor(a!)? out=None) -> Tensor");
  m.def("roi_align(Tensor feat, Tensor boxes, float spatial_scale, int oh, int ow, Tensor(a!)? out=None, bool x2=False) -> Tensor");
  m.def("attention_fwd(Tensor qkv, int b, int ntok, int heads, int prec=0, Tensor? bias=None, bool bias_image=False) -> Tensor");
  m.def("pack_attention_bias(Tensor bias, int ntok) -> Tensor");
  m.def("crop_resize_bilinear(Tensor img_chw, Tensor tiles, int ch, int cw, int oh, int ow, float[]? mean=None, float[]? std=None, "
        "Tensor(a!)? out=None) -> Tensor");
  m.def("roi_gather_pyramid(Tensor[] feats, Tensor boxes, int ph) -> Tensor[]");
  m.def("upsample_bilinear_ac(Tensor x, int oh, int ow, Tensor(a!)? out=None) -> Tensor");
  m.def("blend_init(Tensor(a!) avg, Tensor(b!) cnt, Tensor pred, Tensor mask, Tensor tiles, int th, int tw) -> ()");
  m.def("blend_update(Tensor(a!) avg, Tensor(b!) cnt, Tensor pred, Tensor mask, Tensor tiles, int th, int tw) -> ()");
  m.def("blend_resize(Tensor avg, Tensor cnt, int oh, int ow) -> (Tensor, Tensor)");
  m.def("zoe_attractor(Tensor attr, Tensor bins, float alpha=300.0) -> Tensor");
  m.def("zoe_bins_head(Tensor pt, Tensor centers, float min_temp, float max_temp) -> Tensor");
  m.def("nchw_to_nhwc(Tensor x) -> Tensor");
  m.def("nhwc_to_nchw(Tensor x) -> Tensor");
}

// every op takes GPU tensors: registered for the CUDA dispatch key (= HIP on PyTorch-ROCm).  Calling one with CPU tensors
// fails in the dispatcher ("no kernel for CPU backend"): there is no CPU fallback to fall into.
TORCH_LIBRARY_IMPL(prv2, CUDA, m) {
  m.impl("pack_conv_weight", &pack_conv_weight);
  m.impl("conv2d", &conv2d);
  m.impl("conv3x3_ups", &conv3x3_ups);
  m.impl("upconv3x3", &upconv3x3);
  m.impl("pack_gate_weight", &pack_gate_weight);
  m.impl("conv3x3_ln_gate", &conv3x3_ln_gate);
  m.impl("conv3x3_tail", &conv3x3_tail);
  m.impl("conv3x3_pre", &conv3x3_pre);
  m.impl("pack_conv3x3_f6_weight", &pack_conv3x3_f6_weight);
  m.impl("conv3x3_f6", &conv3x3_f6);
  m.impl("conv3x3_ln_gate_f6", &conv3x3_ln_gate_f6);
  m.impl("upconv5x5", &upconv5x5);
  m.impl("upconv5x5_lines", &upconv5x5_lines);
  m.impl("upconv5x5_ring_", &upconv5x5_ring_);
  m.impl("pack_chain32_weight", &pack_chain32_weight);
  m.impl("chain32_c2f", &chain32_c2f);
  m.impl("chain32_enc", &chain32_enc);
  m.impl("coarse_tap_knots", &coarse_tap_knots);
  m.impl("coarse_tap_gather", &coarse_tap_gather);
  m.impl("conv_cout1", &conv_cout1);
  m.impl("dwconv2d", &dwconv2d);
  m.impl("global_avgpool", &global_avgpool);
  m.impl("se_gate", &se_gate);
  m.impl("channel_scale_", &channel_scale_);
  m.impl("patchify", &patchify);
  m.impl("assemble_tokens", &assemble_tokens);
  m.impl("split_ss", &split_ss);
  m.impl("layernorm_ss", &layernorm_ss);
  m.impl("gemm_ss", &gemm_ss);
  m.impl("attention_ss", &attention_ss);
  m.impl("gemm_ss_qkv", &gemm_ss_qkv);
  m.impl("attention_qkv_ss", &attention_qkv_ss);
  m.impl("bicubic_resize", &bicubic_resize);
  m.impl("depth_pair_fill", &depth_pair_fill);
  m.impl("conv_border_bias_", &conv_border_bias_);
  m.impl("add_nhwc", &add_nhwc);
  m.impl("zero_pad_channels_", &zero_pad_channels_);
  m.impl("layernorm", &layernorm);
  m.impl("attention_fwd", &attention_fwd);
  m.impl("pack_attention_bias", &pack_attention_bias);
  m.impl("crop_resize_bilinear", &crop_resize_bilinear);
  m.impl("roi_gather_pyramid", &roi_gather_pyramid);
  m.impl("roi_align", &roi_align);
  m.impl("upsample_bilinear_ac", &upsample_bilinear_ac);
  m.impl("blend_init", &blend_init);
  m.impl("blend_update", &blend_update);
  m.impl("blend_resize", &blend_resize);
  m.impl("zoe_attractor", &zoe_attractor);
  m.impl("zoe_bins_head", &zoe_bins_head);
  m.impl("nchw_to_nhwc", &nchw_to_nhwc);
  m.impl("nhwc_to_nchw", &nhwc_to_nchw);
}
