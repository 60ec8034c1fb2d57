/*
 * prv2.h -- C ABI of the MI355X (gfx950) hot path of patch-refined depth inference.
 *
 * The reference (zhyever/PatchRefinerV2) is pure Python/PyTorch: it has no FFI.  The seam this
 * library sits behind is therefore the set of torch ops the reference's hot path dispatches
 * (SURVEY.md 2.1 / 8b).  Each entry point below names the reference call site(s) it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless the name ends in _host or the type says otherwise;
 *   - activations are NHWC: element (n,y,x,c) lives at base[n*bstride + (y*W + x)*ld + c];
 *     ``ld`` (pixel stride, in floats) lets a producer write straight into a channel slice of
 *     a wider concatenation buffer, which is how every torch.cat on the path is made free;
 *   - ``stream`` is a hipStream_t passed as void*; all work is enqueued, nothing synchronises;
 *   - return value: 0 on success, non-zero on a rejected argument or a HIP error;
 *     prv2_last_error() returns the message of the calling thread's last failure;
 *   - no allocation, no hidden global state, graph-capture safe.
 */
#ifndef PRV2_H
#define PRV2_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRV2_ABI_VERSION 19

enum prv2_act { PRV2_ACT_NONE = 0, PRV2_ACT_RELU = 1, PRV2_ACT_GELU = 2, PRV2_ACT_SIGMOID = 3, PRV2_ACT_SOFTPLUS = 4,
                PRV2_ACT_SILU = 5 /* x * sigmoid(x): EfficientNet refiner encoder (timm 'swish') */ };

/* arithmetic of the matrix kernels */
enum prv2_prec {
  PRV2_PREC_F32 = 0,    /* v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate            */
  PRV2_PREC_BF16X3 = 1, /* operands split hi+lo bf16; hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 */
  PRV2_PREC_BF16 = 2,   /* plain bf16 operands (fast, ~1e-3 relative; not the parity path)          */
  PRV2_PREC_F16F6 = 3   /* fp16 product + two block-scaled fp6 (e2m3) corrections: what prv2_conv3x3_f6 computes in and reports through
                         * prv2_last_kernel(); the other entry points do not take it                  */
};

int prv2_abi_version(void);
const char* prv2_last_error(void);
/* Name of the device kernel the calling thread's last successful prv2_conv2d dispatched to, as "name<BN,prec>"
 * (e.g. "conv3x3_halo16_kernel<128,bf16x3>", "gemm16_kernel<64,bf16x3>"): what bench.py's roofline attributes time to.
 * Which kernel runs a layer depends on the layer's shape PER IMAGE, never on the batch size (so results do not depend on
 * how tiles are batched).  When a call issues two kernels (f32 mode: 3x3 tiles + remainder strip) it names the last. */
const char* prv2_last_kernel(void);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear layer with fused epilogue.
 * Replaces nn.Conv2d / nn.Linear / nn.ConvTranspose2d(k==stride) on the path:
 *   fusion convs     estimator/models/blocks/convs.py:39-41,70; bi_directional_fusion_model.py:40-51
 *   DPT head convs   external/depth_anything_v2/dpt.py:51-81, util/blocks.py:45-47
 *   ViT linears      external/depth_anything_v2/dinov2_layers/attention.py:44,46; mlp.py:30,32
 *
 *   y[m, n] = epilogue( sum_{ky,kx,c} pre(x[pix(m) + (ky,kx), c]) * w[n, ky, kx, c] )
 *   epilogue: v = acc + bias[n]; [v = LN_row(v) * ln_weight[n] + ln_bias[n]]; v = act(v); v *= gamma[n];
 *             v = mul[m,n] * v; v += res[m,n] + res2[m,n]
 *   (every pointer optional).  pre = ReLU when relu_in != 0.  Zero padding.
 *   LN_row = the channels-first LayerNorm of the fusion convs (convs.py:21-29, biased variance, eps ln_eps) over the
 *   cout channels of the pixel, fused when cout <= 128 (the whole row sits in one workgroup tile).
 *
 * Weights are pre-packed by prv2_pack_conv_weight(): [cout_pad][kh*kw][cin_pad] with
 * cin_pad = roundup(cin, 32), cout_pad = roundup(cout, 128), zero filled.
 * convt_k > 0 turns the call into ConvTranspose2d(kernel=stride=convt_k): kh=kw=1, the GEMM has
 * convt_k^2 * cout columns ordered (ky, kx, co) and each column block is scattered to
 * output pixel (y*k+ky, x*k+kx).
 * ------------------------------------------------------------------------------------------ */
typedef struct prv2_conv_desc {
  int32_t n, h, w;       /* input batch / spatial size                                   */
  int32_t cin, cout;
  int32_t kh, kw, stride, pad;
  int32_t ldx, ldy;      /* pixel strides (floats)                                       */
  int64_t x_bstride;     /* image stride of x in floats (0 => h*w*ldx)                   */
  int64_t y_bstride;     /* image stride of y in floats (0 => oh*ow*ldy)                 */
  int32_t relu_in;       /* apply ReLU to x while loading                                */
  int32_t act;           /* enum prv2_act                                                */
  int32_t convt_k;       /* 0, or k for ConvTranspose2d(k, stride=k)                     */
  int32_t ld_mul, ld_res, ld_res2;
  int32_t prec;          /* enum prv2_prec                                               */
  int32_t force_generic; /* != 0: never use the LDS-halo 3x3 kernel (tests / A-B)        */
  float ln_eps;          /* epsilon of the fused LayerNorm (used when ln_weight != NULL) */
  int32_t part;          /* f32 mode: 3x3 halo convs whose width is 32k + (1..8) run as 32-pixel tiles + a remainder strip
                          * (generic kernel): 0 = both (default), 1 = the tiles only, 2 = the strip only (per-kernel timing).
                          * The bf16 modes do tiles and strip in one launch: part must be 0.                             */
  int32_t same_pad;      /* != 0: TensorFlow "SAME" padding as timm's Conv2dSame (the reference's stem surgery builds one,
                          * patchrefinerplus.py:152-158): out = ceil(in / stride), total = max((out-1)*stride + k - in, 0),
                          * low = total / 2, the rest goes to the high side; ``pad`` is ignored.  Not for convt_k.          */
  int32_t fmt;           /* PRV2_FMT_* bits: operands in the pre-split "X2" activation format -- per 8 channels
                          * [8 x bf16 hi | 8 x bf16 lo] (32 bytes = the bytes of 8 floats; hi = RNE bf16 of v, lo = RNE bf16 of v - hi),
                          * i.e. exactly what the bf16x3 kernels' loaders make of an fp32 input, stored by the producer so that the
                          * consumer only copies.  Taken by the 256-column kernels (prv2_conv3x3_ln_gate / prv2_conv2d layers it routes
                          * there): X_X2 + MUL_X2 by the gate kernel, Y_X2 by the plain conv without LayerNorm.  A layer's result does not
                          * depend on the format (``mul`` counts as hi + lo either way).  0 everywhere else.                        */
} prv2_conv_desc;
#define PRV2_FMT_X_X2 1   /* x is X2 (gate kernel: the GatedConvUnit's [out | coarse ROI] concat)            */
#define PRV2_FMT_MUL_X2 2 /* mul is X2 (gate kernel: ``out``, the first half of that concat; goes together with X_X2) */
#define PRV2_FMT_Y_X2 4   /* y is written X2 (256-column conv, no LayerNorm: GatedConvUnit.conv writing ``out``) */

/* host-side helpers: sizes of the packed weight buffers (in bytes) */
int64_t prv2_packed_weight_bytes(int32_t cout, int32_t cin, int32_t kh, int32_t kw, int32_t convt_k, int32_t prec);

/* w_src: device fp32, PyTorch layout [cout][cin][kh][kw] (Conv2d / Linear with kh=kw=1) or
 * [cin][cout][k][k] when convt_k>0 (ConvTranspose2d).  bn_scale (optional, [cout]) is folded in. */
int prv2_pack_conv_weight(const float* w_src, const float* bn_scale, void* w_packed, int32_t cout, int32_t cin,
                          int32_t kh, int32_t kw, int32_t convt_k, int32_t prec, void* stream);

int prv2_conv2d(const prv2_conv_desc* d, const float* x, const void* w_packed, const float* bias,
                const float* ln_weight, const float* ln_bias, const float* gamma, const float* mul, const float* res,
                const float* res2, float* y, void* stream);

/* 3x3 / stride 1 / pad 1 convolution whose first ``channels`` input channels are a bilinear(align_corners=True) UPSAMPLE of a
 * low-resolution tensor, interpolated while the conv stages its input tile -- the upsampled tensor is never written:
 *   UpSample.forward_hardcode   estimator/models/blocks/fusion_model.py:15-24   x = cat[interpolate(x1, size, 'bilinear',
 *                               align_corners=True), x2, pred1, pred2] -> DoubleConv: the first conv reads channels [0, c1) from x1
 *   C2FModule output_conv1      bi_directional_fusion_model.py:139-142,201      conv3x3(interpolate(path_1, scale 2, align_corners))
 * x (NHWC, d->ldx) supplies the channels [channels, d->cin) at their usual offsets (its first ``channels`` channels are not
 * read; x may equal u->x when channels == d->cin).  Same arithmetic, bit for bit, as prv2_upsample_bilinear into x followed by
 * prv2_conv2d.  Contract (prv2_conv2d_ups_supported(d, u) != 0): bf16 modes, 3x3 s1 p1, cout > 64 and != 256-with-cin%32==0 routing
 * aside the layer must be one the 128-column halo kernel takes (width >= 24 ...), channels % 32 == 0, 0 < channels <= cin,
 * u->ld % 4 == 0, 16-byte aligned, no input ReLU. */
typedef struct prv2_ups_src {
  const float* x;        /* low-resolution source, NHWC [n, h, w, ld], channels [0, channels) used */
  int32_t h, w, ld;
  int32_t channels;
  int64_t bstride;       /* image stride in floats (0 => h*w*ld) */
} prv2_ups_src;
int prv2_conv2d_ups_supported(const prv2_conv_desc* d, const prv2_ups_src* u);
int prv2_conv2d_ups(const prv2_conv_desc* d, const float* x, const prv2_ups_src* u, const void* w_packed, const float* bias,
                    const float* ln_weight, const float* ln_bias, const float* res, float* y, void* stream);

/* 3x3 / stride 1 / pad 1 convolution of a bilinear(align_corners=True) UPSAMPLE (factor >= 5/3) of u, computed at u's resolution
 * (csrc/upconv.hip) -- the layers prv2_conv2d_ups interpolates inside its loader, with 2.3x fewer matrix operations:
 *   C2FModule output_conv1      bi_directional_fusion_model.py:139-142,201   conv3x3(interpolate(path_1, scale 2, align_corners=True))
 *   UpSample.forward_hardcode   fusion_model.py:15-24                        the interpolate(x1) part of DoubleConv.0(cat[x1, x2, pred1, pred2])
 * The conv is linear and its input an interpolation of u, so  conv3x3(up(u); W)(p) = sum_tap [p + d_tap inside] Bil(G_tap; s(p + d_tap))
 * with G_tap = W[:, :, tap] . u at LOW resolution (MFMA) and the 36 corner terms per output gathered on the VALU.
 *     y[n, h, w, 0 .. cout) = act(conv3x3(interpolate(u, (h, w), 'bilinear', align_corners=True)) + bias + add)
 * w_packed: prv2_pack_conv_weight(cout, cin = u->channels, 3, 3) -- for a conv over a concat [up(u) | rest], the weight columns of
 * the upsampled part; ``add`` (NHWC [n, h, w, ld_add >= cout], or NULL) is then prv2_conv2d of ``rest`` with the other columns,
 * activation NONE -- it may be y itself (every output element is read by the thread that writes it).
 * Same split products and fp32 accumulation as the other bf16 kernels; the taps / corners are summed in another order than
 * upsample -> conv (fp32-grade, not bit-identical).  Contract (prv2_upconv3x3_supported != 0): bf16 modes, u->channels % 32 == 0,
 * source step (u->h - 1) / (h - 1) and (u->w - 1) / (w - 1) <= 0.6 pixel (16 x 28 output tiles up to 1/2 -- the x2 upsamples --,
 * 14 x 24 up to 3/5: DepthAnything's 256 -> 448 head resolution), 16-byte aligned NHWC rows. */
int prv2_upconv3x3_supported(const prv2_ups_src* u, int32_t n, int32_t h, int32_t w, int32_t cout, int32_t prec);
int prv2_upconv3x3(const prv2_ups_src* u, const void* w_packed, const float* bias, const float* add, int32_t ld_add, int32_t n, int32_t h,
                   int32_t w, int32_t cout, int32_t act, int32_t prec, float* y, int32_t ldy, int64_t y_bstride, void* stream);

/* TWO back-to-back 3x3 convs (no activation in between) of a bilinear(align_corners=True) x2 UPSAMPLE as ONE 5x5 conv computed at u's
 * resolution (csrc/upconv5.hip) -- C2FModule's ``output_conv2[0] o output_conv1 o interpolate`` (with refinenet1.out_conv folded into
 * output_conv1): bi_directional_fusion_model.py:139-146 (interpolate + out_conv), :169-173 (definitions), :201-203 (use):
 *     v = act(conv3x3(conv3x3(up(u); W1) + b1(p); W2) + b2)  =  act(conv5x5(up(u); Weff) + bias_map(p) - ring_fix(p)),   Weff[d] = sum_{d1+d2=d} W2[d2] W1[d1]
 * 205 k instead of 332 k MACs per output pixel in the reference's terms; the 128-channel full-resolution intermediate is never formed.
 *   prv2_upconv5x5       y = act(conv5x5(interpolate(u, (h, w))) + bias_map[class(y)][class(x)]) for every pixel EXCEPT the one-pixel border
 *                        ring, which is written WITHOUT the activation (the ring fix below finishes it).  w_packed:
 *                        prv2_pack_conv_weight(cout, cin = u->channels, 5, 5) of Weff; bias_map: device fp32 [5][5][cout], row / column
 *                        classes (0, 1, interior, h - 2, h - 1): the inner conv's (position dependent) bias seen through the outer conv's
 *                        zero padding -- data independent.  cout <= 32.
 *   prv2_upconv5x5_lines lines[n][2 uw + 2 uh][channels] = the border lines of up(u) at u's resolution: output row 0, row h - 1 (uw positions
 *                        each), column 0, column w - 1 (uh each) -- align_corners samples of u's first / last rows and columns.
 *   prv2_upconv5x5_ring  the 5x5 form over the zero-padded map includes inner-conv outputs one pixel OUTSIDE the image, which the outer conv's
 *                        zero padding hides: on the ring  y = act(y - fix),  fix = per edge a 1-D five-tap conv of the border line -- as tap
 *                        GEMMs at u's resolution: g_edges[n][2 uw + 2 uh][ldg] = prv2_conv2d (1x1) of ``lines`` with the weights
 *                        [(edge * 7 + j) * cout + c][ci], edge = top, bottom, left, right; j < 5: sum_{k1 + k2 - 2 = j - 2} W2[edge's outer taps k2]
 *                        W1[edge's inner taps k1]; j = 5 / 6: the corner term at the row edges' first / last pixel (the outside corner position
 *                        is in both edges' sums).  tools/studies/composite5x5_ring.py: the float64 algebra (4e-15).
 * Contract (prv2_upconv5x5_supported != 0): bf16 modes, u->channels % 32 == 0, cout % 4 == 0 and <= 32, source step <= 1/2 (x2 upsamples),
 * h, w >= 5, 16-byte aligned NHWC rows.  fp32-grade, not bit-identical to the two-conv sequence. */
int prv2_upconv5x5_supported(const prv2_ups_src* u, int32_t n, int32_t h, int32_t w, int32_t cout, int32_t prec);
int prv2_upconv5x5(const prv2_ups_src* u, const void* w_packed, const float* bias_map, int32_t n, int32_t h, int32_t w, int32_t cout,
                   int32_t act, int32_t prec, float* y, int32_t ldy, int64_t y_bstride, void* stream);
int prv2_upconv5x5_lines(const prv2_ups_src* u, int32_t n, int32_t h, int32_t w, float* lines, void* stream);
int prv2_upconv5x5_ring(float* y, int32_t ldy, int64_t y_bstride, int32_t n, int32_t h, int32_t w, int32_t cout, const float* g_edges,
                        int32_t ldg, int32_t uh, int32_t uw, int32_t act, void* stream);

/* prv2_conv2d (3x3 / stride 1 / pad 1, bias, [LayerNorm,] activation, [+ res]) that ALSO writes the two depth maps every fusion
 * level appends to its features behind its own output channels: y[pixel][cout .. cout + 3] = (p1, p2, 0, 0), p1 / p2 dense
 * [n, ph, pw] resized bilinear(align_corners=True) to the output size -- the ``torch.cat([f, pred1, pred2])`` of
 * fusion_model.py:91-118 / bi_directional_fusion_model.py:424-436 closed by the conv that fills the rest of the row, instead of a
 * prv2_depth_pair_fill launch of 16-byte stores into 400-byte rows.  Same arithmetic as prv2_conv2d + prv2_depth_pair_fill.
 * Contract (prv2_conv2d_tail_supported(d) != 0): bf16 modes, 3x3 s1 p1, width >= 24, height >= 4, cout % 4 == 0 and <= 128 (the
 * widths whose LayerNorm the halo kernels fuse), ldy >= cout + 4, LayerNorm or residual but not both. */
int prv2_conv2d_tail_supported(const prv2_conv_desc* d);
int prv2_conv2d_tail(const prv2_conv_desc* d, const float* x, const void* w_packed, const float* bias, const float* ln_weight,
                     const float* ln_bias, const float* res, const float* p1, const float* p2, int32_t ph, int32_t pw, float* y,
                     void* stream);

/* GatedConvUnit tail in one kernel (estimator/models/blocks/bi_directional_fusion_model.py:44-51 ``fusion_conv`` =
 * Conv3x3(2F -> F) . LayerNorm(channels_first) . ReLU . Conv1x1(F -> F) . Sigmoid, and :70-80 ``out * fusion (+ xs[0])``):
 *     fused = act(LN(conv3x3(x) + bias))                              (d->act: NONE or RELU; ln_eps from d)
 *     y     = mul * sigmoid(conv1x1(fused) + gate_bias) (+ res)
 * gate_w_packed == NULL: y = act(LN(conv3x3(x) + bias)) -- the 256-channel conv with the LayerNorm fused (prv2_conv2d fuses it
 * for cout <= 128 only); mul / res / gate_bias must then be NULL and d->act may be any activation.
 * Shape contract (prv2_conv3x3_ln_gate_supported(d) != 0): 3x3, stride 1, pad 1, cin % 32 == 0, bf16 modes; cout == 256 (F of the
 * refinenets: 8 x 16-pixel x 256-channel tiles, width >= 16), or -- with gate weights -- cout == 128 / 32 (the full-resolution
 * output_conv2_fusion block of the DepthAnything / ZoeDepth fusion configs: 8 x 32-pixel tiles, width >= 24, height >= 4); x / y / mul / res NHWC fp32 with pixel strides ldx / ldy / ld_mul / ld_res (multiples of 4, 16-byte aligned).
 * w_packed: prv2_pack_conv_weight image of the 3x3 weights; gate_w_packed: prv2_pack_gate_weight image (prv2_gate_weight_bytes(cout)
 * bytes) of the 1x1 weights [cout][cout].  Same arithmetic as the unfused sequence conv2d -> layernorm -> conv2d(1x1, sigmoid, mul,
 * res) (split products, accumulation order); the row statistics are reduced in a different (fixed) order. */
int prv2_conv3x3_ln_gate_supported(const prv2_conv_desc* d);
int64_t prv2_gate_weight_bytes(int32_t channels);
int prv2_pack_gate_weight(const float* w_src, void* w_packed, int32_t cout, int32_t cin, void* stream);
int prv2_conv3x3_ln_gate(const prv2_conv_desc* d, const float* x, const void* w_packed, const float* bias, const float* ln_weight,
                         const float* ln_bias, const void* gate_w_packed, const float* gate_bias, const float* mul, const float* res,
                         float* y, void* stream);

/* prv2_conv3x3_ln_gate with a pre-LayerNorm addend: fused = act(LN(conv3x3(x) + bias + pre)), pre NHWC fp32 [n, h, w, ld_pre >= cout]
 * (ld_pre % 4 == 0, 16-byte aligned) -- the coarse half of the unit's ``fusion_conv.0`` from prv2_coarse_tap_gather, x being the fine
 * half only (cin = F instead of 2F).  pre == NULL: prv2_conv3x3_ln_gate. */
int prv2_conv3x3_ln_gate_pre(const prv2_conv_desc* d, const float* x, const void* w_packed, const float* bias, const float* pre,
                             int32_t ld_pre, const float* ln_weight, const float* ln_bias, const void* gate_w_packed, const float* gate_bias,
                             const float* mul, const float* res, float* y, void* stream);

/* prv2_conv2d (3x3 / stride 1 / pad 1) with an addend in front of the epilogue: v = acc + pre[m, n] + bias[n]; [LayerNorm;] act; [+ res]
 * -- ``fusion_layers_1[l](cat([c, f]))`` (bi_directional_fusion_model.py:424-426; FusionUnet encoder_layers_1, fusion_model.py:91-95)
 * run over the fine half f only, with the coarse half of the conv from prv2_coarse_tap_gather as ``pre`` [n, h, w, ld_pre >= cout].
 * Contract (prv2_conv2d_pre_supported(d) != 0): bf16 modes, 3x3 s1 p1, width >= 24, height >= 4, cout % 4 == 0 (the layers the
 * 16x16x32 halo kernels or the 256-column kernel run); a fused LayerNorm needs cout <= 128 or == 256. */
int prv2_conv2d_pre_supported(const prv2_conv_desc* d);
int prv2_conv2d_pre(const prv2_conv_desc* d, const float* x, const void* w_packed, const float* bias, const float* pre, int32_t ld_pre,
                    const float* ln_weight, const float* ln_bias, const float* res, float* y, void* stream);

/* ------------------------------------------------------------------------------------------
 * The 32-channel FULL-RESOLUTION tail of BiDirectionalFusion: two consecutive 3x3 convs with everything between and behind them in
 * ONE kernel (csrc/chain32.hip; 8 x 16-pixel tiles, the first conv's output stays in LDS; bf16x3 arithmetic):
 *   prv2_chain32_c2f   C2FModule ``output_conv2_fusion`` (GatedFusionBlock with one input, upscale=False) + ``output_conv3``
 *                      bi_directional_fusion_model.py:56-82,116-146 (unit / block), :171-180 (definitions), :203-204 (use)
 *        o     = conv3x3(relu(x); W1) + b1 + x                            GateresConfUnit2.conv + skip_add
 *        f     = relu(LN(conv3x3(o; W2) + b2 + pre))                      fusion_conv.0-.2 over cat([o, c_feat]); pre = its coarse half
 *        y     = Wo (o * sigmoid(Wg f + bg)) + bo                         fusion_conv.3, gate, out_conv         -> y  [n, h, w, 32]
 *        depth = w3 . y + b3                                              output_conv3 (1x1 -> 1)              -> depth [n, h, w]
 *   prv2_chain32_enc   ``fusion_layers_1[0]`` + ``fusion_layers_2[0]`` (SingleConvCNNLN)   bi_directional_fusion_model.py:424-431
 *        f = gelu(LN(conv3x3(x; W1) + b1 + pre))                          over cat([c, x]); pre = the coarse half (prv2_coarse_tap_gather)
 *        y = gelu(LN(conv3x3(cat([f, p1, p2]); W2) + b2))                 p1 / p2: dense [n, h, w] depth maps at the level's size
 * Weight images: prv2_pack_chain32_weight(w_src [32][cin_total][taps] fp32 PyTorch layout, kind) -> prv2_chain32_weight_bytes(kind, taps)
 *   kind 0: the FIRST conv (taps 9; input channels [0, 32) of w_src);  kind 1: the second conv (taps 9) and the 1x1 gate / out_conv
 *   (taps 1) -- K in accumulator order;  kind 2: the [p1 | p2] tail of a 3x3 over 34 channels (channels 32, 33 of w_src).
 *   prv2_chain32_c2f: w1 kind 0, w2 kind 1, wg / wo kind 1 (taps 1);  prv2_chain32_enc: w1 kind 0, w2 kind 1, wg = kind 2 of W2.
 * consts: device fp32 [9][32] = b1, LN1 weight, LN1 bias, b2, bg, bo, w3, LN2 weight, LN2 bias (rows a mode does not use: zeros);
 *   LN1 is the chain's FIRST LayerNorm (c2f: behind the second conv; enc: behind the first), LN2 enc's second.
 * Same split products / fp32 accumulation as prv2_conv2d's bf16x3 kernels; not bit-identical to the unfused sequence (LayerNorm sums
 * in another order; ``o`` enters the gate product as hi + lo, as ``mul`` does in prv2_conv3x3_ln_gate).
 * ------------------------------------------------------------------------------------------ */
typedef struct prv2_chain32_desc {
  const float* x;        /* NHWC input [n, h, w, ldx >= 32]                                */
  const void* w1;
  const void* w2;
  const void* wg;
  const void* wo;        /* c2f only                                                       */
  const float* consts;
  const float* pre;      /* [n, h, w, ld_pre >= 32] (c2f: may be NULL)                     */
  const float* p1;       /* enc only                                                       */
  const float* p2;
  float* y;              /* NHWC output [n, h, w, ldy >= 32] (a channel slice is fine)     */
  float* depth;          /* c2f only (may be NULL)                                         */
  int64_t x_bstride;     /* image strides in floats (0 => h*w*ld)                          */
  int64_t y_bstride;
  int32_t n, h, w;
  int32_t ldx, ldy, ld_pre;
  float b3, ln_eps;
} prv2_chain32_desc;
int64_t prv2_chain32_weight_bytes(int32_t kind, int32_t taps);
int prv2_pack_chain32_weight(const float* w_src, int32_t cin_total, int32_t taps, int32_t kind, void* w_packed, void* stream);
int prv2_chain32_c2f(const prv2_chain32_desc* d, void* stream);
int prv2_chain32_enc(const prv2_chain32_desc* d, void* stream);

/* The 256-channel 3x3 conv of a GatedConvUnit / ResidualConvUnit in the fp16 + block-scaled-fp6 arithmetic (csrc/conv3x3_f6.hip;
 * PRV2_PREC_F16F6) -- estimator/models/blocks/bi_directional_fusion_model.py:40-43 (self.conv = ReLU, Conv2d 3x3), :58-64 (out =
 * self.conv(x) + x); the same layer prv2_conv2d runs in bf16x3 on conv3x3_c256_kernel:
 *     y[n, h, w, 0 .. 256) = out_scale * conv3x3(q(relu?(x) * x_scale); q(W * w_scale)) + bias + res        (fp32 NHWC, or X2 with d->fmt = PRV2_FMT_Y_X2)
 * where each product x w is taken as f16(x) f16(w) + q6(x) q6(w - f16 w) + q6(x - f16 x) q6(w), q6 = fp6 e2m3 with one power-of-two
 * scale per 32 channels, fp32 accumulation: two v_mfma_f32_16x16x32_f16 and one v_mfma_scale_f32_16x16x128_f8f6f4 per 64 channels and tap
 * instead of six bf16 MFMAs.  rms error of a dot product against float64 1.2e-5 (bf16x3: 4.4e-6) -- fp32-grade, wider than TF32.
 * x_scale / w_scale: powers of two chosen by the caller so that |relu(x) x_scale| and |W w_scale| stay inside fp16's range (values
 * beyond 65504 are clamped in the fp16 part and carried by the fp6 residual: finite, imprecise); out_scale = 1 / (x_scale w_scale).
 * range_word (device, or NULL): receives atomicMax of the float bits of max |relu(x) x_scale| the launch saw -- the caller's range monitor.
 * Contract (prv2_conv3x3_f6_supported(d) != 0): 3x3 s1 p1, cout = 256, cin % 64 == 0, width >= 16, d->act = NONE, no LayerNorm / gate /
 * mul; d->relu_in and d->ld_res as for prv2_conv2d.  w_packed: prv2_pack_conv3x3_f6_weight (PyTorch [256][cin][3][3] fp32). */
int prv2_conv3x3_f6_supported(const prv2_conv_desc* d);
int64_t prv2_conv3x3_f6_weight_bytes(int32_t cout, int32_t cin);
int prv2_pack_conv3x3_f6_weight(const float* w_src, float w_scale, void* w_packed, int32_t cout, int32_t cin, void* stream);
int prv2_conv3x3_f6(const prv2_conv_desc* d, const float* x, const void* w_packed, const float* bias, const float* res, float x_scale,
                    float out_scale, uint32_t* range_word, float* y, void* stream);
/* The GatedConvUnit tail with its 3x3 conv in the same fp16 + fp6 arithmetic (csrc/conv3x3_f6.hip: conv3x3_c256_gate_f6_kernel) -- what
 * prv2_conv3x3_ln_gate_pre computes on conv3x3_c256_gate_x2_kernel (bi_directional_fusion_model.py:44-51, 70-80):
 *     y = mul * sigmoid(conv1x1(act(LN(out_scale * conv3x3(q(x * x_scale); q(W * w_scale)) + bias + pre))) + gate_bias) (+ res)
 * The LayerNorm, the 256 x 256 gate GEMM (bf16x3) and the final stage are conv3x3_gate.hip's epilogue, unchanged (csrc/conv3x3_gate_epi.h).
 * Contract: prv2_conv3x3_f6_supported(d); x and mul in ONE format -- pre-split (d->fmt = PRV2_FMT_X_X2 | PRV2_FMT_MUL_X2: the unit's ``out`` / its X2 concat) or
 * fp32 (d->fmt = 0: the [out | coarse ROI] concat of the configs whose ROI gather resizes, mul = its first half) --, no input ReLU, fp32 y;
 * ln_weight / ln_bias / gate_w_packed (prv2_pack_gate_weight) required; pre (or NULL) as for prv2_conv3x3_ln_gate_pre; w_packed:
 * prv2_pack_conv3x3_f6_weight of the conv's weights over what x holds (the fine half W[:, :256] with ``pre``, all 512 input channels of the concat form). */
int prv2_conv3x3_ln_gate_f6(const prv2_conv_desc* d, const float* x, const void* w_packed, const float* bias, const float* pre, int32_t ld_pre,
                            const float* ln_weight, const float* ln_bias, const void* gate_w_packed, const float* gate_bias, const float* mul,
                            const float* res, float x_scale, float out_scale, uint32_t* range_word, float* y, void* stream);

/* Convolution with ONE output channel (direct, HBM-bound):
 *   final_conv 3x3 -> 1 + clamp(update_base + offset, 0)   bi_directional_fusion_model.py:438-442, fusion_model.py:113-118
 *   output_conv2.2 1x1 32->1 + Sigmoid * max_depth          external/depth_anything_v2/dpt.py:111-113,190
 *   output_conv3 1x1 32->1                                   bi_directional_fusion_model.py:178-179
 * w: device fp32 PyTorch layout [1][cin][k][k].  y = post( act(conv + bias) * scale + res ), post = max(.,0) if clamp0.
 * y / res are dense [n, h, w] (ld 1). */
int prv2_conv2d_cout1(const float* x, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t ldx, const float* wgt,
                      int32_t k, const float* bias, int32_t act, float scale, const float* res, int32_t clamp0,
                      float* y, void* stream);

/* Depthwise kxk convolution, k in {3, 5, 7}, pad k/2 (+folded BatchNorm bias, optional ReLU) for the refiner encoders
 * (timm, un-vendored; lightweight_refiner.py:260-262,296): MobileNetV4 (3x3 / 5x5) and ConvNeXt (7x7, conv_dw).
 * w: device, TAP-MAJOR [k*k][c] with the BN scale already folded (so that a lane's 4 channels are one float4); bias [c]. */
int prv2_dwconv2d(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ldx, const float* wgt,
                  const float* bias, int32_t k, int32_t stride, int32_t relu, float* y, int32_t ldy, void* stream);
/* The same with any enum prv2_act and, when same_pad != 0, TensorFlow "SAME" padding (see prv2_conv_desc.same_pad):
 * the depthwise convs of timm's tf_efficientnet_* (Conv2dSame; v2_eff_u4k.py:94).  Output ceil(h/stride) x ceil(w/stride). */
int prv2_dwconv2d_ex(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ldx, const float* wgt,
                     const float* bias, int32_t k, int32_t stride, int32_t act, int32_t same_pad, float* y, int32_t ldy,
                     void* stream);

/* Squeeze-and-excitation pieces of the EfficientNet refiner encoder (timm SqueezeExcite, un-vendored):
 *   prv2_global_avgpool: out[n][c] = mean over the h*w pixels of image n (x.mean((2, 3)));  out is dense [n][c];
 *                        workspace: prv2_global_avgpool_workspace_floats(n, hw, c) floats (partial sums of the two-stage,
 *                        fixed-order -- run-to-run deterministic -- reduction).
 *   prv2_se_gate:        g[n][:] = sigmoid(W2 silu(W1 mean[n][:] + b1) + b2), the conv_reduce / conv_expand bottleneck in fp32;
 *                        w1 [cse][c] (PyTorch layout), w2t [cse][c] = conv_expand's weight TRANSPOSED, biases optional,
 *                        workspace n*cse floats.
 *   prv2_channel_scale:  x[n, pix, c] *= s[n][c]  in place (x * gate). */
int64_t prv2_global_avgpool_workspace_floats(int32_t n, int64_t hw, int32_t c);
int prv2_global_avgpool(const float* x, int32_t n, int64_t hw, int32_t c, int32_t ldx, float* out, float* workspace, void* stream);
int prv2_se_gate(const float* mean, int32_t n, int32_t c, const float* w1, const float* b1, int32_t cse, const float* w2t,
                 const float* b2, float* g, float* workspace, void* stream);
int prv2_channel_scale(float* x, int32_t n, int64_t hw, int32_t c, int32_t ldx, const float* s, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row LayerNorm (+activation).  Rows of ``c`` contiguous floats with strides ldx / ldy.
 * Replaces nn.LayerNorm(eps=1e-6) on tokens (dinov2.py:95, block.py:56,68) and -- because the
 * activations are NHWC -- the channels-first LayerNorm of the fusion convs (convs.py:21-29)
 * including the GELU / ReLU that follows it (convs.py:70-72, bi_directional_fusion_model.py:49-50).
 * ------------------------------------------------------------------------------------------ */
int prv2_layernorm(const float* x, int64_t rows, int32_t c, int32_t ldx, const float* weight, const float* bias,
                   float eps, int32_t act, float* y, int32_t ldy, void* stream);

/* ------------------------------------------------------------------------------------------
 * ViT token plumbing (external/depth_anything_v2/dinov2.py:212-231, patch_embed.py:69-82)
 * ------------------------------------------------------------------------------------------ */
/* img NHWC [b, gh*p, gw*p, 3] (ld 3..) -> rows [b*gh*gw, ldo] with (ky, kx, c) column order, zero padded to ldo */
int prv2_patchify(const float* img, int32_t b, int32_t gh, int32_t gw, int32_t p, int32_t ldi, float* rows,
                  int32_t ldo, void* stream);
/* tokens[b, 0] = cls + pos[0]; tokens[b, 1+i] = emb[b, i] + pos[1+i] */
int prv2_assemble_tokens(const float* emb, const float* cls, const float* pos, int32_t b, int32_t np, int32_t dim,
                         float* tokens, void* stream);
/* softmax((q*scale) k^T) v per (batch, head); qkv rows are [3][heads][hd] as produced by the qkv
 * Linear (attention.py:49-62).  hd must be 64.  out rows are [heads][hd]. */
int prv2_attention(const float* qkv, int32_t b, int32_t ntok, int32_t heads, int32_t hd, float* out, int32_t prec,
                   void* workspace, int64_t workspace_bytes, void* stream);
/* The same with an additive score bias shared by the batch: softmax((q*scale) k^T + bias[head]) v -- the relative position
 * bias of the MiDaS BEiT blocks (torch.hub MiDaS midas/backbones/beit.py attention_forward; called by
 * external/zoedepth/models/base_models/midas.py:267).  bias: [heads][ntok][ld_bias] fp32, ld_bias >= roundup(ntok, 64),
 * a multiple of 4, rows 16-byte aligned (pad keys are never read past ntok's 64-key tile and are masked).  bias == NULL: no bias. */
/* The bias of a (model, resolution) is a constant: prv2_pack_attention_bias re-orders its rows once into the image the bf16x3 kernel
 * reads with coalesced loads (per head, block of 32 queries and tile of 64 keys: the values in accumulator order, pre-multiplied by
 * log2 e -- the product the kernel otherwise forms per tile; same bits).  Passed to prv2_attention_bias / prv2_attention_ss as
 * ``bias`` with ld_bias == PRV2_ATTENTION_BIAS_IMAGE (bf16 modes). */
#define PRV2_ATTENTION_BIAS_IMAGE (-1)
int64_t prv2_attention_bias_image_bytes(int32_t heads, int32_t ntok);
int prv2_pack_attention_bias(const float* bias, int32_t heads, int32_t ntok, int32_t ld_bias, float* image, void* stream);
int prv2_attention_bias(const float* qkv, int32_t b, int32_t ntok, int32_t heads, int32_t hd, const float* bias, int32_t ld_bias,
                        float* out, int32_t prec, void* workspace, int64_t workspace_bytes, void* stream);
/* ------------------------------------------------------------------------------------------
 * Split-swizzled ("ss") operand format and the dense layers that consume it (csrc/gemm_ss.hip) -- the ViT blocks'
 * Linear layers (attention.py:44,46; mlp.py:30,32) at large token counts, PRV2_PREC_BF16X3 only.
 * A row of C channels (C % 32 == 0) occupies 4*C bytes like fp32: per 32 channels one 128-byte group of eight 16-byte
 * slots, logical slot s = 0..3: bf16 hi of channels 8s..8s+7, s = 4..7: bf16 lo of channels 8(s-4)..; stored at
 * slot s ^ ((row >> 1) & 7) (the image prv2_pack_conv_weight gives the weights).  Producers write it directly:
 *   prv2_split_ss      fp32 rows -> ss rows
 *   prv2_layernorm_ss  prv2_layernorm with an ss output (no activation)
 *   prv2_attention_ss  prv2_attention_bias (bf16x3) with an ss output
 *   prv2_gemm_ss       y = epilogue(A_ss W^T): t = act(acc + bias[n]); t *= gamma[n]; t += res[m, n]; fp32 rows (y) or ss rows
 *                      (y_ss; then no gamma / res).  w_packed: prv2_pack_conv_weight(.., kh = kw = 1, PRV2_PREC_BF16X3).
 * prv2_gemm_ss is bit-identical to prv2_conv2d's 1x1 path on the same values (same split, products, order, epilogue).
 * ------------------------------------------------------------------------------------------ */
int prv2_split_ss(const float* x, int64_t rows, int32_t c, int32_t ldx, void* y_ss, void* stream);
int prv2_layernorm_ss(const float* x, int64_t rows, int32_t c, int32_t ldx, const float* weight, const float* bias, float eps,
                      void* y_ss, void* stream);
int prv2_attention_ss(const float* qkv, int32_t b, int32_t ntok, int32_t heads, int32_t hd, const float* bias, int32_t ld_bias,
                      void* out_ss, void* workspace, int64_t workspace_bytes, void* stream);
int prv2_gemm_ss(const void* a_ss, int64_t m, int32_t k, const void* w_packed, int32_t n, const float* bias, const float* gamma,
                 const float* res, int32_t ld_res, int32_t act, float* y, int32_t ldy, void* y_ss, void* stream);
/* The attention block without a pre-pass (attention.py:49-62; MiDaS beit.py attention_forward): the qkv Linear writes its [q | k | v] rows
 * split-swizzled with the q third (columns [0, q_cols)) multiplied by q_scale = hd^-0.5 log2 e behind the bias, and prv2_attention_qkv_ss reads
 * those rows directly (q / k fragments are 16-byte copies; v is transposed by the LDS read).  Same values as prv2_gemm_ss (fp32 rows) ->
 * prv2_attention_ss, bit for bit; no workspace, no fp32 qkv tensor.  Exactly one of out (fp32 rows [b * ntok, heads * hd]) / out_ss. */
int prv2_gemm_ss_qkv(const void* a_ss, int64_t m, int32_t k, const void* w_packed, int32_t n, const float* bias, int32_t q_cols, float q_scale,
                     void* qkv_ss, void* stream);
int prv2_attention_qkv_ss(const void* qkv_ss, int32_t b, int32_t ntok, int32_t heads, int32_t hd, const float* bias, int32_t ld_bias, float* out,
                          void* out_ss, void* stream);

/* device scratch the split-bf16 attention needs (pre-split q/k rows + transposed v planes); 0 for PRV2_PREC_F32.
 * The workspace must be 256-byte aligned; its contents are dead when the call returns (stream order). */
int64_t prv2_attention_workspace_bytes(int32_t b, int32_t ntok, int32_t heads, int32_t prec);

/* ------------------------------------------------------------------------------------------
 * ZoeDepth metric-bins head, elementwise parts (the 1x1 MLPs go through prv2_conv2d)
 * ------------------------------------------------------------------------------------------ */
/* y = a + b over [rows, c] with row strides (x + prev_b_embedding, external/zoedepth/models/layers/attractor.py:178) */
int prv2_add(const float* a, int32_t lda, const float* b, int32_t ldb, int64_t rows, int32_t c, float* y, int32_t ldy,
             void* stream);

/* Zero the pad channels [c, ld) of every pixel of an NHWC buffer (rows = n*h*w).  Buffers whose channel count is
 * not a multiple of 4 (the [feat | pred1 | pred2] concats of fusion_model.py:91-118: 34, 66, 98 ...) carry pad
 * channels up to ld; the conv loaders read them against zero weights, so they must be finite. */
int prv2_zero_pad_channels(float* y, int64_t rows, int32_t c, int32_t ld, void* stream);
/* AttractorLayerUnnormed, kind='mean', type='inv' (attractor.py:45-57,186-206):
 *   out[r, j] = bins[r, j] + mean_i( dx / (1 + alpha * dx^2) ),  dx = attr[r, i] - bins[r, j]               */
int prv2_zoe_attractor(const float* attr, int32_t ld_attr, int32_t n_attr, const float* bins, int32_t ld_bins,
                       int32_t n_bins, float alpha, int64_t rows, float* out, int32_t ld_out, void* stream);
/* ConditionalLogBinomial tail + expectation (dist_layers.py:29-69,100-116; zoedepth_v1.py:219):
 *   pt[r, 0..3] = softplus MLP output; p = (pt0+eps)/(pt0+pt1+2eps); t = (max-min)*(pt2+eps)/(pt2+pt3+2eps)+min;
 *   depth[r] = sum_k softmax_k( (logC(K-1,k) + k log p + (K-1-k) log(1-p)) / t ) * centers[r, k]                */
int prv2_zoe_logbinom_depth(const float* pt, int32_t ld_pt, const float* centers, int32_t ld_c, int32_t n_bins,
                            float min_temp, float max_temp, int64_t rows, float* depth, void* stream);

/* ------------------------------------------------------------------------------------------
 * Gathers
 * ------------------------------------------------------------------------------------------ */
/* Input stage of the image dataset (estimator/datasets/general_dataset.py:22-62, generic branch): RGB image, HWC, uint8
 * (src_is_u8) or fp32, device memory -> v / 255 (uint8 only) -> F.interpolate(mode='bicubic', align_corners=True) to H x W ->
 * CHW fp32 image_hr.  Evaluated in float64 like the reference (its cv2-image / 255.0 is a float64 array), rounded once. */
int prv2_bicubic_resize(const void* src_hwc, int32_t src_is_u8, int32_t h, int32_t w, float* dst_chw, int32_t H, int32_t W,
                        void* stream);

/* Crop + bilinear(align_corners=True) resize of K tiles of a CHW image into NHWC patches, with the
 * (v - mean[c]) / std[c] input normalisation fused.
 *   baseline_pretrain.py:169-170,274-275 -> external/depth_anything/transform.py:127-129 (ResizeDA)
 *   or external/zoedepth/models/base_models/midas.py:171-174 (ResizeZoe); dpt.py:183; lightweight_refiner.py:293.
 * tiles: device int32 [k][2] = (h_start, w_start); crop size ch x cw; output [k, oh, ow, ldo] channels 0..2. */
int prv2_crop_resize(const float* img_chw, int32_t img_h, int32_t img_w, const int32_t* tiles, int32_t k, int32_t ch,
                     int32_t cw, int32_t oh, int32_t ow, const float* mean3_host, const float* std3_host, float* out,
                     int32_t ldo, void* stream);

/* torchvision.ops.roi_align(feat.repeat(K), boxes, (oh, ow), spatial_scale, sampling_ratio=-1, aligned=True)
 * without materialising the K copies (patchrefinerplus.py:263-276, patchrefiner.py:199-210).
 * feat NHWC [1, h, w, c]; boxes device fp32 [k][4] = (x1, y1, x2, y2) in lr-frame pixels. */
int prv2_roi_align(const float* feat, int32_t h, int32_t w, int32_t c, int32_t ldf, const float* boxes, int32_t k,
                   float spatial_scale, int32_t oh, int32_t ow, float* out, int32_t ldo, void* stream);

/* prv2_roi_align writing the pre-split "X2" format of prv2_conv_desc.fmt (c % 8 == 0): the coarse half of a GatedConvUnit's concat
 * buffer, whose only consumer is the gate kernel.  Same arithmetic; out[...] = X2(roi value). */
int prv2_roi_align_x2(const float* feat, int32_t h, int32_t w, int32_t c, int32_t ldf, const float* boxes, int32_t k,
                      float spatial_scale, int32_t oh, int32_t ow, float* out, int32_t ldo, void* stream);

/* The coarse half of a ``cat([fine, coarse_roi])`` 3x3 convolution, once per frame at coarse resolution (csrc/coarse_taps.hip):
 *   GatedConvUnit.forward        bi_directional_fusion_model.py:70-73    fusion_conv.0(cat([out, c_feat]))    (no activation in front)
 *   BiDirectionalFusion.forward  bi_directional_fusion_model.py:424-426  fusion_layers_1[l](cat([c, f]))
 *   with c_feat = roi_align(feat.repeat(K), boxes, (h, w), h / P) of the per-frame pyramid level (patchrefinerplus.py:263-283).
 * The conv is linear and its coarse input is a bilinear zoom of ``feat``, so
 *     conv3x3(c_feat; W_c)(p) = sum_tap [p + d_tap inside the tile] Bil(G_tap; s(p + d_tap)),   G_tap = W_c[:, :, tap] . feat
 * (s = roi_align's sample position, Bil = its clamped bilinear sample).  G is a 1x1 GEMM at COARSE resolution (prv2_conv2d with the
 * weights [tap * c + co][ci]); the reference spends 9 * cin * cout MACs per output pixel of each of the frame's tiles on it.
 *
 * prv2_coarse_tap_knots: tiles of one frame share their size, i.e. consecutive output pixels are knot_b = tile / frame (per axis,
 *   <= 1/2) apart in coarse coordinates; U(y, x) = sum_tap Bil(G_tap; y + dy knot_bh, x + dx knot_bw) is then piecewise bilinear on
 *   the knot grid {k - b, k, k + b} and this call tabulates it there: g [h, w, ldg] (channel tap * c + co) -> v [3h, 3w, ldv].
 * prv2_coarse_tap_gather: out[k, i, j, :] = U(s(i, j)) - the taps the zero padding hides on the tile's border pixels (sampled from
 *   g): the pre-LayerNorm addend of prv2_conv3x3_ln_gate_pre / the pre-activation addend of the level's fusion conv.  A 4-tap
 *   gather, like the prv2_roi_align of c_feat it replaces.  boxes / spatial_scale as prv2_roi_align; every box must lie inside the
 *   frame and yield one sample per bin (roi extent <= output size).  Exact algebra; rounding differs from the reference's order by
 *   ~1e-7 relative. */
int prv2_coarse_tap_knots(const float* g, int32_t h, int32_t w, int32_t c, int32_t ldg, float knot_bh, float knot_bw, float* v,
                          int32_t ldv, void* stream);
int prv2_coarse_tap_gather(const float* v, const float* g, int32_t h, int32_t w, int32_t c, int32_t ldv, int32_t ldg, float knot_bh,
                           float knot_bw, const float* boxes, int32_t k, float spatial_scale, int32_t oh, int32_t ow, float* out,
                           int32_t ldo, void* stream);

/* F.interpolate(mode='bilinear', align_corners=True) on NHWC (every decoder upsample; Appendix C row 1) */
int prv2_upsample_bilinear(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ldx, int32_t oh,
                           int32_t ow, float* y, int32_t ldy, void* stream);

/* The two depth maps every fusion level appends to its features ([feat | pred1 | pred2], fusion_model.py:91-118,
 * bi_directional_fusion_model.py:424-436): p1, p2 dense [n, h, w], resized bilinear(align_corners=True) to oh x ow and written
 * as y[pixel][0..3] = (p1, p2, 0, 0) -- y points at the first of the buffer's last four channels (the two maps and the two
 * pad channels), 16-byte aligned, pixel stride ldy.  Same arithmetic per map as prv2_upsample_bilinear. */
int prv2_depth_pair_fill(const float* p1, const float* p2, int32_t n, int32_t h, int32_t w, int32_t oh, int32_t ow, float* y,
                         int32_t ldy, void* stream);

/* Border correction of a 3x3 / pad 1 conv whose bias was folded from an upstream constant: GatedFusionBlock's ``out_conv`` (1x1, bias b)
 * followed by the bilinear x2 and ``output_conv1`` (3x3, W) is ONE 3x3 conv with weights W o out_conv and bias + sum_taps W_tap b
 * (bi_directional_fusion_model.py:139-142,201: a 1x1 commutes with the interpolation, whose weights sum to one) -- except that at
 * the image border the zero padding hides some taps from b.  y[n, oy, ox, :c] -= sum of tap_bias[tap][:] over the taps of (oy, ox)
 * that fall outside the image; tap_bias: [9][c] (ky, kx order), c % 4 == 0. */
int prv2_conv_border_bias(float* y, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ldy, const float* tap_bias, void* stream);

/* NCHW <-> NHWC layout changes at the boundary (image_lr in, coarse_prediction out) */
int prv2_nchw_to_nhwc(const float* x, int32_t n, int32_t c, int32_t h, int32_t w, float* y, int32_t ldy, void* stream);
int prv2_nhwc_to_nchw(const float* x, int32_t n, int32_t c, int32_t h, int32_t w, int32_t ldx, float* y, void* stream);

/* ------------------------------------------------------------------------------------------
 * Overlap blend = RunningAverageMap kept on the device (estimator/models/utils.py:22-49,
 * paste/update loops baseline_pretrain.py:212-229, 347-373).
 * avg / cnt: dense [H, W] maps.  pred: [k, ph, pw] patch predictions.  mask: [th, tw] blend weights.
 * tiles: device int32 [k][2] = (h_start, w_start) in map coordinates; tile size th x tw.
 * When (ph, pw) != (th, tw) the prediction is upsampled with legacy 'nearest' (baseline_pretrain.py:210).
 * Tiles are applied in order k = 0..K-1 (the running mean is order dependent).
 * ------------------------------------------------------------------------------------------ */
int prv2_blend_paste(float* avg, float* cnt, int32_t map_h, int32_t map_w, const float* pred, int32_t ph, int32_t pw,
                     const float* mask, const int32_t* tiles, int32_t k, int32_t th, int32_t tw, void* stream);
int prv2_blend_update(float* avg, float* cnt, int32_t map_h, int32_t map_w, const float* pred, int32_t ph,
                      int32_t pw, const float* mask, const int32_t* tiles, int32_t k, int32_t th, int32_t tw,
                      void* stream);
/* RunningAverageMap.resize: avg -> nearest, cnt -> bilinear(align_corners=True) (utils.py:38-43) */
int prv2_blend_resize(const float* avg, const float* cnt, int32_t h, int32_t w, float* avg_out, float* cnt_out,
                      int32_t oh, int32_t ow, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PRV2_H */
