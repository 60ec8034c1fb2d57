"""The per-patch fusion networks on the HIP kernels.

Host-side mirrors of
  FusionUnet           estimator/models/blocks/fusion_model.py:53-122
  BiDirectionalFusion  estimator/models/blocks/bi_directional_fusion_model.py:290-446 ('coarse-gated' | 'coarse-fusion' | 'self-agg')
  C2FModule / GatedFusionBlock / GatedConvUnit   ...bi_directional_fusion_model.py:26-208
with the reference constructor arguments and state-dict names.

Layout: NHWC ``Feat``s.  Every ``torch.cat`` of the reference is realised by letting producers
write into channel slices of one pre-allocated buffer; the channels-first LayerNorm+GELU is the
row-LayerNorm kernel run in place on the conv output; residual / gate / bias / activation are
conv-epilogue flags.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import ops
from . import weights as W
from .dav2 import StateDictModule
from .ops import ACT_GELU, ACT_RELU, ACT_SIGMOID, Feat


import os

FOLD_OUT_CONV = os.environ.get("PRV2_FOLD_OUT_CONV", "1") != "0"  # A/B and test switch (BiDirectionalFusion._pack)
# A/B and test switch: which consumers take their coarse half from the per-frame tap tables (prepare_frame)
UPCONV_MIN_C = int(os.environ.get("PRV2_UPCONV_MIN_C", "256"))  # decoder stages with at least this many interpolated input channels split (see _pack_encdec)
TAPS_PARTS = tuple(os.environ.get("PRV2_TAPS_PARTS", "gate256,gate_narrow,enc1").split(","))


def as_feat1(t: torch.Tensor) -> Feat:
    """dense [B,1,H,W] tensor == NHWC [B,H,W,1]."""
    B, _, H, Wd = t.shape
    return Feat(t.contiguous().view(B, H, Wd, 1))


def place(src, dst: Feat):
    """bilinear(align_corners) resize of ``src`` into ``dst`` (an exact copy when sizes match).  A coarse-pyramid ROI that has
    not been materialised (ops.RoiSource) is gathered straight into ``dst`` when the sizes match."""
    if isinstance(src, ops.RoiSource):
        if (src.h, src.w) == (dst.h, dst.w):
            return src.write(dst)
        src = src.materialize()
    ops.upsample_bilinear(src, dst.h, dst.w, out=dst)


def _fused_tail(c0: int) -> bool:
    """can [.. c0 channels | pred1 | pred2 | pad | pad] be closed by ops.depth_pair_fill?"""
    return c0 % 4 == 0 and ops.DIRECT_PLACEMENT


def alloc_with_pred_tail(B, h, w, c0, dev) -> Feat:
    """concat buffer [c0 feature channels | pred1 | pred2]; when the tail is 16-byte aligned ``place_preds`` writes the two
    pad channels too, so nothing has to be zeroed beforehand"""
    return Feat.alloc_raw(B, h, w, c0 + 2, dev) if _fused_tail(c0) else Feat.alloc(B, h, w, c0 + 2, dev)


def place_preds(pred1: Feat, pred2: Feat, buf: Feat, c0: int):
    """the [pred1 | pred2] tail of a buffer from ``alloc_with_pred_tail`` (the two depth maps resized to the level)"""
    assert buf.c0 == 0 and buf.c == c0 + 2 and pred1.ld == 1 and pred2.ld == 1
    if _fused_tail(c0):
        return ops.depth_pair_fill(pred1, pred2, buf, c0)
    place(pred1, buf.slice(c0, 1))
    place(pred2, buf.slice(c0 + 1, 1))


class _EncDec(StateDictModule):
    """Shared fine2coarse encoder + decoder of FusionUnet / BiDirectionalFusion."""

    ENC1 = ENC2 = DEC = ""
    HEAVY = False  # BiDirectionalFusionHeavy: three convs per encoder layer, five per decoder stage
    trace = None  # tests set this to a dict: intermediate maps of the last call (c2f_depth, c2f_last, dec_last, offset)

    def _init_encdec(self, in_chl: Sequence[int], temp_chl: Sequence[int], dec_chl: Sequence[int]):
        self.in_chl, self.temp_chl, self.dec_chl = list(in_chl), list(temp_chl), list(dec_chl)
        t = self.temp_chl[::-1]
        self.dec_in = []
        ch = t[0]
        for tc, dc in zip(t[1:], self.dec_chl):
            self.dec_in.append((ch, tc, dc))  # (x1 channels, x2 channels, out channels)
            ch = dc
        self.last_chl = self.dec_chl[-1] if self.dec_chl else ch

    def _pack_encdec(self):
        P = {}
        for l in range(len(self.temp_chl)):
            for e in (self.ENC1, self.ENC2):
                P[f"{e}.{l}"] = (self._conv(f"{e}.{l}.single_conv.0"), self._dev(f"{e}.{l}.single_conv.1.weight"),
                                 self._dev(f"{e}.{l}.single_conv.1.bias"))
                if self.HEAVY:  # SingleConvCNNLNHeavy (bi_directional_fusion_model.py:449-463): ... -> conv -> LN -> conv -> GELU
                    P[f"{e}.{l}.heavy"] = (self._conv(f"{e}.{l}.single_conv.2"), self._dev(f"{e}.{l}.single_conv.3.weight"),
                                           self._dev(f"{e}.{l}.single_conv.3.bias"), self._conv(f"{e}.{l}.single_conv.4"))
        for j, (c1, c2, _) in enumerate(self.dec_in):
            P[f"{self.DEC}.{j}"] = (self._conv(f"{self.DEC}.{j}.conv.double_conv.0"),
                                    self._conv(f"{self.DEC}.{j}.conv.double_conv.{8 if self.HEAVY else 2}"))
            if self.HEAVY:  # DoubleConvHeavy (:465-485): three more n -> n convs in between
                P[f"{self.DEC}.{j}.mid"] = [self._conv(f"{self.DEC}.{j}.conv.double_conv.{i}") for i in (2, 4, 6)]
            # UpSample.forward_hardcode (fusion_model.py:15-24): double_conv.0 over cat([interpolate(x1), x2, pred1, pred2]) is linear in
            # its input, so it splits by weight columns: the interpolated x1 part runs as tap GEMMs at x1's resolution (ops.upconv3x3),
            # the rest as an ordinary conv whose raw output is that kernel's pre-activation addend.  Worth it from 256 upsampled
            # channels on (layer report: 128 of 194 breaks even, 64 of 98 loses to the gather's fixed cost per output channel).
            w0 = self._sd[f"{self.DEC}.{j}.conv.double_conv.0.weight"]
            if self.prec != ops.PREC_F32 and ops.UPCONV and c1 >= UPCONV_MIN_C and c1 % 32 == 0 and w0.shape[1] == c1 + c2 + 2:
                P[f"{self.DEC}.{j}.split"] = (ops.pack_conv(w0[:, :c1], None, device=self.device, prec=self.prec),
                                              ops.pack_conv(w0[:, c1:], None, device=self.device, prec=self.prec))
        P["final_w"] = self._dev("final_conv.weight")
        return P

    def _pack_enc1_taps(self, P, coarse_chl, extra=None):
        """``ENC1[l]`` over cat([c, f]) (fusion_model.py:91-95, bi_directional_fusion_model.py:424-426; coarse half FIRST): the fine
        half as a conv of its own and the coarse half as 1x1 GEMM weights of the nine tap maps (ops.CoarseTaps / prepare_frame).
        ``extra[l]``: further consumers of the level's coarse map [(name, tap weights)] that share the level's GEMM.
        P["taps"][l] = (packed [sum 9 * cout, c_l] weights, [(consumer name, cout), ...]).  bf16 modes only."""
        P["taps"] = {}
        if self.prec == ops.PREC_F32 or self.HEAVY:  # (heavy: the first conv's LayerNorm has no activation behind it -- general kernels)
            return
        for l in range(len(self.temp_chl)):
            rows = list(extra[l]) if extra is not None else []
            cc = coarse_chl[l]
            w1 = self._sd[f"{self.ENC1}.{l}.single_conv.0.weight"]
            if w1.shape[1] > cc and w1.shape[0] % 4 == 0 and "enc1" in TAPS_PARTS:
                P[f"{self.ENC1}.{l}.fine"] = ops.pack_conv(w1[:, cc:], None, device=self.device, prec=self.prec)
                rows.append(("enc1", ops.coarse_tap_weight(w1[:, :cc])))
            if rows:
                P["taps"][l] = (ops.pack_conv(torch.cat([t for _, t in rows], 0), None, device=self.device, prec=self.prec),
                                [(n, t.shape[0] // 9) for n, t in rows])
            # full-resolution 32-channel level: ENC1[l] (fine half) + ENC2[l] as ONE kernel (ops.chain32_enc, csrc/chain32.hip)
            w2 = self._sd[f"{self.ENC2}.{l}.single_conv.0.weight"]
            if (f"{self.ENC1}.{l}.fine" in P and self.prec == ops.L.PREC_BF16X3 and w1.shape[0] == 32 and w1.shape[1] - cc == 32 and tuple(w2.shape[:2]) == (32, 34)
                    and ops.CHAIN32):
                sd = lambda k: self._sd.get(k)  # noqa: E731
                e1, e2 = f"{self.ENC1}.{l}.single_conv.", f"{self.ENC2}.{l}.single_conv."
                P[f"chain_enc.{l}"] = dict(
                    w1=ops.pack_chain32(w1[:, cc:], 0, self.device), w2=ops.pack_chain32(w2, 1, self.device), wt=ops.pack_chain32(w2, 2, self.device),
                    consts=ops.chain32_consts(self.device, b1=sd(e1 + "0.bias"), ln1w=sd(e1 + "1.weight"), ln1b=sd(e1 + "1.bias"), b2=sd(e2 + "0.bias"),
                                              ln2w=sd(e2 + "1.weight"), ln2b=sd(e2 + "1.bias")))

    def _enc1_taps_of(self, P, c_feat, f_sizes):
        """per level: (ops.CoarseTaps, the level's ops.RoiSource) when ENC1[l] takes its coarse half from the frame's tap table, else None"""
        out = [None] * len(self.temp_chl)
        for l, c in enumerate(c_feat):
            aux = c.feat.aux if isinstance(c, ops.RoiSource) else None
            key = f"{self.ENC1}.{l}.fine"
            if (aux is not None and "enc1" in aux["taps"] and ops.COARSE_TAPS and key in P and (c.h, c.w) == tuple(f_sizes[l])
                    and P[key].cin == self.in_chl[l] - c.c and ops.conv2d_pre_supported(*f_sizes[l], P[key], ln=True)):
                out[l] = (aux["taps"]["enc1"], c)
        return out

    # -- once per frame ------------------------------------------------------------------------
    def prepare_frame(self, c_feat: List[Feat], knot_b):
        """Per-frame part of the fusion network: the coarse half of every conv that reads cat([., c_feat[l]]) -- the GatedConvUnits'
        ``fusion_conv.0`` and ``fusion_layers_1[l]`` -- as the pyramid level through the convs' coarse weights (ONE 1x1 GEMM per level at
        COARSE resolution) + the knot table of each consumer (ops.CoarseTaps); the reference convolves the x``split`` zoom of the same
        map once per tile.  c_feat: the 6 pyramid levels high -> low (as ``forward`` takes their ROIs); knot_b = (tile height / frame
        height, tile width / frame width).  Attached to the maps (``Feat.aux``): ``forward`` finds it through its ``ops.RoiSource``s.
        A no-op in the f32 mode (which keeps the reference's order of operations)."""
        P = self._packed
        if P is None or not ops.COARSE_TAPS or not P.get("taps") or max(knot_b) > 0.5 or min(knot_b) <= 0:
            return
        kb = (float(knot_b[0]), float(knot_b[1]))
        for l, (tw, cons) in P["taps"].items():
            f = c_feat[l]
            if f.c != tw.cin or f.n != 1 or (f.aux is not None and f.aux.get("kb") == kb):
                continue
            g = ops.conv2d(f, tw, algo=0.0)  # [1, H, W, sum 9 * cout]
            taps, o = {}, 0
            for n, co in cons:
                taps[n] = ops.CoarseTaps(g.slice(o, 9 * co), co, kb)
                o += 9 * co
            f.aux = dict(kb=kb, g=g, taps=taps)

    @staticmethod
    def frame_tensors(c_feat: List[Feat]):
        """the device tensors ``prepare_frame`` attached (for stream bookkeeping by the caller)"""
        out = []
        for f in c_feat:
            if f.aux is not None:
                out.append(f.aux["g"].buf)
                out.extend(t.v.buf for t in f.aux["taps"].values())
        return out

    def _encode_decode(self, P, pairs, sizes, pred1: Feat, pred2: Feat, update_base: Optional[torch.Tensor], out=None,
                       cat1_bufs=None, enc1_taps=None):
        """pairs[l] = (fill_fn(dst_cat: Feat) writing the level-l [c, f] concat), sizes[l] = (h, w);
        levels high -> low resolution (fusion_model.py:91-118).  ``cat1_bufs`` = already allocated (and
        possibly partly filled) level concat buffers.  ``enc1_taps[l]`` = (ops.CoarseTaps, the level's ops.RoiSource, fine feature) for
        levels whose first encoder conv takes the coarse half of cat([c, f]) from the per-frame tap table (no concat buffer), or None."""
        B, dev = pred1.n, pred1.device
        L_ = len(self.temp_chl)
        nd = len(self.dec_in)
        # decoder concat buffers [up(x1) | x2 | pred1 | pred2]; level l feeds decoder stage nd-1-l as x2
        dec_bufs = []
        for j, (c1, c2, dc) in enumerate(self.dec_in):
            h, w = sizes[L_ - 2 - j]
            dec_bufs.append(alloc_with_pred_tail(B, h, w, c1 + c2, dev))
        temps = [None] * L_
        dec_tail_done = [False] * nd  # decoder concat buffers whose [pred1 | pred2] tail the conv that fills x2 has written

        def conv_ln_gelu(x, conv, lnw, lnb, dst, tail_c0=None, buf=None, heavy=None):
            """conv -> LN -> GELU (convs.py:67-72) into ``dst``; when dst's row ends in the [pred1 | pred2] tail and the library fuses
            it, the same launch writes the tail (ops.conv2d_tail); returns whether it did"""
            if heavy is not None:  # conv -> LN -> conv -> LN -> conv -> GELU (SingleConvCNNLNHeavy)
                conv2, lnw2, lnb2, conv3 = heavy
                t = ops.conv2d(x, conv, ln=(lnw, lnb))
                t = ops.conv2d(t, conv2, ln=(lnw2, lnb2))
                ops.conv2d(t, conv3, dst, act=ACT_GELU)
                return False
            if tail_c0 is not None and _fused_tail(tail_c0) and ops.conv2d_tail_supported(x, conv, dst):
                ops.conv2d_tail(x, conv, dst, pred1, pred2, act=ACT_GELU, ln=(lnw, lnb))
                return True
            ops.conv2d(x, conv, dst, act=ACT_GELU, ln=(lnw, lnb))
            return False

        for l in range(L_):
            h, w = sizes[l]
            tc = self.temp_chl[l]
            conv, lnw, lnb = P[f"{self.ENC1}.{l}"]
            j = L_ - 2 - l  # decoder stage that consumes this level as the skip x2 (fusion_model.py:104-111)
            chain = P.get(f"chain_enc.{l}") if enc1_taps is not None and enc1_taps[l] is not None and not self.HEAVY else None
            if chain is not None and (pred1.h, pred1.w, pred2.h, pred2.w) == (h, w, h, w) and enc1_taps[l][2].c == 32:
                # ENC1[l] + ENC2[l] in one kernel: the level's 32-channel map never leaves the chip between the two convs
                taps, roi, fine = enc1_taps[l]
                pre = taps.gather(roi.boxes, roi.scale, h, w)
                dst = dec_bufs[j].slice(self.dec_in[j][0], tc) if 0 <= j < nd else Feat.alloc(B, h, w, tc, dev)
                temps[l] = ops.chain32_enc(fine, chain, pre, pred1.buf, pred2.buf, out=dst, pre_cin=roi.c)
                continue
            cat2 = alloc_with_pred_tail(B, h, w, tc, dev)  # (behind the chain branch, which never needs it: 1.1 GB at 41 x 384 x 512 x 36)
            if enc1_taps is not None and enc1_taps[l] is not None:
                taps, roi, fine = enc1_taps[l]
                pre = taps.gather(roi.boxes, roi.scale, h, w)  # conv3x3(c; W[:, :c_l]) of the tiles, from the frame's table
                fused_ln = tc <= 128 or tc == 256
                ops.conv2d_pre(fine, P[f"{self.ENC1}.{l}.fine"], pre, cat2.slice(0, tc) if fused_ln else None, act=ACT_GELU if fused_ln else ops.ACT_NONE,
                               ln=(lnw, lnb) if fused_ln else None, pre_cin=roi.c)
                assert fused_ln  # (wider levels are not taken: BiDirectionalFusion.forward asks conv2d_pre_supported(ln=True))
                place_preds(pred1, pred2, cat2, tc)
            else:
                cat1 = cat1_bufs[l] if cat1_bufs is not None else Feat.alloc(B, h, w, self.in_chl[l], dev)
                pairs[l](cat1)
                if not conv_ln_gelu(cat1, conv, lnw, lnb, cat2.slice(0, tc), tail_c0=tc, heavy=P.get(f"{self.ENC1}.{l}.heavy")):
                    place_preds(pred1, pred2, cat2, tc)
            conv, lnw, lnb = P[f"{self.ENC2}.{l}"]
            if 0 <= j < nd:
                c1, c2, _ = self.dec_in[j]
                assert c2 == tc
                dst = dec_bufs[j].slice(c1, c2)
                dec_tail_done[j] = conv_ln_gelu(cat2, conv, lnw, lnb, dst, tail_c0=c1 + c2, heavy=P.get(f"{self.ENC2}.{l}.heavy"))
            else:
                dst = Feat.alloc(B, h, w, tc, dev)
                conv_ln_gelu(cat2, conv, lnw, lnb, dst, heavy=P.get(f"{self.ENC2}.{l}.heavy"))
            temps[l] = dst
        feat = temps[L_ - 1] if nd > 0 else temps[0]
        for j, (c1, c2, dc) in enumerate(self.dec_in):
            buf = dec_bufs[j]
            if not dec_tail_done[j]:
                place_preds(pred1, pred2, buf, c1 + c2)
            c0w, c2w = P[f"{self.DEC}.{j}"]
            split = P.get(f"{self.DEC}.{j}.split")
            if split is not None and isinstance(feat, Feat) and ops.upconv3x3_supported(feat, buf.h, buf.w, split[0]):
                # conv over [x2 | pred1 | pred2] with the weights' other columns (raw), then the interpolated x1 part on top + GELU
                t = ops.conv2d(buf.slice(c1, buf.c - c1), split[1])
                ops.upconv3x3(feat, buf.h, buf.w, split[0], out=t, act=ACT_GELU, add=t, bias=False)
            elif isinstance(feat, Feat) and ops.conv2d_ups_supported(buf, feat, c0w):
                # bilinear(x1 -> size of x2) is formed inside the conv's tile loader (fusion_model.py:16-18): the upsampled x1 is
                # never written; channels [0, c1) of ``buf`` stay unused
                t = ops.conv2d_ups(buf, feat, c0w, act=ACT_GELU)
            else:
                place(feat, buf.slice(0, c1))
                t = ops.conv2d(buf, c0w, act=ACT_GELU)
            for cm in P.get(f"{self.DEC}.{j}.mid", ()):
                t = ops.conv2d(t, cm, act=ACT_GELU)
            feat = ops.conv2d(t, c2w, act=ACT_GELU)
        if self.trace is not None:  # tests: the last decoder stage, as the oracle sees it
            self.trace["dec_last"] = feat.to_nchw()
            self.trace["offset"] = ops.conv2d_cout1(feat, P["final_w"], None, 3)
        # final_conv 3x3 -> 1 ; clamp(update_base + offset, min=0)
        return ops.conv2d_cout1(feat, P["final_w"], None, 3, res=update_base, clamp0=update_base is not None, out=out)


class FusionUnet(_EncDec):
    ENC1, ENC2, DEC = "encoder_layers_1", "encoder_layers_2", "decoder_layers"

    def __init__(self, input_chl=(64, 512, 512), temp_chl=(32, 256, 256), dec_chl=(256, 32), device="cuda", prec="f32"):
        """(the reference's constructor: input_chl[l] = coarse + fine channels of level l; the split is read off the first call's inputs)"""
        super().__init__()
        self.device = torch.device(device)
        self.prec = ops.L.PREC_NAMES[prec] if isinstance(prec, str) else prec
        self._init_encdec(input_chl, temp_chl, dec_chl)
        self._spec = W.fusion_unet_spec("", input_chl, temp_chl, dec_chl)
        self._packed = None
        self.glb_att = False

    def _pack(self):
        if len(self._sd) == len(self._spec):
            self._packed = self._pack_encdec()
            self._packed["taps"] = {}

    def prepare_frame(self, c_feat: List[Feat], knot_b):
        """_EncDec.prepare_frame; the coarse / fine split of ``input_chl`` is only known from the maps themselves (the reference's
        constructor takes the sums): the coarse-half weights are packed at the first frame"""
        P = self._packed
        if P is not None and not P["taps"] and self.prec != ops.PREC_F32 and ops.COARSE_TAPS and len(c_feat) == len(self.temp_chl):
            self._pack_enc1_taps(P, [f.c for f in c_feat])
        super().prepare_frame(c_feat, knot_b)

    def forward(self, c_feat: List[Feat], f_feat: List[Feat], pred1: torch.Tensor, pred2: torch.Tensor,
                update_base: Optional[torch.Tensor] = None, out=None) -> torch.Tensor:
        """c_feat / f_feat: high -> low resolution; pred1/pred2/update_base dense [B,1,H,W]."""
        P = self._packed
        if P is None:
            raise RuntimeError("FusionUnet: weights not loaded")

        def fill(l):
            def fn(cat: Feat):
                place(c_feat[l], cat.slice(0, c_feat[l].c))
                place(f_feat[l], cat.slice(c_feat[l].c, f_feat[l].c))
            return fn

        sizes = [(f.h, f.w) for f in f_feat]
        # levels whose first encoder conv takes the coarse half of cat([c, f]) from the frame's tap table convolve f alone
        taps = [None if t is None or not isinstance(f_feat[l], Feat) or (f_feat[l].h, f_feat[l].w) != sizes[l] else (t[0], t[1], f_feat[l])
                for l, t in enumerate(self._enc1_taps_of(P, c_feat, sizes))]
        return self._encode_decode(P, [fill(l) for l in range(len(c_feat))], sizes, as_feat1(pred1),
                                   as_feat1(pred2), update_base, out=out, enc1_taps=taps)

    __call__ = forward


class BiDirectionalFusion(_EncDec):
    ENC1, ENC2, DEC = "fusion_layers_1", "fusion_layers_2", "f2r_agg"
    FEATURES = 256  # hard-wired in C2FModule (bi_directional_fusion_model.py:149)

    def __init__(self, encoder_name="", coarse2fine=True, coarse2fine_type="coarse-gated", fine2coarse=True,
                 coarse_chl=(32, 256, 256, 256, 256, 256), fine_chl=(32, 32, 64, 96, 960),
                 fine_chl_after_coarse2fine=(32, 256, 256, 256, 256, 256), temp_chl=(32, 64, 64, 128, 256, 512),
                 dec_chl=(512, 256, 128, 64, 32), glb_att=False, device="cuda", prec="f32", **_unused):
        super().__init__()
        if (coarse2fine and coarse2fine_type not in W.C2F_TYPES) or glb_att:
            raise NotImplementedError(f"coarse2fine_type {coarse2fine_type!r} / glb_att={glb_att}: built are {sorted(W.C2F_TYPES)} with "
                                      "glb_att=False ('coarse-gated' is every released V2 config, the others are the ablations of "
                                      "bi_directional_fusion_model.py:355-372); glb_att (TwoWayTransformer) is enabled by no config")
        # coarse2fine=False (the "base" ablations): no c2f module -- the six refiner maps enter fusion_layers_1 as they are (:407-426)
        self.coarse2fine = bool(coarse2fine)
        if not self.coarse2fine:
            coarse2fine_type = "coarse-gated"  # (unused by the reference too)
        # (fusion, gate) of every GatedConvUnit (:149-176): 'coarse-fusion' hands on the fusion_conv output itself (:79-80),
        # 'self-agg' has no fusion_conv and never looks at the coarse pyramid inside the c2f module
        self.coarse2fine_type = coarse2fine_type
        self.c2f_fusion, self.c2f_gate = W.C2F_TYPES[coarse2fine_type]
        self.device = torch.device(device)
        self.prec = ops.L.PREC_NAMES[prec] if isinstance(prec, str) else prec
        self.f16f6 = prec == "f16f6" or (ops.F16F6 and self.prec == ops.PREC_BF16X3)  # GatedConvUnit.conv in the fp16 + fp6 arithmetic
        self.encoder_name = encoder_name
        self.glb_att = False
        self.coarse_chl, self.fine_chl = list(coarse_chl), list(fine_chl)
        self._init_encdec([c + f for c, f in zip(coarse_chl, fine_chl_after_coarse2fine)], temp_chl, dec_chl)
        self._spec = W.bidir_fusion_spec("", coarse_chl, fine_chl, fine_chl_after_coarse2fine, temp_chl, dec_chl,
                                         coarse2fine_type=coarse2fine_type, coarse2fine=self.coarse2fine, heavy=self.HEAVY)
        self._packed = None

    def _pack(self):
        if len(self._sd) < len(self._spec):
            return
        P = self._pack_encdec()
        if not self.coarse2fine:
            self._pack_enc1_taps(P, self.coarse_chl)
            self._packed = P
            return
        s = "c2f.scratch."
        P["rn"] = [self._conv(f"{s}layer{i + 1}_rn") for i in range(5)]
        if self.coarse2fine_type == "only-gate":  # C2FNOENCModule (:211-251)
            fu = lambda b: dict(conv=self._conv(b + "conv"), f0=self._conv(b + "fusion_conv.0"), lnw=self._dev(b + "fusion_conv.1.weight"),  # noqa: E731
                                lnb=self._dev(b + "fusion_conv.1.bias"), f3=self._conv(b + "fusion_conv.3"))
            P["noenc"] = [(fu(f"{s}layer{k}_gate1."), fu(f"{s}layer{k}_gate2.")) for k in range(1, 7)]
            P["up0"] = ops.pack_conv(self._sd[s + "upsample_conv.0.weight"], self._sd[s + "upsample_conv.0.bias"], convt_k=2,
                                     device=self.device, prec=self.prec)
            P["up2"] = self._conv(s + "upsample_conv.2")
            P["outc_w"], P["outc_b"] = self._dev(s + "output_conv.weight"), self._dev(s + "output_conv.bias")
            self._pack_enc1_taps(P, self.coarse_chl)
            self._packed = P
            return

        def f6(u, b):
            # the fp16 + fp6 arithmetic for GatedConvUnit.conv (ops.F16F6; csrc/conv3x3_f6.hip): a second packed image of the same weights
            wc = self._sd[b + "conv.weight"]
            if self.f16f6 and tuple(wc.shape[2:]) == (3, 3) and ops.L.load().prv2_conv3x3_f6_weight_bytes(wc.shape[0], wc.shape[1]) > 0:
                u["conv_f6"] = ops.pack_conv3x3_f6(wc, self._sd.get(b + "conv.bias"), device=self.device)
            return u

        def unit(b):
            if not self.c2f_fusion:
                return f6(dict(conv=self._conv(b + "conv")), b)
            u = dict(conv=self._conv(b + "conv"), f0=self._conv(b + "fusion_conv.0"),
                     lnw=self._dev(b + "fusion_conv.1.weight"), lnb=self._dev(b + "fusion_conv.1.bias"),
                     f3=self._conv(b + "fusion_conv.3"))
            f6(u, b)
            w0f = self._sd[b + "fusion_conv.0.weight"]
            if self.f16f6 and ops.F6_GATE and self.c2f_gate and ops.L.load().prv2_conv3x3_f6_weight_bytes(w0f.shape[0], w0f.shape[1]) > 0:
                # the unit's tail over the whole [out | coarse ROI] concat (the configs whose ROI gather resizes: no tap tables)
                u["f0_f6"] = ops.pack_conv3x3_f6(w0f, self._sd.get(b + "fusion_conv.0.bias"), device=self.device)
            w3 = self._sd[b + "fusion_conv.3.weight"]
            if self.c2f_gate and self.prec != ops.PREC_F32 and w3.shape[0] == w3.shape[1] and w3.shape[0] in ops.GATE_CHANNELS:  # fused tail kernel (ops.conv3x3_ln_gate)
                u["f3g"] = ops.pack_gate(w3.to(self.device))
                w0 = self._sd[b + "fusion_conv.0.weight"]
                F_ = w3.shape[0]
                if w0.shape[1] == 2 * F_:
                    # fusion_conv.0 over cat([out, c_feat]) (:70-73) = conv3x3(out; W[:, :F]) + conv3x3(c_feat; W[:, F:]); the second
                    # term is linear in the per-frame pyramid level and is computed once per frame at coarse resolution
                    # (prepare_frame / ops.CoarseTaps): ``f0a`` = the fine half (with the bias), ``tapw`` = the coarse half as the
                    # 1x1 GEMM weights of the nine tap maps
                    u["f0a"] = ops.pack_conv(w0[:, :F_], self._sd.get(b + "fusion_conv.0.bias"), device=self.device, prec=self.prec)
                    if self.f16f6 and ops.F6_GATE and ops.L.load().prv2_conv3x3_f6_weight_bytes(F_, F_) > 0:  # stage 2: the unit's tail kernel too
                        u["f0a_f6"] = ops.pack_conv3x3_f6(w0[:, :F_], self._sd.get(b + "fusion_conv.0.bias"), device=self.device)
                    u["tapw"] = ops.coarse_tap_weight(w0[:, F_:])
            return u

        def block(b):
            return dict(out_conv=self._conv(b + "out_conv"), u1=unit(b + "GateresConfUnit1."),
                        u2=unit(b + "GateresConfUnit2."))

        P["refine"] = {r: block(f"{s}refinenet{r}.") for r in range(1, 6)}
        P["out1"] = self._conv(s + "output_conv1")
        if self.prec != ops.PREC_F32 and FOLD_OUT_CONV:
            # refinenet1.out_conv (1x1) -> bilinear x2 -> output_conv1 (3x3) is ONE 3x3 conv on the upsampled gate output: a 1x1
            # commutes with the interpolation (its weights sum to one), so W' = W1 o W_oc and bias' = b1 + sum_taps W1_tap b_oc -- except
            # at the image border, where the zero padding hides taps from b_oc (ops.conv_border_bias).  Saves the 256 -> 256 GEMM at
            # 192 x 256 (2.5 ms per frame); the f32 mode keeps the reference's layer order.
            w1, b1 = self._sd[s + "output_conv1.weight"].double(), self._sd[s + "output_conv1.bias"].double()
            woc, boc = self._sd[s + "refinenet1.out_conv.weight"].double()[:, :, 0, 0], self._sd[s + "refinenet1.out_conv.bias"].double()
            wf64 = torch.einsum("omyx,mi->oiyx", w1, woc)
            wf = wf64.float()
            bt = torch.einsum("omyx,m->yxo", w1, boc)  # [3, 3, out]: what each tap contributes from the folded constant
            P["out1_folded"] = ops.pack_conv(wf, (b1 + bt.sum((0, 1))).float(), pad=1, device=self.device, prec=self.prec)
            P["out1_tap_bias"] = bt.reshape(9, -1).float().contiguous().to(self.device)
            # output_conv2[0] follows output_conv1 with nothing non-linear in between (:201-203): the pair -- behind the bilinear x2 -- is ONE
            # 5x5 conv 256 -> 32 evaluated at path_1's resolution (ops.upconv5x5, csrc/upconv5.hip): the 128-channel full-resolution map
            # is never formed.  Composite weights, the 25 bias classes and the border ring's edge weights in float64 (ops.compose_upconv5x5).
            w2 = self._sd[s + "output_conv2.0.weight"]
            if ops.UPCONV5 and ops.UPCONV and w2.shape[0] <= 32 and w2.shape[0] % 4 == 0:
                P["out5"] = ops.compose_upconv5x5(wf64, b1, bt.reshape(9, -1), w2, self._sd.get(s + "output_conv2.0.bias"), self.device, self.prec)
                P["out5"]["algo_per_px"] = 2.0 * 9 * (w1.shape[1] * w1.shape[0] + w2.shape[1] * w2.shape[0])
        P["out2_0"] = self._conv(s + "output_conv2.0")
        P["out2_fusion"] = block(s + "output_conv2_fusion.")
        P["out3_w"] = self._dev(s + "output_conv3.0.weight")
        P["out3_b"] = self._dev(s + "output_conv3.0.bias")
        # the full-resolution tail output_conv2_fusion (one GatedConvUnit + out_conv) + output_conv3 as ONE kernel (ops.chain32_c2f)
        b = s + "output_conv2_fusion."
        w0 = self._sd.get(b + "GateresConfUnit2.fusion_conv.0.weight")
        if (self.prec == ops.L.PREC_BF16X3 and ops.CHAIN32 and self.c2f_fusion and self.c2f_gate and self.coarse_chl[0] == 32 and w0 is not None and
                tuple(w0.shape[:2]) == (32, 64)):
            u = b + "GateresConfUnit2."
            sd = lambda k: self._sd.get(k)  # noqa: E731
            P["chain_c2f"] = dict(
                w1=ops.pack_chain32(sd(u + "conv.weight"), 0, self.device), w2=ops.pack_chain32(w0[:, :32], 1, self.device),
                wg=ops.pack_chain32(sd(u + "fusion_conv.3.weight")[:, :, 0, 0], 1, self.device), wo=ops.pack_chain32(sd(b + "out_conv.weight")[:, :, 0, 0], 1, self.device),
                consts=ops.chain32_consts(self.device, b1=sd(u + "conv.bias"), ln1w=sd(u + "fusion_conv.1.weight"), ln1b=sd(u + "fusion_conv.1.bias"),
                                          b2=sd(u + "fusion_conv.0.bias"), bg=sd(u + "fusion_conv.3.bias"), bo=sd(b + "out_conv.bias"),
                                          w3=sd(s + "output_conv3.0.weight").reshape(32)),
                b3=float(sd(s + "output_conv3.0.bias").reshape(-1)[0]))
        # Per pyramid level: every conv that reads cat([., c_feat[l]]) with no activation in front -- the level's GatedConvUnits
        # (refinenet{l}: unit 2, and unit 1 where the block has two inputs, :125-129; level 0: output_conv2_fusion's unit 2) and
        # fusion_layers_1[l] -- hands the coarse half of its weights to ONE 1x1 GEMM per level and frame (prepare_frame).
        extra = []
        for l in range(6):
            blk = P["refine"][l] if l >= 1 else P["out2_fusion"]
            cons = [("u2", blk["u2"])] + ([("u1", blk["u1"])] if 1 <= l < 5 else [])
            extra.append([(n, u["tapw"]) for n, u in cons if "tapw" in u and u["tapw"].shape[1] == self.coarse_chl[l] and
                          ("gate256" if u["tapw"].shape[0] == 9 * 256 else "gate_narrow") in TAPS_PARTS])
        self._pack_enc1_taps(P, self.coarse_chl, extra)
        self._packed = P

    # -- coarse2fine ---------------------------------------------------------------------------
    @staticmethod
    def _unit_conv(u, x: Feat, out: Optional[Feat], res: Feat) -> Feat:
        """GatedConvUnit.conv + the skip (:58-64): conv3x3(relu(x)) + res, in the fp16 + fp6 arithmetic where the unit has that image"""
        cw6 = u.get("conv_f6")
        if cw6 is not None and ops.conv3x3_f6_supported(x, cw6.cout, cw6.cin):
            return ops.conv3x3_f6(x, cw6, out, relu_in=True, res=res)
        return ops.conv2d(x, u["conv"], out, relu_in=True, res=res)

    @staticmethod
    def _gated_unit_taps(u, x: Feat, taps: "ops.CoarseTaps", coarse: "ops.RoiSource", F_: int, res: Optional[Feat] = None) -> Feat:
        """GatedConvUnit.forward with the coarse half of ``fusion_conv.0`` taken from the per-frame tap table: the 3x3 conv runs over
        ``out`` only (K = F instead of 2F), ``c_feat`` is never gathered."""
        out = Feat(torch.empty((x.n, x.h, x.w, F_), device=x.device, dtype=torch.float32), x2=F_ == 256)  # (256: the gate kernel's operand format)
        BiDirectionalFusion._unit_conv(u, x, out, x)                                                # conv(relu(x)) + x
        pre = taps.gather(coarse.boxes, coarse.scale, x.h, x.w)                                      # conv3x3(c_feat; W[:, F:]) per tile
        cw6 = u.get("f0a_f6")
        if cw6 is not None and out.x2 and ops.conv3x3_f6_supported(x, cw6.cout, cw6.cin):
            return ops.conv3x3_ln_gate_f6(out, cw6, (u["lnw"], u["lnb"]), u["f3g"], u["f3"].bias, act=ACT_RELU, mul=out, res=res, pre=pre, pre_cin=F_)
        return ops.conv3x3_ln_gate(out, u["f0a"], (u["lnw"], u["lnb"]), u["f3g"], u["f3"].bias, act=ACT_RELU, mul=out, res=res, pre=pre,
                                   pre_cin=F_)

    @staticmethod
    def _plain_unit(u, x: Feat, res: Optional[Feat] = None) -> Feat:
        """GatedConvUnit(fusion=False).forward ('self-agg', :65-70): conv(relu(x)) + x, plus the block's ``xs[0]`` (:127) when given."""
        return BiDirectionalFusion._unit_conv(u, x, None, x if res is None else ops.add(x, res))

    @staticmethod
    def _gated_unit(u, x: Feat, cat: Feat, F_: int, res: Optional[Feat] = None, gate: bool = True, dst: Optional[Feat] = None) -> Feat:
        """GatedConvUnit.forward (bi_directional_fusion_model.py:56-82).  ``cat`` = [B,h,w,2F] whose upper
        half already holds the coarse feature; the lower half receives ``out``.  ``gate=False`` ('coarse-fusion', :79-80): the
        fusion_conv output is the unit's output."""
        out = BiDirectionalFusion._unit_conv(u, x, cat.slice(0, F_), x)                  # conv(relu(x)) + x
        if not gate:
            fused = ops.conv2d(cat, u["f0"], act=ACT_RELU, ln=(u["lnw"], u["lnb"]))   # conv -> LN -> ReLU (:47-50)
            return ops.conv2d(fused, u["f3"], dst, res=res)                              # the 1x1 (:51) (+ xs[0], :127)
        if "f3g" in u and ops.conv3x3_ln_gate_supported(cat, u["f0"]):                  # the whole fusion_conv + gate in one kernel
            cw6 = u.get("f0_f6")
            if cw6 is not None and ops.conv3x3_f6_supported(cat, cw6.cout, cw6.cin, allow_x2=True):
                return ops.conv3x3_ln_gate_f6(cat, cw6, (u["lnw"], u["lnb"]), u["f3g"], u["f3"].bias, act=ACT_RELU, mul=out, res=res)
            return ops.conv3x3_ln_gate(cat, u["f0"], (u["lnw"], u["lnb"]), u["f3g"], u["f3"].bias, act=ACT_RELU, mul=out, res=res)
        fused = ops.conv2d(cat, u["f0"], act=ACT_RELU, ln=(u["lnw"], u["lnb"]))       # conv -> LN -> ReLU (:47-50)
        return ops.conv2d(fused, u["f3"], act=ACT_SIGMOID, mul=out, res=res)             # out * sigmoid(.) (+ xs[0])

    def _gated_block(self, blk, xs: List[Feat], coarse: Feat, F_: int, size=None, upscale=True, dest=None, skip_out_conv=False,
                     defer_upsample=None) -> Feat:
        """GatedFusionBlock.forward (bi_directional_fusion_model.py:116-146).  ``coarse`` may have a
        different size (it is resized while being placed).  The 1x1 ``out_conv`` is applied BEFORE the
        bilinear upsample: both are linear and the bilinear weights sum to one, so
        out_conv(up(x)) == up(out_conv(x)) up to rounding -- 4x fewer FLOPs and no upsampled temporary."""
        ref = xs[-1]
        units = [blk["u2"]] + ([blk["u1"]] if len(xs) == 2 else [])
        aux = coarse.feat.aux if isinstance(coarse, ops.RoiSource) else None
        if (aux is not None and ops.COARSE_TAPS and self.prec != ops.PREC_F32 and (coarse.h, coarse.w) == (ref.h, ref.w)
                and all(n in aux["taps"] for n in (["u2"] + (["u1"] if len(xs) == 2 else [])))
                and all("f0a" in u and ops.conv3x3_ln_gate_supported(ref, u["f0a"]) for u in units)
                and (F_ != 256 or (ops.X2_FORMAT and all(ops._c256(ref, u["conv"]) and u["conv"].cout == 256 for u in units)))):
            # the coarse half of both units comes from the per-frame tap tables (prepare_frame): no concat buffer, no ROI gather
            out = xs[0]
            if len(xs) == 2:
                out = self._gated_unit_taps(blk["u1"], xs[1], aux["taps"]["u1"], coarse, F_, res=xs[0])
            out = self._gated_unit_taps(blk["u2"], out, aux["taps"]["u2"], coarse, F_)
            if not upscale:
                return ops.conv2d(out, blk["out_conv"], dest)
            y = out if skip_out_conv else ops.conv2d(out, blk["out_conv"])
            if defer_upsample is not None and (ops.upconv3x3_supported(y, size[0], size[1], defer_upsample) or
                                               ops.conv2d_ups_supported(ops.UpsOnly(y, size[0], size[1]), y, defer_upsample)):
                return y
            return ops.upsample_bilinear(y, size[0], size[1], out=dest)
        if not self.c2f_fusion:  # 'self-agg': the coarse pyramid is not read here
            out = xs[0]
            if len(xs) == 2:
                out = self._plain_unit(blk["u1"], xs[1], res=xs[0])
            out = self._plain_unit(blk["u2"], out)
            return self._block_tail(blk, out, size, upscale, dest, skip_out_conv, defer_upsample)
        cat = Feat.alloc(ref.n, ref.h, ref.w, 2 * F_, ref.device)
        # The concat [out | coarse ROI] is read by the fused gate kernel only (as its conv input and, first half, as ``mul``): when
        # every writer can produce it -- the ROI gather and the 256-column conv of GatedConvUnit.conv -- it is kept in the kernel's
        # own pre-split operand format (ops.Feat.x2; include/prv2.h PRV2_FMT_*): the halo loader then only copies.  Same results.
        cat.x2 = bool(ops.X2_FORMAT and self.prec != ops.PREC_F32 and F_ % 8 == 0 and
                      isinstance(coarse, ops.RoiSource) and ops.DIRECT_PLACEMENT and (coarse.h, coarse.w) == (ref.h, ref.w) and
                      all("f3g" in u and ops.conv3x3_ln_gate_supported(cat, u["f0"]) and ops._c256(ref, u["conv"]) and u["conv"].cout == 256
                          for u in units))
        place(coarse, cat.slice(F_, F_))
        out = xs[0]
        if len(xs) == 2:
            out = self._gated_unit(blk["u1"], xs[1], cat, F_, res=xs[0], gate=self.c2f_gate)
        out = self._gated_unit(blk["u2"], out, cat, F_, gate=self.c2f_gate)
        return self._block_tail(blk, out, size, upscale, dest, skip_out_conv, defer_upsample)

    @staticmethod
    def _block_tail(blk, out: Feat, size, upscale, dest, skip_out_conv, defer_upsample) -> Feat:
        """The end of GatedFusionBlock.forward (:131-144): interpolate + out_conv, as out_conv + interpolate."""
        if upscale:
            y = out if skip_out_conv else ops.conv2d(out, blk["out_conv"])  # (skip: folded into the consumer's weights, _pack)
            if defer_upsample is not None and (ops.upconv3x3_supported(y, size[0], size[1], defer_upsample) or
                                               ops.conv2d_ups_supported(ops.UpsOnly(y, size[0], size[1]), y, defer_upsample)):
                return y  # the consumer (a 3x3 conv) works from the low-resolution tensor: ops.upconv3x3 / ops.conv2d_ups
            return ops.upsample_bilinear(y, size[0], size[1], out=dest)
        return ops.conv2d(out, blk["out_conv"], dest)

    def _c2f_noenc(self, P, fine: List[Feat], coarse: List[Feat], dests):
        """C2FNOENCModule.forward (bi_directional_fusion_model.py:253-286; 'only-gate'): per level two fusion units on the projected refiner
        map (no top-down path); level 0 = ConvTranspose2d(k2, s2) + ReLU + conv3x3 of the highest refiner map.  dests[l] receives path_l."""
        rn = [ops.conv2d(fine[i], P["rn"][i]) for i in range(5)]
        rn0 = ops.conv2d(ops.conv2d(fine[0], P["up0"], act=ACT_RELU), P["up2"])
        for lvl, x in enumerate([rn0] + rn):               # layer{6 - lvl}_gate* works on pyramid level lvl (:265-281)
            u1, u2 = P["noenc"][5 - lvl]
            F_ = x.c
            cat = Feat.alloc(x.n, x.h, x.w, 2 * F_, x.device)
            place(coarse[lvl], cat.slice(F_, F_))
            x = self._gated_unit(u1, x, cat, F_, gate=False)
            self._gated_unit(u2, x, cat, F_, gate=False, dst=dests[lvl])
        return ops.conv2d_cout1(dests[0], P["outc_w"], P["outc_b"], 3)

    def _c2f(self, P, fine: List[Feat], coarse: List[Feat], dests):
        """C2FModule.forward (bi_directional_fusion_model.py:184-208); fine: 5 maps, coarse: 6 maps, high -> low.
        dests[l] = where feature l of the returned list [last, path2, path3, path4, path5, rn5] is written."""
        F_ = self.FEATURES
        rn = [ops.conv2d(fine[i], P["rn"][i], dests[5] if i == 4 else None) for i in range(5)]
        R = P["refine"]
        path5 = self._gated_block(R[5], [rn[4]], coarse[5], F_, size=(rn[3].h, rn[3].w), dest=dests[4])
        path4 = self._gated_block(R[4], [path5, rn[3]], coarse[4], F_, size=(rn[2].h, rn[2].w), dest=dests[3])
        path3 = self._gated_block(R[3], [path4, rn[2]], coarse[3], F_, size=(rn[1].h, rn[1].w), dest=dests[2])
        path2 = self._gated_block(R[2], [path3, rn[1]], coarse[2], F_, size=(rn[0].h, rn[0].w), dest=dests[1])
        folded = "out1_folded" in P
        w1 = P["out1_folded"] if folded else P["out1"]
        size1 = (rn[0].h * 2, rn[0].w * 2)
        path1 = self._gated_block(R[1], [path2, rn[0]], coarse[1], F_, size=size1, skip_out_conv=folded, defer_upsample=w1)
        if (path1.h, path1.w) != size1 and folded and "out5" in P and ops.upconv5x5_supported(path1, size1[0], size1[1], P["out5"]):
            # output_conv2[0](output_conv1(interpolate(path_1))) (:139-142, :201-203) as one 5x5 conv at path_1's resolution
            last = ops.upconv5x5(path1, size1[0], size1[1], P["out5"], act=ACT_RELU)
            return self._c2f_tail(P, last, coarse, dests)
        if (path1.h, path1.w) != size1 and ops.upconv3x3_supported(path1, size1[0], size1[1], w1):
            # not upsampled yet, and never: output_conv1(interpolate(path_1)) (:139-142, :201) as nine tap GEMMs at path_1's resolution
            # and a four-corner gather per tap (csrc/upconv.hip): 2.3x fewer matrix operations than the conv over the upsampled map
            out = ops.upconv3x3(path1, size1[0], size1[1], w1)
        elif (path1.h, path1.w) != size1:  # output_conv1 samples path_1 bilinearly inside its loader
            out = ops.conv2d_ups(ops.UpsOnly(path1, *size1), path1, w1)
        else:
            out = ops.conv2d(path1, w1)
        if folded:
            ops.conv_border_bias(out, P["out1_tap_bias"])
        last = ops.conv2d(out, P["out2_0"], act=ACT_RELU)
        return self._c2f_tail(P, last, coarse, dests)

    def _c2f_tail(self, P, last: Feat, coarse: List[Feat], dests):
        """C2FModule.forward behind output_conv2 (:203-204): output_conv2_fusion + output_conv3; dests[0] receives the last feature"""
        aux = coarse[0].feat.aux if isinstance(coarse[0], ops.RoiSource) else None
        if ("chain_c2f" in P and aux is not None and ops.COARSE_TAPS and "u2" in aux["taps"] and (coarse[0].h, coarse[0].w) == (last.h, last.w) and
                dests[0].c == 32 and aux["taps"]["u2"].cout == 32):
            # GateresConfUnit2 + out_conv + output_conv3 in one kernel; the unit's coarse half from the frame's tap table as before
            pre = aux["taps"]["u2"].gather(coarse[0].boxes, coarse[0].scale, last.h, last.w)
            _, depth = ops.chain32_c2f(last, P["chain_c2f"], pre, out=dests[0], pre_cin=coarse[0].c)
            return depth
        last = self._gated_block(P["out2_fusion"], [last], coarse[0], self.coarse_chl[0], upscale=False, dest=dests[0])
        depth = ops.conv2d_cout1(last, P["out3_w"], P["out3_b"], 1)
        return depth

    def forward(self, c_feat: List[Feat], f_feat: List[Optional[Feat]], pred1: torch.Tensor, pred2: torch.Tensor,
                update_base: Optional[torch.Tensor] = None, f_sizes=None, out=None, **_unused) -> torch.Tensor:
        """c_feat: 6 coarse ROI maps high -> low.  f_feat: 6 refiner maps high -> low, where entry 0 (the
        2x-upsampled copy the reference builds and then drops, :408) may be None; ``f_sizes`` gives the
        six (h, w) the reference would see."""
        P = self._packed
        if P is None:
            raise RuntimeError("BiDirectionalFusion: weights not loaded")
        if f_sizes is None:  # (a None entry = the x2 copy of the next level: lightweight_refiner.py:314-316)
            f_sizes = [(f.h, f.w) if f is not None else (f_feat[l + 1].h * 2, f_feat[l + 1].w * 2) for l, f in enumerate(f_feat)]
        c_feat = list(c_feat)
        # The reference resizes ALL coarse maps to the refiner's sizes iff the lowest level differs (:389-393).
        # Here the resize happens while the map is placed into its two consumers' concat buffers (no temporary).
        if (c_feat[-1].h, c_feat[-1].w) == tuple(f_sizes[-1]):
            for c, sz in zip(c_feat, f_sizes):
                if (c.h, c.w) != tuple(sz):
                    raise RuntimeError("coarse / refiner pyramids differ at a middle level only: the reference "
                                       "cannot run this shape either (torch.cat fails)")
        B, dev = c_feat[0].n, c_feat[0].device
        # fusion_layers_1[l] over cat([c, f]) (:424-426).  Levels whose coarse half comes from the per-frame tap table (prepare_frame)
        # convolve the c2f feature alone; the others get a concat buffer [coarse | c2f feature] the c2f outputs are written straight into.
        enc1_taps = self._enc1_taps_of(P, c_feat, f_sizes)
        cat1 = [None if enc1_taps[l] is not None else Feat.alloc(B, f_sizes[l][0], f_sizes[l][1], self.in_chl[l], dev) for l in range(6)]
        dests = [Feat.alloc(B, f_sizes[l][0], f_sizes[l][1], self.in_chl[l] - c_feat[l].c, dev) if cat1[l] is None else
                 cat1[l].slice(c_feat[l].c, self.in_chl[l] - c_feat[l].c) for l in range(6)]
        if self.coarse2fine:
            out_depth = (self._c2f_noenc if self.coarse2fine_type == "only-gate" else self._c2f)(P, list(f_feat[1:]), c_feat, dests)
            if self.trace is not None:
                self.trace["c2f_depth"], self.trace["c2f_last"] = out_depth.clone(), dests[0].to_nchw()
        else:
            # the refiner's own maps; level 0 = the x2 bilinear copy of level 1 (lightweight_refiner.py:314-316) when handed over as None
            for l, f in enumerate(f_feat):
                if f is None:
                    ops.upsample_bilinear(f_feat[l + 1], f_sizes[l][0], f_sizes[l][1], out=dests[l])
                else:
                    place(f, dests[l])
            out_depth = pred2  # the caller's (zeros: lightweight_refiner.py:320)

        def fill(l):
            def fn(cat: Feat):
                place(c_feat[l], cat.slice(0, c_feat[l].c))
            return fn

        return self._encode_decode(P, [fill(l) for l in range(6)], list(f_sizes), as_feat1(pred1), as_feat1(out_depth),
                                   update_base, out=out, cat1_bufs=cat1, enc1_taps=[None if t is None else (t[0], t[1], dests[l]) for l, t in enumerate(enc1_taps)])

    __call__ = forward


class BiDirectionalFusionHeavy(BiDirectionalFusion):
    """bi_directional_fusion_model.py:518-560: BiDirectionalFusion with SingleConvCNNLNHeavy encoder layers (conv -> LN -> conv -> LN ->
    conv -> GELU) and DoubleConvHeavy decoder stages (five conv + GELU); the forward is the same (:598-668).  Three ablation configs
    (patchrefinerv2_zoedepth_ablation/plus_*_u4k_base_coarse_heavy.py), all with coarse2fine=False; general conv kernels, not tuned."""
    HEAVY = True
