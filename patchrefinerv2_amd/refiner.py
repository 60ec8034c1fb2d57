"""LightWeightRefiner (V2 per-patch encoder) on the HIP kernels: MobileNetV4-conv-small and ConvNeXt-L encoders.

Host-side mirror of ``LightWeightRefiner`` (estimator/models/blocks/lightweight_refiner.py:242-322)
with ``with_decoder=False`` and the ``mobilenetv4_conv_small`` encoder after the reference's
4-channel stem surgery (estimator/models/patchrefinerplus.py:159-165).

The encoder arithmetic is timm's (``timm.create_model(..., features_only=True)``,
lightweight_refiner.py:260-262) -- timm is not vendored in the reference: the architecture
is restated from the public MobileNetV4 definition (weights.py::MNV4_SMALL) and is PARITY
UNPINNED.  BatchNorm (eval) is folded into the preceding convolution at load time.

``convnext_large`` (configs/patchrefinerv2_zoedepth/v2_convx_u4k.py:90-101; stem surgery patchrefinerplus.py:194-200;
``upsample_convx`` lightweight_refiner.py:277-283,307-313): block arithmetic pinned against HuggingFace transformers'
ConvNext through the reference's own LightWeightRefiner class (tests/golden/convnext_refiner.npz).
"""
from __future__ import annotations

from typing import List

import torch

from . import ops
from . import weights as W
from .dav2 import StateDictModule
from .ops import ACT_NONE, ACT_RELU, Feat

BN_EPS = 1e-5
SUPPORTED_ENCODERS = {"mobilenetv4_conv_small.e2400_r224_in1k": W.MNV4_SMALL, "mobilenetv4_conv_small": W.MNV4_SMALL,
                      "convnext_large": W.CONVNEXT_LARGE, "tf_efficientnet_b5_ap": W.EFFNET_B5}
LN_EPS = 1e-6


class LightWeightRefiner(StateDictModule):
    def __init__(self, encoder_name, coarse_condition=True, with_decoder=False, cls_pretrain=True, device="cuda",
                 prec="f32", arch=None, **_unused):
        super().__init__()
        if encoder_name not in SUPPORTED_ENCODERS:
            raise NotImplementedError(
                f"refiner encoder '{encoder_name}': built are {sorted(SUPPORTED_ENCODERS)} (the V2 configs' three timm "
                "encoders; SURVEY.md 8f rank 3)")
        if with_decoder:
            raise NotImplementedError("with_decoder=True (SimpleDPTHead on the refiner) is not used by any V2 config")
        if not coarse_condition and ("convnext" in encoder_name or "efficientnet" in encoder_name):
            raise NotImplementedError("coarse_condition=False is built for the MobileNetV4 encoder only (the two configs that use it: "
                                      "patchrefinerv2_zoedepth_ablation/plus_mobile_u4k_base{,_e2e}.py)")
        self.encoder_name = encoder_name
        self.arch = arch or SUPPORTED_ENCODERS[encoder_name]  # ``arch``: reduced dims for the parity tests
        # False: the encoder sees the image only (lightweight_refiner.py:298-299; a 3-channel stem, no surgery at patchrefinerplus.py:144)
        self.coarse_condition = bool(coarse_condition)
        self.device = torch.device(device)
        self.prec = ops.L.PREC_NAMES[prec] if isinstance(prec, str) else prec
        self.mean, self.std = self.arch["mean"], self.arch["std"]
        self._packed = None
        self.convnext = "convnext" in encoder_name
        self.effnet = "efficientnet" in encoder_name
        if self.effnet:
            self._spec = W.effnet_spec("refiner_encoder.", self.arch, in_chans=4)
            return
        if self.convnext:
            d0 = self.arch["dims"][0]
            self._spec = W.convnext_spec("refiner_encoder.", self.arch, in_chans=4)
            self._spec["upsample_convx.0.weight"] = (d0, d0 // 2, 2, 2)
            self._spec["upsample_convx.0.bias"] = (d0 // 2,)
            return
        self.layers, self.taps = W.mnv4_layers(self.arch, in_chans=4 if self.coarse_condition else 3)
        self._spec = W.mnv4_spec("refiner_encoder.", self.arch, in_chans=4 if self.coarse_condition else 3)

    def _pack(self):
        if len(self._sd) < len(self._spec):
            return
        if self.convnext:
            return self._pack_convnext()
        if self.effnet:
            return self._pack_effnet()
        P = []
        for Lr in self.layers:
            b = "refiner_encoder." + Lr["bn"] + "."
            scale = self._sd[b + "weight"] / torch.sqrt(self._sd[b + "running_var"] + BN_EPS)
            bias = self._sd[b + "bias"] - self._sd[b + "running_mean"] * scale
            w = self._sd["refiner_encoder." + Lr["conv"] + ".weight"]
            if Lr["g"] == 1:
                P.append(("conv", ops.pack_conv(w, bias, stride=Lr["s"], pad=Lr["k"] // 2, bn_scale=scale,
                                                device=self.device, prec=self.prec)))
            else:
                k = Lr["k"]
                wt = (w.view(w.shape[0], k * k) * scale[:, None]).t().contiguous()  # [k*k][C], BN folded
                P.append(("dw", wt.to(self.device), bias.to(self.device).contiguous()))
        self._packed = P

    def forward(self, crop: Feat, coarse_depth=None, *a, **k):
        """crop: NHWC [B,h,w,4]: channels 0..2 = (rgb - mean)/std (already applied by the crop kernel),
        channel 3 = the metric coarse depth ROI (un-normalised, lightweight_refiner.py:293-296).
        Returns (features high -> low as the reference's ``refiner_features[::-1]`` reversed back, i.e.
        [None(2x copy placeholder), /2, /4, /8, /16, /32], sizes) -- out_depth is zeros (:320)."""
        if self._packed is None:
            raise RuntimeError("LightWeightRefiner: weights not loaded")
        if self.convnext:
            return self._forward_convnext(crop)
        if self.effnet:
            return self._forward_effnet(crop)
        x = crop if self.coarse_condition else crop.slice(0, 3)  # (the crop buffer always carries the depth ROI as channel 3)
        feats: List[Feat] = []
        skip = None
        for i, (Lr, pk) in enumerate(zip(self.layers, self._packed)):
            if Lr.get("res_begin"):
                skip = x
            if pk[0] == "conv":
                x = ops.conv2d(x, pk[1], act=ACT_RELU if Lr["act"] else ACT_NONE,
                               res=skip if Lr.get("res_end") else None)
            else:
                assert not Lr.get("res_end")
                x = ops.dwconv2d(x, pk[1], pk[2], Lr["k"], Lr["s"], Lr["act"])
            if Lr.get("res_end"):
                skip = None
            if i in self.taps:
                feats.append(x)
        sizes = [(feats[0].h * 2, feats[0].w * 2)] + [(f.h, f.w) for f in feats]
        return [None] + feats, sizes

    # -- ConvNeXt ------------------------------------------------------------------------------------------
    def _pack_convnext(self):
        sd, dev, e = self._sd, self.device, "refiner_encoder."
        pk = lambda w, b, **kw: ops.pack_conv(w, b, device=dev, prec=self.prec, **kw)  # noqa: E731
        v = lambda k: sd[k].to(dev).contiguous()  # noqa: E731
        P = dict(stem=pk(sd[e + "stem_0.weight"], sd[e + "stem_0.bias"], stride=4, pad=0),
                 stem_ln=(v(e + "stem_1.weight"), v(e + "stem_1.bias")), stages=[])
        for i, (c, n) in enumerate(zip(self.arch["dims"], self.arch["depths"])):
            st = f"{e}stages_{i}."
            S = dict(blocks=[])
            if i > 0:
                S["down_ln"] = (v(st + "downsample.0.weight"), v(st + "downsample.0.bias"))
                S["down"] = pk(sd[st + "downsample.1.weight"], sd[st + "downsample.1.bias"], stride=2, pad=0)
            for j in range(n):
                b = f"{st}blocks.{j}."
                S["blocks"].append(dict(
                    dw=sd[b + "conv_dw.weight"].view(c, 49).t().contiguous().to(dev), dw_b=v(b + "conv_dw.bias"),
                    ln=(v(b + "norm.weight"), v(b + "norm.bias")),
                    fc1=pk(sd[b + "mlp.fc1.weight"].view(4 * c, c, 1, 1), sd[b + "mlp.fc1.bias"]),
                    fc2=pk(sd[b + "mlp.fc2.weight"].view(c, 4 * c, 1, 1), sd[b + "mlp.fc2.bias"]), gamma=v(b + "gamma")))
            P["stages"].append(S)
        P["up"] = ops.pack_conv(sd["upsample_convx.0.weight"], sd["upsample_convx.0.bias"], convt_k=2, device=dev, prec=self.prec)
        self._packed = P

    def _forward_convnext(self, crop: Feat):
        """dw 7x7 -> LN -> fc1 + GELU -> fc2, * gamma, + x: the dw conv and the LN are HBM-bound row kernels, the two
        linears run on the 1x1 GEMM kernel with GELU / layer-scale / residual fused into their epilogues."""
        P = self._packed
        x = ops.conv2d(crop, P["stem"])
        ops.layernorm_feat(x, *P["stem_ln"], eps=LN_EPS)
        feats: List[Feat] = []
        for S in P["stages"]:
            if "down" in S:
                t = ops.layernorm_feat(x, *S["down_ln"], eps=LN_EPS, out=Feat.alloc(x.n, x.h, x.w, x.c, x.device))
                x = ops.conv2d(t, S["down"])
            for B in S["blocks"]:
                t = ops.dwconv2d(x, B["dw"], B["dw_b"], 7, 1, False)
                ops.layernorm_feat(t, *B["ln"], eps=LN_EPS)
                u = ops.conv2d(t, B["fc1"], act=ops.ACT_GELU)
                x = ops.conv2d(u, B["fc2"], gamma=B["gamma"], res=x)
            feats.append(x)
        up = ops.conv2d(feats[0], P["up"], act=ACT_RELU)  # upsample_convx: ConvTranspose2d(k=2, s=2) + ReLU
        feats = [up] + feats
        sizes = [(up.h * 2, up.w * 2)] + [(f.h, f.w) for f in feats]
        return [None] + feats, sizes  # the stride-1 bilinear copy of ``up`` is dropped by the fusion model

    # -- EfficientNet -----------------------------------------------------------------------------------------
    def _fold(self, b):
        """eval BatchNorm (eps 1e-3) as a per-channel scale folded into the conv + a bias"""
        sd = self._sd
        scale = sd[b + "weight"] / torch.sqrt(sd[b + "running_var"] + self.arch["bn_eps"])
        return scale, sd[b + "bias"] - sd[b + "running_mean"] * scale

    def _pack_effnet(self):
        sd, dev, e = self._sd, self.device, "refiner_encoder."
        pk = lambda w, bn, **kw: ops.pack_conv(w, bn[1], bn_scale=bn[0], device=dev, prec=self.prec, **kw)  # noqa: E731

        def dw(w, bn):  # tap-major [k*k][C] with the BN scale folded
            c, k = w.shape[0], w.shape[-1]
            return (w.view(c, k * k) * bn[0][:, None]).t().contiguous().to(dev), bn[1].to(dev).contiguous()

        P = dict(stem=pk(sd[e + "conv_stem.weight"], self._fold(e + "bn1."), stride=2, same_pad=True), blocks=[])
        for B in W.effnet_blocks(self.arch):
            b = e + B["name"]
            Q = dict(B)
            if B["kind"] == "ds":
                Q["dw"] = dw(sd[b + "conv_dw.weight"], self._fold(b + "bn1."))
                Q["proj"] = pk(sd[b + "conv_pw.weight"], self._fold(b + "bn2."))
            else:
                Q["expand"] = pk(sd[b + "conv_pw.weight"], self._fold(b + "bn1."))
                Q["dw"] = dw(sd[b + "conv_dw.weight"], self._fold(b + "bn2."))
                Q["proj"] = pk(sd[b + "conv_pwl.weight"], self._fold(b + "bn3."))
            v = lambda t: t.to(dev).contiguous()  # noqa: E731
            Q["se"] = (v(sd[b + "se.conv_reduce.weight"].view(B["cse"], B["cmid"])), v(sd[b + "se.conv_reduce.bias"]),
                       v(sd[b + "se.conv_expand.weight"].view(B["cmid"], B["cse"]).t()), v(sd[b + "se.conv_expand.bias"]))
            P["blocks"].append(Q)
        self._packed = P

    def _forward_effnet(self, crop: Feat):
        """MBConv: [1x1 expand + SiLU] -> dw kxk 'SAME' + SiLU -> SE (global mean -> FC + SiLU -> FC + sigmoid -> scale)
        -> 1x1 project (+ x): BatchNorms folded, SiLU / residual fused into the conv epilogues, the SE bottleneck is one
        fp32 launch (prv2_se_gate)."""
        P = self._packed
        x = ops.conv2d(crop, P["stem"], act=ops.ACT_SILU)
        feats: List[Feat] = []
        for B in P["blocks"]:
            h = ops.conv2d(x, B["expand"], act=ops.ACT_SILU) if B["kind"] == "ir" else x
            h = ops.dwconv2d(h, B["dw"][0], B["dw"][1], B["k"], B["s"], act=ops.ACT_SILU, same_pad=True)
            ops.channel_scale_(h, ops.se_gate(ops.global_avgpool(h), *B["se"]))
            x = ops.conv2d(h, B["proj"], res=x if B["res"] else None)
            if B["tap"]:
                feats.append(x)
        sizes = [(feats[0].h * 2, feats[0].w * 2)] + [(f.h, f.w) for f in feats]
        return [None] + feats, sizes

    __call__ = forward
