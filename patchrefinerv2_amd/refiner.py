"""LightWeightRefiner (V2 per-patch encoder) on the HIP kernels.

Host-side mirror of ``LightWeightRefiner`` (estimator/models/blocks/lightweight_refiner.py:242-322)
with ``with_decoder=False`` and the ``mobilenetv4_conv_small`` encoder after the reference's
4-channel stem surgery (estimator/models/patchrefinerplus.py:159-165).

The encoder arithmetic is timm's (``timm.create_model(..., features_only=True)``,
lightweight_refiner.py:260-262) -- timm is not vendored in the reference: the architecture
is restated from the public MobileNetV4 definition (weights.py::MNV4_SMALL) and is PARITY
UNPINNED.  BatchNorm (eval) is folded into the preceding convolution at load time.
"""
from __future__ import annotations

from typing import List

import torch

from . import ops
from . import weights as W
from .dav2 import StateDictModule
from .ops import ACT_NONE, ACT_RELU, Feat

BN_EPS = 1e-5
SUPPORTED_ENCODERS = {"mobilenetv4_conv_small.e2400_r224_in1k": W.MNV4_SMALL, "mobilenetv4_conv_small": W.MNV4_SMALL}


class LightWeightRefiner(StateDictModule):
    def __init__(self, encoder_name, coarse_condition=True, with_decoder=False, cls_pretrain=True, device="cuda",
                 prec="f32", **_unused):
        super().__init__()
        if encoder_name not in SUPPORTED_ENCODERS:
            raise NotImplementedError(
                f"refiner encoder '{encoder_name}': only mobilenetv4_conv_small is built (EfficientNet-B5-AP / "
                "ConvNeXt-L are timm models not vendored in the reference; SURVEY.md 8f rank 3)")
        if with_decoder or not coarse_condition:
            raise NotImplementedError("with_decoder=True / coarse_condition=False are not used by any V2 config")
        self.encoder_name = encoder_name
        self.arch = SUPPORTED_ENCODERS[encoder_name]
        self.coarse_condition = True
        self.device = torch.device(device)
        self.prec = ops.L.PREC_NAMES[prec] if isinstance(prec, str) else prec
        self.layers, self.taps = W.mnv4_layers(self.arch, in_chans=4)
        self._spec = W.mnv4_spec("refiner_encoder.", self.arch, in_chans=4)
        self.mean, self.std = self.arch["mean"], self.arch["std"]
        self._packed = None

    def _pack(self):
        if len(self._sd) < len(self._spec):
            return
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
        x = crop
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

    __call__ = forward
