"""ZoeDepth (single metric-bins head) over the vendored DepthAnything core, on the HIP kernels.

Host-side mirror of ``ZoeDepth`` (external/zoedepth/models/zoedepth/zoedepth_v1.py:39-311) built through
``ZoeDepth.build(**cfg)`` with ``midas_model_type in {'vits','vitb','vitl'}`` -> ``DepthAnythingCore``
(external/zoedepth/models/base_models/depth_anything.py:193-365), i.e. the reference's
``coarse_branch.type == 'DA-ZoeDepth'`` flavour.  Same state-dict names (``core.core.pretrained.*``,
``core.core.depth_head.*``, ``conv2``, ``seed_bin_regressor._net.*`` ...).

``midas_model_type='DPT_BEiT_L_384'`` (the default; ``coarse_branch.type == 'ZoeDepth'``) selects the MiDaS DPT-BEiT-L core
(patchrefinerv2_amd/midas.py): the head is the same, the inputs are then multiples of 32 and normalised with 0.5 / 0.5.
"""
from __future__ import annotations

import torch

from . import ops
from . import weights as W
from .dav2 import DepthAnythingV2, StateDictModule
from .ops import ACT_GELU, ACT_RELU, ACT_SOFTPLUS, Feat


class ZoeDepth(StateDictModule):
    def __init__(self, device="cuda", prec="f32", **cfg):
        super().__init__()
        self.zcfg = W.zoedepth_cfg(cfg)
        self.device = torch.device(device)
        self.prec = ops.L.PREC_NAMES[prec] if isinstance(prec, str) else prec
        core = self.zcfg["core"]
        if "beit" in core:  # midas_model_type='DPT_BEiT_L_384': MidasCore (midas.py:191) over MiDaS DPT-BEiT-L
            from .midas import MidasBeitCore
            self.core = MidasBeitCore(self.zcfg["core_type"], beit=cfg.get("beit"), device=device, prec=self.prec)
        else:
            self.core = DepthAnythingV2(encoder=core["encoder"], features=core["features"], out_channels=core["out_channels"],
                                        vit=cfg.get("vit"), device=device, prec=self.prec, variant="v1")
        self.input_mean = getattr(self.core, "input_mean", (0.485, 0.456, 0.406))  # PrepForMidas normalisation of the core
        self.input_std = getattr(self.core, "input_std", (0.229, 0.224, 0.225))
        self._children = {"core.core": self.core}
        full = W.zoedepth_spec("", cfg)
        self._spec = type(full)((k, v) for k, v in full.items() if not k.startswith("core.core."))
        self._packed = None

    @staticmethod
    def build(**cfg):
        return ZoeDepth(**cfg)

    def _pack(self):
        need = [k for k in self._spec if "log_binomial_transform" not in k]
        if any(k not in self._sd for k in need):
            return

        def mlp(name):
            return self._conv(name + "._net.0"), self._conv(name + "._net.2")

        P = dict(conv2=self._conv("conv2"), seed=mlp("seed_bin_regressor"), seed_proj=mlp("seed_projector"),
                 proj=[mlp(f"projectors.{i}") for i in range(4)], attr=[mlp(f"attractors.{i}") for i in range(4)],
                 clb0=self._conv("conditional_log_binomial.mlp.0"), clb2=self._conv("conditional_log_binomial.mlp.2"))
        self._packed = P

    def forward(self, x: torch.Tensor, return_final_centers=False, **kwargs) -> dict:
        """x: [B,3,H,W] in [0,1]; H, W multiples of 14 (DepthAnything cores) / 32 (MiDaS BEiT): PrepForMidas with
        do_resize=False only normalises."""
        P = self._packed
        if P is None:
            raise RuntimeError("ZoeDepth: weights not loaded")
        z = self.zcfg
        emb_dim = z["bin_embedding_dim"]
        return self.forward_nhwc(self.core.normalize_nchw(x))

    def forward_nhwc(self, xn: Feat) -> dict:
        """xn: normalised NHWC input [B, H, W, >=3] (what PrepForMidas hands to the core)"""
        P = self._packed
        if P is None:
            raise RuntimeError("ZoeDepth: weights not loaded")
        z = self.zcfg
        emb_dim = z["bin_embedding_dim"]
        B, H, Wd = xn.n, xn.h, xn.w
        x = xn
        # the conditional-log-binomial input [out_conv(32) | rel_depth(1) | b_embedding(emb)] as one buffer:
        # the DPT head writes its 32-channel out_conv feature straight into channels 0..31
        last = Feat.alloc(B, H, Wd, 32 + 1 + emb_dim, x.device)
        co = self.core.forward_nhwc(xn, out_conv_dest=last.slice(0, 32))
        rel, (outconv, btlnck, *blocks) = co["rel_depth"], co["feats"]
        x_d0 = ops.conv2d(btlnck, P["conv2"])
        b_prev = ops.conv2d(ops.conv2d(x_d0, P["seed"][0], act=ACT_RELU), P["seed"][1], act=ACT_SOFTPLUS)
        prev_emb = ops.conv2d(ops.conv2d(x_d0, P["seed_proj"][0], act=ACT_RELU), P["seed_proj"][1])
        for i, xb in enumerate(blocks):
            emb = ops.conv2d(ops.conv2d(xb, P["proj"][i][0], act=ACT_RELU), P["proj"][i][1])
            xa = ops.add(emb, ops.upsample_bilinear(prev_emb, emb.h, emb.w))     # x + interpolate(prev_b_embedding)
            A = ops.conv2d(ops.conv2d(xa, P["attr"][i][0], act=ACT_RELU), P["attr"][i][1], act=ACT_SOFTPLUS)
            b_prev = ops.zoe_attractor(A, ops.upsample_bilinear(b_prev, A.h, A.w), 300.0)  # defaults alpha=300, gamma=2 (Q7)
            prev_emb = emb
        ops.upsample_bilinear(Feat(rel.view(B, H, Wd, 1)), H, Wd, out=last.slice(32, 1))
        ops.upsample_bilinear(emb, H, Wd, out=last.slice(33, emb_dim))
        pt = ops.conv2d(ops.conv2d(last, P["clb0"], act=ACT_GELU), P["clb2"], act=ACT_SOFTPLUS)
        centers = ops.upsample_bilinear(b_prev, H, Wd)
        depth = ops.zoe_logbinom_depth(pt, centers, z["min_temp"], z["max_temp"])
        feats = dict(x_d0=x_d0, x_blocks_feat_0=blocks[0], x_blocks_feat_1=blocks[1], x_blocks_feat_2=blocks[2],
                     x_blocks_feat_3=blocks[3], midas_final_feat=outconv)
        return dict(metric_depth=depth, temp_features=feats)

    __call__ = forward
