"""Depth-Anything-V2 (DINOv2 ViT + DPT head) on the HIP kernels.

Host-side mirror of the reference module ``DepthAnythingV2``
(external/depth_anything_v2/dpt.py:153-203; DinoVisionTransformer dinov2.py:44-321; DPTHead
dpt.py:38-150; FeatureFusionBlock/ResidualConvUnit util/blocks.py:28-148): same constructor
arguments, same state-dict names, same ``forward`` output dict.  All arithmetic runs in the
C-ABI kernels (patchrefinerv2_amd/csrc); torch only owns the device buffers.

Internal activations are NHWC (``ops.Feat``); ``temp_features`` therefore holds ``Feat`` objects
-- their consumers are this package's own ROI-gather / conv kernels.  ``metric_depth`` is a
dense [B,1,H,W] tensor exactly as in the reference.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import ops
from . import weights as W
from .ops import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, Feat

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class StateDictModule:
    """Name/shape contract + load/save, shared by the HIP-backed modules."""

    def __init__(self):
        self._spec: "OrderedDict[str, tuple]" = OrderedDict()
        self._sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self._children: Dict[str, "StateDictModule"] = {}
        self.device = torch.device("cuda")
        self.prec = ops.PREC_F32

    # -- reference-style API --------------------------------------------------------------
    def spec(self) -> "OrderedDict[str, tuple]":
        s = OrderedDict(self._spec)
        for name, ch in self._children.items():
            for k, v in ch.spec().items():
                s[f"{name}.{k}"] = v
        return s

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        s = OrderedDict(self._sd)
        for name, ch in self._children.items():
            for k, v in ch.state_dict().items():
                s[f"{name}.{k}"] = v
        return s

    def load_state_dict(self, sd, strict: bool = True):
        spec = self.spec()
        missing = [k for k in spec if k not in sd]
        unexpected = [k for k in sd if k not in spec]
        for k in spec:
            if k in sd and tuple(sd[k].shape) != tuple(spec[k]):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(sd[k].shape)} vs model {tuple(spec[k])}")
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]} ({len(missing)}), unexpected {unexpected[:5]} "
                               f"({len(unexpected)})")
        if torch.cuda.is_available() and self.device.type == "cuda":
            with torch.cuda.device(self.device):  # weight packing kernels run on the model's device
                self._assign(sd)
        else:
            self._assign(sd)
        return dict(missing_keys=missing, unexpected_keys=unexpected)

    def _assign(self, sd, prefix: str = ""):
        mine = {k: sd[prefix + k].detach().to(torch.float32) for k in self._spec if prefix + k in sd}
        self._sd.update(mine)
        for name, ch in self._children.items():
            ch._assign(sd, f"{prefix}{name}.")
        if mine:
            self._pack()

    def _pack(self):  # device-side packing of the weights that changed
        raise NotImplementedError

    def to(self, device):
        self.device = torch.device(device)
        return self

    def eval(self):
        return self

    def cuda(self):
        return self.to("cuda")

    def _dev(self, name):
        return self._sd[name].to(self.device).contiguous()

    def _conv(self, name, **kw):
        b = self._sd.get(name + ".bias")
        return ops.pack_conv(self._sd[name + ".weight"], b, device=self.device, prec=self.prec, **kw)


def interpolate_pos_encoding(pos_embed: torch.Tensor, h: int, w: int, patch: int, offset: float = 0.1) -> torch.Tensor:
    """dinov2.py:179-210 (constant per input size: computed once on the host, cached on device).
    The reference passes (H, W) under the names (w, h); follow the code, not the names."""
    N = pos_embed.shape[1] - 1
    npatch = (h // patch) * (w // patch)
    if npatch == N and h == w:
        return pos_embed[0]
    pe = pos_embed.float().cpu()
    dim = pe.shape[-1]
    w0, h0 = h // patch + offset, w // patch + offset
    sq = math.sqrt(N)
    grid = pe[:, 1:].reshape(1, int(sq), int(sq), dim).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=(float(w0) / sq, float(h0) / sq), mode="bicubic", antialias=False)
    assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
    grid = grid.permute(0, 2, 3, 1).reshape(-1, dim)
    return torch.cat((pe[0, :1], grid), dim=0)


class DepthAnythingV2(StateDictModule):
    def __init__(self, encoder="vitl", features=256, out_channels=(256, 512, 1024, 1024), use_bn=False,
                 use_clstoken=False, max_depth=20.0, vit=None, device="cuda", prec="f32", variant="v2"):
        """variant='v2': DepthAnythingV2 (sigmoid * max_depth head, taps dpt.py:165-170);
        variant='v1': DPT_DINOv2 of external/depth_anything/dpt.py:133-165 (last-4-blocks taps, ReLU head),
        the core of the 'DA-ZoeDepth' coarse branch."""
        super().__init__()
        self.variant = variant
        if use_bn or use_clstoken:
            raise NotImplementedError("use_bn / use_clstoken are never set on the inference path")
        self.cfg = W.dav2_cfg(dict(encoder=encoder, features=features, out_channels=list(out_channels),
                                   max_depth=max_depth, vit=vit or {}))
        if self.cfg["vit"]["dim"] != 64 * self.cfg["vit"]["heads"]:
            raise NotImplementedError("the attention kernel is built for head_dim 64 (every DINOv2 size: "
                                      "384/6, 768/12, 1024/16)")
        if variant == "v1":
            d = self.cfg["vit"]["depth"]
            self.cfg["vit"]["taps"] = list(range(d - 4, d))  # get_intermediate_layers(x, 4, ...) (depth_anything/dpt.py:152)
        self.max_depth = float(max_depth)
        self.encoder = encoder
        self.device = torch.device(device)
        self.prec = ops.L.PREC_NAMES[prec] if isinstance(prec, str) else prec
        self._spec = W.dav2_spec("", dict(encoder=encoder, features=features, out_channels=list(out_channels),
                                          vit=vit or {}))
        self._pos_cache: Dict[tuple, torch.Tensor] = {}
        self._packed = None

    # -- weights ------------------------------------------------------------------------------
    def _pack(self):
        if len(self._sd) < len(self._spec):
            return  # partially loaded; pack when complete
        vit = self.cfg["vit"]
        D, p = vit["dim"], vit["patch"]
        P = {}
        wpe = self._sd["pretrained.patch_embed.proj.weight"].permute(0, 2, 3, 1).reshape(D, p * p * 3)
        P["patch_embed"] = ops.pack_conv(wpe, self._sd["pretrained.patch_embed.proj.bias"], device=self.device,
                                         prec=self.prec)
        P["cls"] = self._dev("pretrained.cls_token").view(-1)
        blocks = []
        for i in range(vit["depth"]):
            b = f"pretrained.blocks.{i}."
            blocks.append(dict(
                n1w=self._dev(b + "norm1.weight"), n1b=self._dev(b + "norm1.bias"),
                qkv=self._conv(b + "attn.qkv"), proj=self._conv(b + "attn.proj"), ls1=self._dev(b + "ls1.gamma"),
                n2w=self._dev(b + "norm2.weight"), n2b=self._dev(b + "norm2.bias"),
                fc1=self._conv(b + "mlp.fc1"), fc2=self._conv(b + "mlp.fc2"), ls2=self._dev(b + "ls2.gamma")))
        P["blocks"] = blocks
        P["norm_w"], P["norm_b"] = self._dev("pretrained.norm.weight"), self._dev("pretrained.norm.bias")
        h = "depth_head."
        P["projects"] = [self._conv(f"{h}projects.{i}") for i in range(4)]
        P["resize0"] = self._conv(h + "resize_layers.0", convt_k=4)
        P["resize1"] = self._conv(h + "resize_layers.1", convt_k=2)
        P["resize3"] = self._conv(h + "resize_layers.3", stride=2, pad=1)
        s = h + "scratch."
        P["layer_rn"] = [self._conv(f"{s}layer{i + 1}_rn") for i in range(4)]
        P["refine"] = {}
        for r in (1, 2, 3, 4):
            rb = f"{s}refinenet{r}."
            P["refine"][r] = dict(
                out_conv=self._conv(rb + "out_conv"),
                u1c1=self._conv(rb + "resConfUnit1.conv1"), u1c2=self._conv(rb + "resConfUnit1.conv2"),
                u2c1=self._conv(rb + "resConfUnit2.conv1"), u2c2=self._conv(rb + "resConfUnit2.conv2"))
        P["out1"] = self._conv(s + "output_conv1")
        P["out2_0"] = self._conv(s + "output_conv2.0")
        P["out2_2_w"] = self._dev(s + "output_conv2.2.weight")
        P["out2_2_b"] = self._dev(s + "output_conv2.2.bias")
        self._packed = P
        self._pos_cache.clear()

    def _pos(self, h, w):
        key = (h, w)
        if key not in self._pos_cache:
            pe = interpolate_pos_encoding(self._sd["pretrained.pos_embed"], h, w, self.cfg["vit"]["patch"])
            self._pos_cache[key] = pe.to(self.device).contiguous()
        return self._pos_cache[key]

    # -- forward ---------------------------------------------------------------------------------
    def normalize_nchw(self, x: torch.Tensor) -> Feat:
        """(x - mean) / std (dpt.py:183) fused into the NCHW->NHWC layout kernel."""
        B, _, H, Wd = x.shape
        out = Feat.alloc(B, H, Wd, 3, x.device, pad_to=4)
        zero = torch.zeros((1, 2), dtype=torch.int32, device=x.device)
        for b in range(B):
            ops.crop_resize(x[b].contiguous(), zero, H, Wd, H, Wd, IMAGENET_MEAN, IMAGENET_STD, out.batch(b, b + 1))
        return out

    def _rcu(self, p, tag, x: Feat, res2: Optional[Feat] = None) -> Feat:
        """ResidualConvUnit (util/blocks.py:57-80): conv2(relu(conv1(relu(x)))) + x (+ res2)."""
        t = ops.conv2d(x, p[tag + "c1"], relu_in=True)
        return ops.conv2d(t, p[tag + "c2"], relu_in=True, res=x, res2=res2)

    def _fusion_block(self, p, xs: List[Feat], size) -> Feat:
        """FeatureFusionBlock.forward (util/blocks.py:123-148)."""
        out = xs[0]
        if len(xs) == 2:
            out = self._rcu(p, "u1", xs[1], res2=xs[0])  # xs[0] + resConfUnit1(xs[1])
        out = self._rcu(p, "u2", out)
        # the reference upsamples, then applies the 1x1 out_conv; both are linear and the bilinear weights
        # sum to one, so the order commutes up to rounding: conv at low resolution = 4x fewer FLOPs
        return ops.upsample_bilinear(ops.conv2d(out, p["out_conv"]), size[0], size[1])

    def forward_nhwc(self, xn: Feat, out_conv_dest: Optional[Feat] = None) -> dict:
        """xn: ImageNet-normalised NHWC input [B, H, W, >=3]; H, W multiples of 14."""
        P = self._packed
        if P is None:
            raise RuntimeError("DepthAnythingV2: weights not loaded")
        vit = self.cfg["vit"]
        D, p, heads = vit["dim"], vit["patch"], vit["heads"]
        B, H, Wd = xn.n, xn.h, xn.w
        assert H % p == 0 and Wd % p == 0, (H, Wd)
        gh, gw = H // p, Wd // p
        npatch, N = gh * gw, gh * gw + 1
        rows = ops.patchify(Feat(xn.buf, 3, xn.c0), p, ops.roundup(p * p * 3, 32))
        emb = ops.linear(rows, P["patch_embed"])
        x = ops.assemble_tokens(emb, P["cls"], self._pos(H, Wd), B, npatch, D).view(B * N, D)
        M = B * N
        h = torch.empty_like(x)
        taps = []
        # large token counts (a batch of tiles): every Linear input is produced directly in the matrix pipe's operand format
        # (ops.gemm_ss); same values bit for bit as the fp32-operand path below, so results do not depend on the batch
        use_ss = self.prec == ops.L.PREC_BF16X3 and M >= ops.SS_MIN_ROWS and D % 32 == 0 and not ops.SS_DISABLED
        for i, blk in enumerate(P["blocks"]):
            if use_ss:
                ops.layernorm_ss(x, M, D, D, blk["n1w"], blk["n1b"], 1e-6, h)
                if ops.QKV_SS:  # (no pre-pass: the Linear writes the attention kernel's operands; same bits)
                    a = ops.attention_qkv_ss(ops.gemm_ss_qkv(h, blk["qkv"], heads), B, N, heads)
                else:
                    qkv = ops.gemm_ss(h, blk["qkv"])
                    a = ops.attention(qkv, B, N, heads, self.prec, out_ss=True)
                ops.gemm_ss(a, blk["proj"], out=x, gamma=blk["ls1"], res=x)
                ops.layernorm_ss(x, M, D, D, blk["n2w"], blk["n2b"], 1e-6, h)
                f = ops.gemm_ss(h, blk["fc1"], act=ACT_GELU, out_ss=True)
                ops.gemm_ss(f, blk["fc2"], out=x, gamma=blk["ls2"], res=x)
            else:
                ops.layernorm_rows(x, M, D, D, blk["n1w"], blk["n1b"], 1e-6, ACT_NONE, h, D)
                qkv = ops.linear(h, blk["qkv"])
                a = ops.attention(qkv, B, N, heads, self.prec)
                ops.linear(a, blk["proj"], out=x, gamma=blk["ls1"], res=x)          # x += ls1 * proj(attn)
                ops.layernorm_rows(x, M, D, D, blk["n2w"], blk["n2b"], 1e-6, ACT_NONE, h, D)
                f = ops.linear(h, blk["fc1"], act=ACT_GELU)
                ops.linear(f, blk["fc2"], out=x, gamma=blk["ls2"], res=x)           # x += ls2 * fc2(gelu(fc1))
            if i in vit["taps"]:
                t = torch.empty_like(x)
                ops.layernorm_rows(x, M, D, D, P["norm_w"], P["norm_b"], 1e-6, ACT_NONE, t, D)
                taps.append(t)
        # DPT head (dpt.py:116-150); token rows 1.. of each image form an NHWC [gh, gw, D] map
        feats = []
        for i, t in enumerate(taps):
            tok = Feat(t.view(B, N, 1, D))
            tok_map = _TokenMap(tok, gh, gw, D)
            y = ops.conv2d(tok_map, P["projects"][i], x_bstride=N * D)
            if i == 0:
                y = ops.conv2d(y, P["resize0"])
            elif i == 1:
                y = ops.conv2d(y, P["resize1"])
            elif i == 3:
                y = ops.conv2d(y, P["resize3"])
            feats.append(y)
        rn = [ops.conv2d(feats[i], P["layer_rn"][i]) for i in range(4)]
        R = P["refine"]
        path4 = self._fusion_block(R[4], [rn[3]], (rn[2].h, rn[2].w))
        path3 = self._fusion_block(R[3], [path4, rn[2]], (rn[1].h, rn[1].w))
        path2 = self._fusion_block(R[2], [path3, rn[1]], (rn[0].h, rn[0].w))
        path1 = self._fusion_block(R[1], [path2, rn[0]], (rn[0].h * 2, rn[0].w * 2))
        o = ops.conv2d(path1, P["out1"])
        out_feat = ops.upsample_bilinear(o, gh * 14, gw * 14)
        o = ops.conv2d(out_feat, P["out2_0"], out_conv_dest, act=ACT_RELU)
        if self.variant == "v1":
            # ReLU(conv1x1) ; the interpolate to (h, w) is the identity for 14-multiples; ReLU again is idempotent
            rel = ops.conv2d_cout1(o, P["out2_2_w"], P["out2_2_b"], 1, act=ACT_RELU)
            return dict(rel_depth=rel, feats=[o, rn[3], path4, path3, path2, path1])
        depth = ops.conv2d_cout1(o, P["out2_2_w"], P["out2_2_b"], 1, act=ACT_SIGMOID, scale=self.max_depth)
        return dict(metric_depth=depth,
                    temp_features=dict(x_d0=rn[3], x_blocks_feat_0=path4, x_blocks_feat_1=path3,
                                       x_blocks_feat_2=path2, x_blocks_feat_3=path1, midas_final_feat=out_feat))

    def forward(self, x: torch.Tensor, **kwargs) -> dict:
        """x: [B,3,H,W] in [0,1] on the GPU.  Extra kwargs (``return_final_centers``) are accepted and
        ignored exactly like the reference (dpt.py:182)."""
        return self.forward_nhwc(self.normalize_nchw(x))

    __call__ = forward


class _TokenMap(Feat):
    """View of normalised tokens [B, N, D] as an NHWC [B, gh, gw, D] map skipping the cls row
    (dpt.py:125 ``x.permute(0,2,1).reshape(B, D, ph, pw)``) -- no copy: pointer offset + image stride."""

    __slots__ = ("_off",)

    def __init__(self, tok: Feat, gh: int, gw: int, d: int):
        self.buf = tok.buf
        self.n, self.h, self.w, self.c, self.c0 = tok.n, gh, gw, d, 0
        self.x2, self.aux = False, None
        self._off = d  # skip the cls token of image 0; x_bstride = N*D skips the others

    def view(self):
        """the strided tensor [B, gh, gw, D] the torch.ops.prv2 route takes (image stride N * D, first row skipped)"""
        ntok = self.buf.shape[1]
        return torch.as_strided(self.buf, (self.n, self.h, self.w, self.c), (ntok * self.c, self.w * self.c, self.c, 1), self.buf.storage_offset() + self._off)

    @property
    def ld(self):
        return self.c

    @property
    def ptr(self):
        return self.buf.data_ptr() + 4 * self._off
