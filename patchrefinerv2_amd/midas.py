"""MiDaS v3.1 ``DPT_BEiT_L_384`` -- the core of ``coarse_branch.type == 'ZoeDepth'`` -- on the HIP kernels.

Host-side mirror of ``MidasCore`` (external/zoedepth/models/base_models/midas.py:191-355) over the network the reference
fetches with ``torch.hub.load("AyaanShah2204/MiDaS", "DPT_BEiT_L_384")`` (:342-347; un-vendored -- restated from the
published MiDaS 3.1 code, see oracle/midas_beit.py, and pinned there against HuggingFace transformers' independent port).
Same state-dict names under ``core.core.`` (``pretrained.model.blocks.{i}.attn.relative_position_bias_table``,
``pretrained.act_postprocess1.0.project.0.weight``, ``scratch.refinenet1.resConfUnit1.conv1.weight``,
``scratch.output_conv.0.weight`` ...), same six hooked tensors (midas.py:296-318).

What runs where: patch embed / qkv / proj / MLP / readout-project linears and every conv on the matrix kernels; the
relative-position-bias attention on ``prv2_attention_bias``; LayerNorm, upsample, token assembly on their kernels.  The
per-block bias [heads, N, N] is a constant of (weights, input size): it is expanded from the 47 x 47 table once per input
size on the host (the reference recomputes it in every block of every forward, beit.py::_get_rel_pos_bias) and cached in HBM
(24 x 38 MB at 384 x 512).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import ops
from . import weights as W
from .dav2 import StateDictModule
from .ops import ACT_GELU, ACT_NONE, ACT_RELU, Feat

MIDAS_MEAN = (0.5, 0.5, 0.5)   # PrepForMidas (midas.py:181-182)
MIDAS_STD = (0.5, 0.5, 0.5)


def _relative_position_index(wh: int, ww: int) -> torch.Tensor:
    """midas/backbones/beit.py::gen_relative_position_index (host, constant per window)"""
    nrd = (2 * wh - 1) * (2 * ww - 1) + 3
    coords = torch.stack(torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += wh - 1
    rel[:, :, 1] += ww - 1
    rel[:, :, 0] *= 2 * ww - 1
    idx = torch.zeros((wh * ww + 1,) * 2, dtype=rel.dtype)
    idx[1:, 1:] = rel.sum(-1)
    idx[0, 0:] = nrd - 3
    idx[0:, 0] = nrd - 2
    idx[0, 0] = nrd - 1
    return idx


class MidasBeitCore(StateDictModule):
    input_mean, input_std = MIDAS_MEAN, MIDAS_STD

    def __init__(self, model_type="DPT_BEiT_L_384", beit=None, device="cuda", prec="f32"):
        super().__init__()
        self.cfg = W.midas_beit_cfg(model_type, **(beit or {}))
        if self.cfg["dim"] != 64 * self.cfg["heads"]:
            raise NotImplementedError("the attention kernel is built for head_dim 64 (BEiT-L: 1024 / 16)")
        self.device = torch.device(device)
        self.prec = ops.L.PREC_NAMES[prec] if isinstance(prec, str) else prec
        self._spec = W.midas_beit_spec("", self.cfg)
        self._packed = None
        self._bias_cache: Dict[tuple, list] = {}
        self._zero_pos: Dict[tuple, torch.Tensor] = {}

    # -- weights ------------------------------------------------------------------------------------
    def _pack(self):
        if len(self._sd) < len(self._spec):
            return
        c = self.cfg
        D, p = c["dim"], c["patch"]
        sd = self._sd
        m = "pretrained.model."
        P = {}
        wpe = sd[m + "patch_embed.proj.weight"].permute(0, 2, 3, 1).reshape(D, p * p * 3)  # (ky, kx, c) columns
        P["patch_embed"] = ops.pack_conv(wpe, sd[m + "patch_embed.proj.bias"], device=self.device, prec=self.prec)
        P["cls"] = self._dev(m + "cls_token").view(-1)
        blocks = []
        for i in range(c["depth"]):
            b = f"{m}blocks.{i}."
            qkv_bias = torch.cat((sd[b + "attn.q_bias"], torch.zeros_like(sd[b + "attn.v_bias"]), sd[b + "attn.v_bias"]))
            blocks.append(dict(
                n1w=self._dev(b + "norm1.weight"), n1b=self._dev(b + "norm1.bias"),
                qkv=ops.pack_conv(sd[b + "attn.qkv.weight"], qkv_bias, device=self.device, prec=self.prec),
                proj=self._conv(b + "attn.proj"), g1=self._dev(b + "gamma_1"),
                n2w=self._dev(b + "norm2.weight"), n2b=self._dev(b + "norm2.bias"),
                fc1=self._conv(b + "mlp.fc1"), fc2=self._conv(b + "mlp.fc2"), g2=self._dev(b + "gamma_2")))
        P["blocks"] = blocks
        post = []
        for i in range(4):
            a = f"pretrained.act_postprocess{i + 1}."
            e = dict(readout=self._conv(a + "0.project.0"), proj=self._conv(a + "3"))
            if i == 0:
                e["resize"] = self._conv(a + "4", convt_k=4)
            elif i == 1:
                e["resize"] = self._conv(a + "4", convt_k=2)
            elif i == 3:
                e["resize"] = self._conv(a + "4", stride=2, pad=1)
            post.append(e)
        P["post"] = post
        s = "scratch."
        P["layer_rn"] = [self._conv(f"{s}layer{i + 1}_rn") for i in range(4)]
        P["refine"] = {}
        for r in (1, 2, 3, 4):
            rb = f"{s}refinenet{r}."
            P["refine"][r] = dict(out_conv=self._conv(rb + "out_conv"),
                                  u1c1=self._conv(rb + "resConfUnit1.conv1"), u1c2=self._conv(rb + "resConfUnit1.conv2"),
                                  u2c1=self._conv(rb + "resConfUnit2.conv1"), u2c2=self._conv(rb + "resConfUnit2.conv2"))
        P["out0"] = self._conv(s + "output_conv.0")
        P["out2"] = self._conv(s + "output_conv.2")
        P["out4_w"], P["out4_b"] = self._dev(s + "output_conv.4.weight"), self._dev(s + "output_conv.4.bias")
        self._packed = P
        self._bias_cache.clear()

    def _rel_pos_bias(self, gh: int, gw: int):
        """per block [heads, N, ld] (ld = N rounded up to 64 keys, zero padded): beit.py::_get_rel_pos_bias, once per size"""
        key = (gh, gw)
        if key not in self._bias_cache:
            c = self.cfg
            oh, ow = 2 * c["window"][0] - 1, 2 * c["window"][1] - 1
            nh, nw = 2 * gh - 1, 2 * gw - 1
            idx = _relative_position_index(gh, gw).view(-1)
            n = gh * gw + 1
            ld = ops.roundup(n, 64)
            out = []
            for i in range(c["depth"]):
                table = self._sd[f"pretrained.model.blocks.{i}.attn.relative_position_bias_table"].float().cpu()
                sub = table[:oh * ow].reshape(1, ow, oh, -1).permute(0, 3, 1, 2)  # (old_width, old_height): as the source has it
                new = F.interpolate(sub, size=(nh, nw), mode="bilinear").permute(0, 2, 3, 1).reshape(nh * nw, -1)
                full = torch.cat([new, table[oh * ow:]])
                bias = torch.zeros((c["heads"], n, ld), dtype=torch.float32)
                bias[:, :, :n] = full[idx].view(n, n, -1).permute(2, 0, 1)
                bias = bias.to(self.device)
                # bf16 modes: the rows re-ordered once for the attention kernel's coalesced loads (ops.pack_attention_bias)
                out.append(ops.pack_attention_bias(bias, n) if (self.prec != ops.PREC_F32 and ops.ATT_BIAS_IMAGE) else bias)
            self._bias_cache[key] = out
        return self._bias_cache[key]

    # -- forward ------------------------------------------------------------------------------------
    def normalize_nchw(self, x: torch.Tensor) -> Feat:
        """PrepForMidas with do_resize=False: (x - 0.5) / 0.5, fused into the NCHW -> NHWC layout kernel"""
        B, _, H, Wd = x.shape
        out = Feat.alloc(B, H, Wd, 3, x.device, pad_to=4)
        zero = torch.zeros((1, 2), dtype=torch.int32, device=x.device)
        for b in range(B):
            ops.crop_resize(x[b].contiguous(), zero, H, Wd, H, Wd, MIDAS_MEAN, MIDAS_STD, out.batch(b, b + 1))
        return out

    def _rcu(self, p, tag, x: Feat, res2: Optional[Feat] = None) -> Feat:
        t = ops.conv2d(x, p[tag + "c1"], relu_in=True)
        return ops.conv2d(t, p[tag + "c2"], relu_in=True, res=x, res2=res2)

    def _fusion_block(self, p, xs, size) -> Feat:
        """FeatureFusionBlock_custom.forward (midas/blocks.py): the 1x1 out_conv commutes with the bilinear resize"""
        out = xs[0]
        if len(xs) == 2:
            out = self._rcu(p, "u1", xs[1], res2=xs[0])
        out = self._rcu(p, "u2", out)
        return ops.upsample_bilinear(ops.conv2d(out, p["out_conv"]), size[0], size[1])

    def forward_nhwc(self, xn: Feat, out_conv_dest: Optional[Feat] = None) -> dict:
        """xn: (x - 0.5) / 0.5 as NHWC [B, H, W, >= 3]; H, W multiples of 32.  Returns rel_depth [B,1,H,W] and
        feats = [out_conv, l4_rn, r4, r3, r2, r1] (MidasCore.layer_names, midas.py:193)."""
        P = self._packed
        if P is None:
            raise RuntimeError("MidasBeitCore: weights not loaded")
        c = self.cfg
        D, p, heads = c["dim"], c["patch"], c["heads"]
        B, H, Wd = xn.n, xn.h, xn.w
        if H % 32 or Wd % 32:
            raise ValueError(f"MiDaS DPT-BEiT input {H}x{Wd}: height and width must be multiples of 32 (PrepForMidas, midas.py:183)")
        gh, gw = H // p, Wd // p
        npatch, N = gh * gw, gh * gw + 1
        rows = ops.patchify(Feat(xn.buf, 3, xn.c0), p, ops.roundup(p * p * 3, 32))
        emb = ops.linear(rows, P["patch_embed"])
        if (N, D) not in self._zero_pos:  # BEiT has no absolute position embedding
            self._zero_pos[(N, D)] = torch.zeros((N, D), device=self.device)
        x = ops.assemble_tokens(emb, P["cls"], self._zero_pos[(N, D)], B, npatch, D).view(B * N, D)
        M = B * N
        h = torch.empty_like(x)
        biases = self._rel_pos_bias(gh, gw)
        taps = []
        use_ss = self.prec == ops.L.PREC_BF16X3 and M >= ops.SS_MIN_ROWS and D % 32 == 0 and not ops.SS_DISABLED  # (see dav2.py)
        for i, blk in enumerate(P["blocks"]):
            if use_ss:
                ops.layernorm_ss(x, M, D, D, blk["n1w"], blk["n1b"], 1e-6, h)
                if ops.QKV_SS:  # (no pre-pass: the Linear writes the attention kernel's operands; same bits)
                    a = ops.attention_qkv_ss(ops.gemm_ss_qkv(h, blk["qkv"], heads), B, N, heads, bias=biases[i])
                else:
                    qkv = ops.gemm_ss(h, blk["qkv"])
                    a = ops.attention(qkv, B, N, heads, self.prec, bias=biases[i], out_ss=True)
                ops.gemm_ss(a, blk["proj"], out=x, gamma=blk["g1"], res=x)
                ops.layernorm_ss(x, M, D, D, blk["n2w"], blk["n2b"], 1e-6, h)
                f = ops.gemm_ss(h, blk["fc1"], act=ACT_GELU, out_ss=True)
                ops.gemm_ss(f, blk["fc2"], out=x, gamma=blk["g2"], res=x)
            else:
                ops.layernorm_rows(x, M, D, D, blk["n1w"], blk["n1b"], 1e-6, ACT_NONE, h, D)
                qkv = ops.linear(h, blk["qkv"])
                a = ops.attention(qkv, B, N, heads, self.prec, bias=biases[i])
                ops.linear(a, blk["proj"], out=x, gamma=blk["g1"], res=x)            # x += gamma_1 * proj(attn)
                ops.layernorm_rows(x, M, D, D, blk["n2w"], blk["n2b"], 1e-6, ACT_NONE, h, D)
                f = ops.linear(h, blk["fc1"], act=ACT_GELU)
                ops.linear(f, blk["fc2"], out=x, gamma=blk["g2"], res=x)             # x += gamma_2 * mlp
            if i in c["taps"]:
                taps.append(x.clone())  # forward hook on blocks[i]: the raw block output
        layers = []
        for i, t in enumerate(taps):
            e = P["post"][i]
            t3 = t.view(B, N, D)
            # ProjectReadout: Linear(2D -> D) + GELU on [token | cls] (data movement only: the concat of the reference)
            cat = torch.cat((t3[:, 1:], t3[:, :1].expand(-1, npatch, -1)), dim=-1).reshape(B * npatch, 2 * D)
            f = ops.linear(cat, e["readout"], act=ACT_GELU)
            y = ops.conv2d(Feat(f.view(B, gh, gw, D)), e["proj"])
            if "resize" in e:
                y = ops.conv2d(y, e["resize"])
            layers.append(y)
        rn = [ops.conv2d(layers[i], P["layer_rn"][i]) for i in range(4)]
        R = P["refine"]
        r4 = self._fusion_block(R[4], [rn[3]], (rn[2].h, rn[2].w))
        r3 = self._fusion_block(R[3], [r4, rn[2]], (rn[1].h, rn[1].w))
        r2 = self._fusion_block(R[2], [r3, rn[1]], (rn[0].h, rn[0].w))
        r1 = self._fusion_block(R[1], [r2, rn[0]], (rn[0].h * 2, rn[0].w * 2))
        o = ops.upsample_bilinear(ops.conv2d(r1, P["out0"]), r1.h * 2, r1.w * 2)
        out_conv = ops.conv2d(o, P["out2"], out_conv_dest, act=ACT_RELU)
        rel = ops.conv2d_cout1(out_conv, P["out4_w"], P["out4_b"], 1, act=ACT_RELU)
        return dict(rel_depth=rel, feats=[out_conv, rn[3], r4, r3, r2, r1])
