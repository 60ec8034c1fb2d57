"""Oracle (test infrastructure): ZoeDepth single metric-bins head over the DepthAnything (v1) core.

ZoeDepth.forward            external/zoedepth/models/zoedepth/zoedepth_v1.py:125-233
DepthAnythingCore.forward   external/zoedepth/models/base_models/depth_anything.py:262-278 (hooks :299-321)
DPT_DINOv2 / DPTHead (v1)   external/depth_anything/dpt.py:22-165
SeedBinRegressorUnnormed    external/zoedepth/models/layers/localbins_layers.py:73-96
Projector                   external/zoedepth/models/layers/localbins_layers.py:99-117
AttractorLayerUnnormed      external/zoedepth/models/layers/attractor.py:139-208 (inv_attractor :45-57; the
                            call passes no alpha/gamma, so the defaults 300 / 2 always apply -- SURVEY Q7)
ConditionalLogBinomial      external/zoedepth/models/layers/dist_layers.py:72-120 (log_binom :29-33, LogBinomial :36-69)
The MiDaS DPT-BEiT-L core (torch.hub, un-vendored) is restated in oracle/midas_beit.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import dav2
from .ops import bilinear_ac


def da_v1_core(sd, p, x, cfg):
    """DPT_DINOv2.forward with the DepthAnythingCore hooks.  x: [B,3,H,W] in [0,1] (PrepForMidas with
    do_resize=False only normalises).  Returns rel_depth [B,H,W] and [out_conv, l4_rn, r4, r3, r2, r1]."""
    mean = torch.tensor(dav2.IMAGENET_MEAN, dtype=x.dtype).view(-1, 1, 1)
    std = torch.tensor(dav2.IMAGENET_STD, dtype=x.dtype).view(-1, 1, 1)
    x = (x - mean) / std
    h, w = x.shape[-2:]
    ph, pw = h // 14, w // 14
    vit = dict(cfg["vit"])
    vit["taps"] = list(range(vit["depth"] - 4, vit["depth"]))  # get_intermediate_layers(x, 4, ...)
    feats = dav2.dinov2_intermediate(sd, p + "pretrained.", x, vit)
    hp = p + "depth_head."
    out = []
    for i, (t, _cls) in enumerate(feats):
        t = t.permute(0, 2, 1).reshape(t.shape[0], t.shape[-1], ph, pw)
        t = F.conv2d(t, sd[f"{hp}projects.{i}.weight"], sd[f"{hp}projects.{i}.bias"])
        if i == 0:
            t = F.conv_transpose2d(t, sd[hp + "resize_layers.0.weight"], sd[hp + "resize_layers.0.bias"], stride=4)
        elif i == 1:
            t = F.conv_transpose2d(t, sd[hp + "resize_layers.1.weight"], sd[hp + "resize_layers.1.bias"], stride=2)
        elif i == 3:
            t = F.conv2d(t, sd[hp + "resize_layers.3.weight"], sd[hp + "resize_layers.3.bias"], stride=2, padding=1)
        out.append(t)
    s = hp + "scratch."
    rn = [F.conv2d(out[i], sd[f"{s}layer{i + 1}_rn.weight"], None, padding=1) for i in range(4)]
    r4 = dav2.feature_fusion_block(sd, s + "refinenet4.", [rn[3]], size=rn[2].shape[2:])
    r3 = dav2.feature_fusion_block(sd, s + "refinenet3.", [r4, rn[2]], size=rn[1].shape[2:])
    r2 = dav2.feature_fusion_block(sd, s + "refinenet2.", [r3, rn[1]], size=rn[0].shape[2:])
    r1 = dav2.feature_fusion_block(sd, s + "refinenet1.", [r2, rn[0]])
    o = F.conv2d(r1, sd[s + "output_conv1.weight"], sd[s + "output_conv1.bias"], padding=1)
    o = bilinear_ac(o, (ph * 14, pw * 14))
    out_conv = F.relu(F.conv2d(o, sd[s + "output_conv2.0.weight"], sd[s + "output_conv2.0.bias"], padding=1))
    d = F.relu(F.conv2d(out_conv, sd[s + "output_conv2.2.weight"], sd[s + "output_conv2.2.bias"]))
    d = F.relu(bilinear_ac(d, (h, w)))
    return d.squeeze(1), [out_conv, rn[3], r4, r3, r2, r1]


def _mlp(sd, p, x, act_out=None):
    x = F.relu(F.conv2d(x, sd[p + "_net.0.weight"], sd[p + "_net.0.bias"]))
    x = F.conv2d(x, sd[p + "_net.2.weight"], sd[p + "_net.2.bias"])
    return F.softplus(x) if act_out == "softplus" else x


def inv_attractor(dx, alpha: float = 300.0, gamma: int = 2):
    return dx.div(1 + alpha * dx.pow(gamma))


def attractor_unnormed(sd, p, x, b_prev, prev_emb, n_attractors):
    prev_emb = bilinear_ac(prev_emb, x.shape[-2:])
    x = x + prev_emb
    A = _mlp(sd, p, x, "softplus")
    b_centers = bilinear_ac(b_prev, A.shape[-2:])
    delta = torch.mean(inv_attractor(A.unsqueeze(2) - b_centers.unsqueeze(1)), dim=1)  # kind='mean'
    return b_centers + delta


def log_binom(n, k, eps=1e-7):
    n = n + eps
    k = k + eps
    return n * torch.log(n) - k * torch.log(k) - (n - k) * torch.log(n - k + eps)


def conditional_log_binomial(sd, p, x, cond, n_classes, min_temp, max_temp, p_eps=1e-4):
    pt = torch.cat((x, cond), dim=1)
    pt = F.gelu(F.conv2d(pt, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"]))
    pt = F.softplus(F.conv2d(pt, sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"]))
    pp, t = pt[:, :2], pt[:, 2:]
    pp = pp + p_eps
    pp = pp[:, 0] / (pp[:, 0] + pp[:, 1])
    t = t + p_eps
    t = (t[:, 0] / (t[:, 0] + t[:, 1])).unsqueeze(1)
    t = (max_temp - min_temp) * t + min_temp
    xx = pp.unsqueeze(1)
    eps = 1e-4
    one_minus = torch.clamp(1 - xx, eps, 1)
    xx = torch.clamp(xx, eps, 1)
    k_idx = torch.arange(0, n_classes).view(1, -1, 1, 1)
    Km1 = torch.Tensor([n_classes - 1]).view(1, -1, 1, 1)
    y = log_binom(Km1, k_idx) + k_idx * torch.log(xx) + (n_classes - 1 - k_idx) * torch.log(one_minus)
    return torch.softmax(y / t, dim=1)


def zoedepth_forward(sd, prefix, x, cfg):
    """ZoeDepth.forward (bin_centers_type='softplus', attractor_kind='mean', attractor_type='inv')."""
    if "beit" in cfg["core"]:  # type='ZoeDepth': MidasCore over MiDaS DPT_BEiT_L_384 (oracle/midas_beit.py)
        from .midas_beit import midas_beit_core
        rel, out = midas_beit_core(sd, prefix + "core.core.", x, cfg["core"])
    else:
        rel, out = da_v1_core(sd, prefix + "core.core.", x, cfg["core"])
    outconv, btlnck, blocks = out[0], out[1], out[2:]
    x_d0 = F.conv2d(btlnck, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"])
    temp = dict(x_d0=x_d0)
    b_prev = _mlp(sd, prefix + "seed_bin_regressor.", x_d0, "softplus")
    prev_emb = _mlp(sd, prefix + "seed_projector.", x_d0)
    for i, xb in enumerate(blocks):
        emb = _mlp(sd, f"{prefix}projectors.{i}.", xb)
        temp[f"x_blocks_feat_{i}"] = xb
        b = attractor_unnormed(sd, f"{prefix}attractors.{i}.", emb, b_prev, prev_emb, cfg["n_attractors"][i])
        b_prev, prev_emb = b, emb
    temp["midas_final_feat"] = outconv
    rel_cond = bilinear_ac(rel.unsqueeze(1), outconv.shape[2:])
    last = torch.cat([outconv, rel_cond], dim=1)
    emb = bilinear_ac(emb, last.shape[-2:])
    probs = conditional_log_binomial(sd, prefix + "conditional_log_binomial.", last, emb, cfg["n_bins"], cfg["min_temp"],
                                     cfg["max_temp"])
    centers = bilinear_ac(b, probs.shape[-2:])
    depth = torch.sum(probs * centers, dim=1, keepdim=True)
    return dict(metric_depth=depth, temp_features=temp)
