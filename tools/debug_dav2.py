import torch, torch.nn.functional as F, sys
sys.path.insert(0, ".")
from oracle import dav2 as od
from oracle.cases import *
from patchrefinerv2_amd import weights as W, ops
from patchrefinerv2_amd.dav2 import DepthAnythingV2, _TokenMap
from patchrefinerv2_amd.ops import Feat
torch.set_grad_enabled(False)
sd = tiny_dav2_sd(); cfg = W.dav2_cfg(TINY_DAV2["model_cfg"]); vit = cfg["vit"]
m = DepthAnythingV2(**TINY_DAV2["model_cfg"]); m.load_state_dict(sd)
x = rand_image(1, 2, 56, 84)
def d(name, got, ref): print(f"{name:28s} max|d|={float((got.cpu()-ref).abs().max()):.3e} ref_max={float(ref.abs().max()):.3f}")
mean = torch.tensor(od.IMAGENET_MEAN).view(-1,1,1); std = torch.tensor(od.IMAGENET_STD).view(-1,1,1)
xn_ref = (x-mean)/std
xn = m.normalize_nchw(x.cuda())
d("normalize", xn.to_nchw(), xn_ref)
p="pretrained."
t = F.conv2d(xn_ref, sd[p+"patch_embed.proj.weight"], sd[p+"patch_embed.proj.bias"], stride=14).flatten(2).transpose(1,2)
P = m._packed
rows = ops.patchify(Feat(xn.buf, 3, xn.c0), 14, ops.roundup(588, 32))
emb = ops.linear(rows, P["patch_embed"])
d("patch_embed", emb.view(2,-1,64), t)
tok_ref = torch.cat((sd[p+"cls_token"].expand(2,-1,-1), t), 1)
pos_ref = od.interpolate_pos_encoding(sd[p+"pos_embed"], tok_ref.shape[1]-1, 56, 84, 14)
d("pos", m._pos(56,84), pos_ref[0])
tok_ref = tok_ref + pos_ref
B, N, D = 2, tok_ref.shape[1], 64
xg = ops.assemble_tokens(emb, P["cls"], m._pos(56,84), B, N-1, D).view(B*N, D)
d("tokens", xg.view(B,N,D), tok_ref)
blk = P["blocks"][0]; bp = p+"blocks.0."
h = torch.empty_like(xg)
ops.layernorm_rows(xg, B*N, D, D, blk["n1w"], blk["n1b"], 1e-6, 0, h, D)
h_ref = F.layer_norm(tok_ref, (D,), sd[bp+"norm1.weight"], sd[bp+"norm1.bias"], 1e-6)
d("ln1", h.view(B,N,D), h_ref)
qkv = ops.linear(h, blk["qkv"])
qkv_ref = F.linear(h_ref, sd[bp+"attn.qkv.weight"], sd[bp+"attn.qkv.bias"])
d("qkv", qkv.view(B,N,3*D), qkv_ref)
a = ops.attention(qkv, B, N, vit["heads"])
qq = qkv_ref.reshape(B,N,3,vit["heads"],D//vit["heads"]).permute(2,0,3,1,4)
att = ((qq[0]*(32**-0.5)) @ qq[1].transpose(-2,-1)).softmax(-1) @ qq[2]
a_ref = att.transpose(1,2).reshape(B,N,D)
d("attn", a.view(B,N,D), a_ref)
x1 = od.block(sd, bp, tok_ref, vit["heads"])
out = m(x.cuda()); ref = od.dav2_forward(sd, "", x, cfg)
d("depth", out["metric_depth"], ref["metric_depth"])
for k in ref["temp_features"]: d(k, out["temp_features"][k].to_nchw(), ref["temp_features"][k])
