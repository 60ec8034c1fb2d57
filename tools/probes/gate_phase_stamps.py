"""see gate_phase_stamps.sh"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
stamps = torch.zeros(100000 * 80, dtype=torch.int64, device="cuda")
os.environ["PRV2_STAMP_PTR"] = hex(stamps.data_ptr())
from patchrefinerv2_amd import lib as L
L.LIB_PATH = os.environ["PRV2_LIB_OVERRIDE"]
from patchrefinerv2_amd import ops as P
PR = L.PREC_NAMES["bf16x3"]
NAMES = ["prologue", "main loop", "C tile -> LDS", "row stats", "normalise + split", "gate GEMM", "gate acc -> LDS", "store loop", "store drain"]
for n, h, w, cin in ((14, 192, 256, 512), (14, 96, 128, 512), (14, 192, 256, 256)):
    x = P.Feat(torch.randn(n, h, w, cin, device="cuda"))
    cw0 = P.pack_conv(torch.randn(256, cin, 3, 3, device="cuda") / (3 * cin ** 0.5), torch.randn(256, device="cuda"), pad=1, prec=PR)
    w3 = torch.randn(256, 256, 1, 1, device="cuda") / 16
    gw, gb = P.pack_gate(w3), torch.randn(256, device="cuda")
    ln = (torch.rand(256, device="cuda") + 0.5, torch.randn(256, device="cuda") * 0.1)
    mul, res = P.Feat(torch.randn(n, h, w, 256, device="cuda")), P.Feat(torch.randn(n, h, w, 256, device="cuda"))
    out = P.Feat.alloc(n, h, w, 256, "cuda")
    for _ in range(3):
        P.conv3x3_ln_gate(x, cw0, ln, gw, gb, out, act=P.ACT_RELU, mul=mul, res=res)
    torch.cuda.synchronize()
    nblk = n * (h // 8) * (w // 16)
    sw = stamps[: nblk * 80].view(nblk, 8, 10).cpu().double()
    s = sw[:, 0]  # wave 0
    d = (s[:, 1:] - s[:, :-1]) / 1000.0
    tot = (s[:, 9] - s[:, 0]) / 1000.0
    print(f"{n}x{h}x{w} {cin}->256->256: {nblk} workgroups, median {tot.median():.1f} kcycles (s_memtime units) per workgroup: " +
          "  ".join(f"{nm} {d[:, i].median():.2f}" for i, nm in enumerate(NAMES)), flush=True)
    rel = (sw - sw[:, :1, :1]) / 1000.0  # every wave's stamps relative to wave 0's kernel start
    for i in (4, 5, 6, 7, 8):
        print(f"    stamp {i} ({NAMES[i - 1]} done) per wave, median kcycles since start: " + " ".join(f"{rel[:, wv, i].median():.1f}" for wv in range(8)), flush=True)

# the round-4 form of the unit: K = 256 on the pre-split (X2) ``out`` + the coarse half as a pre-LayerNorm addend (stamp 3 = C tile + addend in LDS)
for n, h, w in ((14, 192, 256),):
    F_ = 256
    out = P.Feat(torch.randn(n, h, w, F_, device="cuda"), x2=True)   # (bytes of random floats read as bf16 pairs: timing only)
    pre = P.Feat(torch.randn(n, h, w, F_, device="cuda"))
    res = P.Feat(torch.randn(n, h, w, F_, device="cuda"))
    y = P.Feat.alloc(n, h, w, F_, "cuda")
    cwa = P.pack_conv(torch.randn(F_, F_, 3, 3, device="cuda") / (3 * F_ ** 0.5), torch.randn(F_, device="cuda"), pad=1, prec=PR)
    for _ in range(3):
        P.conv3x3_ln_gate(out, cwa, ln, gw, gb, y, act=P.ACT_RELU, mul=out, res=res, pre=pre, pre_cin=F_)
    torch.cuda.synchronize()
    nblk = n * (h // 8) * (w // 16)
    sw = stamps[: nblk * 80].view(nblk, 8, 10).cpu().double()
    s0 = sw[:, 0]
    d = (s0[:, 1:] - s0[:, :-1]) / 1000.0
    tot = (s0[:, 9] - s0[:, 0]) / 1000.0
    print(f"{n}x{h}x{w} 256(+pre, X2)->256->256: {nblk} workgroups, median {tot.median():.1f} kcycles per workgroup: " +
          "  ".join(f"{nm} {d[:, i].median():.2f}" for i, nm in enumerate(NAMES)), flush=True)
