"""phase stamps of conv3x3_c256_f6_kernel (a -DF6_STAMPS build: tools/probes/f6_stamps.sh): cycles per workgroup phase, median over waves"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
DEV = "cuda"
n, h, wd = 14, 192, 256
blocks = n * (h // 8) * (wd // 16)
st = torch.zeros(blocks * 8 * 8, dtype=torch.int64, device=DEV)
os.environ["PRV2_F6_STAMPS"] = hex(st.data_ptr())
os.environ["PRV2_DISPATCH"] = "ctypes"
from patchrefinerv2_amd import ops as P
g = torch.Generator(device=DEV).manual_seed(0)
w = torch.randn(256, 256, 3, 3, device=DEV, generator=g) / 48
b = torch.randn(256, device=DEV, generator=g)
x = P.Feat(torch.randn(n, h, wd, 256, device=DEV, generator=g))
cw = P.pack_conv3x3_f6(w, b)
out = P.Feat.alloc(n, h, wd, 256, DEV)
for _ in range(3):
    P.conv3x3_f6(x, cw, out, relu_in=True, res=x)
torch.cuda.synchronize()
s = st.view(-1, 8, 8).double()
s = s[s[:, 0, 2] > 0]  # workgroups that ran (persistent grid: 256)
K = s[:, :, 2]
print(f"{s.shape[0]} workgroups, tiles per workgroup {K.min().item():.0f}..{K.max().item():.0f}")
print(f"prologue (first halo + weights)  median {s[:, :, 3].median().item():9.0f} cycles")
print(f"main loop per tile               median {(s[:, :, 0] / K).median().item():9.0f} cycles = {(s[:, :, 0] / K / 36).median().item():.0f} per step (MFMA-bound: 2 waves x 48 MFMAs x 16-19 cycles = 1536-1632)")
print(f"epilogue per tile                median {(s[:, :, 1] / K).median().item():9.0f} cycles")
print(f"in-kernel clock: s_memtime / s_memrealtime (100 MHz) = {(s[:, :, 4] / s[:, :, 5]).median().item() * 0.1:.3f} GHz; workgroup lifetime {s[:, :, 5].median().item() / 100:.1f} us")
print(f"workgroup total                  median {s[:, :, 4].median().item():9.0f}  max {s[:, :, 4].max().item():9.0f} cycles")
