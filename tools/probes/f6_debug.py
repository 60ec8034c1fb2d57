import sys, os
sys.path.insert(0, os.getcwd())
import torch
from patchrefinerv2_amd import ops as P
DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)
w = torch.randn(256, 256, 3, 3, device=DEV, generator=g) / 48
b = torch.randn(256, device=DEV, generator=g)
cw = P.pack_conv3x3_f6(w, b)
torch.cuda.synchronize(); print("packed", flush=True)
for n, h, wd in [(1, 24, 32), (2, 24, 32), (1, 48, 64), (4, 48, 64), (1, 96, 128), (2, 96, 128), (1, 192, 256), (4, 192, 256), (14, 192, 256)]:
    x = P.Feat(torch.randn(n, h, wd, 256, device=DEV, generator=g))
    res = P.Feat(torch.randn(n, h, wd, 256, device=DEV, generator=g))
    out = P.Feat.alloc(n, h, wd, 256, DEV)
    for rep in range(3):
        P.conv3x3_f6(x, cw, out, relu_in=True, res=res)
        torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(torch.relu(x.buf[:1]).permute(0, 3, 1, 2), w, b, padding=1).permute(0, 2, 3, 1) + res.buf[:1]
    print(n, h, wd, "ok", float((out.buf[:1] - ref).norm() / ref.norm()), flush=True)
