import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from patchrefinerv2_amd import ops as P, lib as L
pr = L.PREC_NAMES['bf16x3']
for (n, h, w, cin, cout) in ((1, 128, 128, 256, 256), (1, 256, 256, 256, 256), (4, 256, 256, 256, 256), (14, 196, 259, 256, 256), (27, 448, 448, 256, 256), (14, 196, 259, 256, 128), (14,196,259,512,256), (14, 196, 259, 1024, 256)):
    x = P.Feat.alloc(n, h, w, cin, 'cuda'); x.buf.normal_()
    cw = P.pack_conv(torch.randn(cout, cin, 1, 1, device='cuda') / cin ** 0.5, None, prec=pr)
    y = P.Feat.alloc(n, h, w, cout, 'cuda')
    for force in (False, True):
        for _ in range(3): P.conv2d(x, cw, y, force_generic=force)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): P.conv2d(x, cw, y, force_generic=force)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        m = n * h * w
        print(f"M={m:8d} K={cin} N={cout} {'generic' if force else 'gemm16 '} {ms:7.3f} ms {2.0*m*cin*cout/ms/1e9:7.1f} TF  {(m*cin*4+m*cout*4)/ms/1e9:6.2f} TB/s")
