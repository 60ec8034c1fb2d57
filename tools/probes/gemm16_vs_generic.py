import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from patchrefinerv2_amd import ops as P, lib as L
pr = L.PREC_NAMES[sys.argv[1] if len(sys.argv) > 1 else 'bf16x3']
torch.manual_seed(0)
for (m, cin, cout) in ((1024, 512, 128), (1024, 512, 128), (256, 512, 128) , (512, 512, 128), (1024, 256, 128), (1024, 384, 128), (1024, 448, 128), (1024, 544, 128), (1024, 1024, 128)):
    x = torch.randn(1, cin, m // 8, 8, device='cuda')
    w = torch.randn(cout, cin, 1, 1, device='cuda') / cin ** 0.5
    cw = P.pack_conv(w, None, prec=pr)
    xf = P.Feat.from_nchw(x)
    a = P.conv2d(xf, cw).to_nchw().reshape(cout, m)
    g = P.conv2d(xf, cw, force_generic=True).to_nchw().reshape(cout, m)
    d = (a - g).abs()
    bad = ~(d <= (3e-2 if len(sys.argv) > 1 and sys.argv[1] == 'bf16' else 1e-3))
    rows = bad.any(0).nonzero().flatten().tolist()
    cols = bad.any(1).nonzero().flatten().tolist()
    print(m, cin, cout, 'max diff', float(d.max()), 'nbad', int(bad.sum()), 'rows', rows[:6], '..', rows[-3:], len(rows), 'cols', cols[:20], len(cols))
