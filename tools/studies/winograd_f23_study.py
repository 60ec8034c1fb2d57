"""CPU study (numpy, float64 reference): Winograd F(2x2, 3x3) for the 256-channel 3x3 convs in the split-operand arithmetics -- how much accuracy the
transforms cost when the sixteen transform-domain GEMMs take bf16x3 / fp16 + fp6 operands (tools/studies/split_arith_study.py's quantisers), fp32
input / output transforms, accumulation exact (isolates the operand errors, as in that study).

    Y = A^T [ (G g G^T) . (B^T d B) ] A          4 instead of 9 multiplications per output and channel pair (2.25 x fewer MFMA operations)

DESIGN.md section 9: the next algebraic lever once cheaper MACs stopped buying time (profiles/r05_experiments.txt #19).   python tools/studies/winograd_f23_study.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from split_arith_study import E2M3, bf16, f16, mx_quant  # noqa: E402

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def gemm(x, w, mode):
    """x [M, K], w [N, K] float32 operands -> [M, N] in the named arithmetic (products and sums exact)"""
    if mode == "exact":
        return x.astype(np.float64) @ w.astype(np.float64).T
    if mode == "bf16x3":
        xh, wh = bf16(x), bf16(w)
        xl, wl = bf16(x - xh), bf16(w - wh)
        return xh.astype(np.float64) @ wh.T + xh.astype(np.float64) @ wl.T + xl.astype(np.float64) @ wh.T
    x1, w1 = f16(x), f16(w)
    q = lambda v: mx_quant(v, E2M3, 2).astype(np.float64)  # noqa: E731
    return x1.astype(np.float64) @ w1.T + q(x) @ q(w - w1).T + q(x - x1) @ q(w).T


def direct(x, w, mode):  # x [H, W, C], w [Co, C, 3, 3] -> [H - 2, W - 2, Co] (valid)
    H, W, C = x.shape
    y = 0.0
    for ky in range(3):
        for kx in range(3):
            y = y + gemm(x[ky:H - 2 + ky, kx:W - 2 + kx].reshape(-1, C), w[:, :, ky, kx], mode).reshape(H - 2, W - 2, -1)
    return y


def winograd(x, w, mode):
    H, W, C = x.shape
    th, tw = (H - 2) // 2, (W - 2) // 2
    U = np.einsum("ij,ocjk,lk->iloc", G, w.astype(np.float64), G)                                        # [4, 4, Co, C]   (host side, float64, once per layer)
    d = np.stack([np.stack([x[2 * i:2 * i + 4, 2 * j:2 * j + 4] for j in range(tw)]) for i in range(th)])  # [th, tw, 4, 4, C]
    V = np.einsum("ij,abjkc,lk->abilc", BT, d.astype(np.float64), BT).astype(np.float32)                  # fp32 input transform (adds only: exact up to one rounding)
    M = np.empty((th, tw, 4, 4, w.shape[0]))
    for i in range(4):
        for l in range(4):
            M[:, :, i, l] = gemm(V[:, :, i, l].reshape(-1, C), U[i, l].astype(np.float32), mode).reshape(th, tw, -1)
    Y = np.einsum("ij,abjko,lk->abilo", AT, M, AT)                                                        # [th, tw, 2, 2, Co]
    return Y.transpose(0, 2, 1, 3, 4).reshape(2 * th, 2 * tw, -1)


def main():
    rng = np.random.default_rng(0)
    C, Co, H, W = 256, 64, 18, 34
    for label, x in (("x = relu(N(0,1))", np.maximum(rng.standard_normal((H, W, C)), 0)), ("x ~ N(0,1)", rng.standard_normal((H, W, C))),
                     ("x smooth (neighbouring pixels correlated 0.9)", None)):
        if x is None:
            z = rng.standard_normal((H, W, C))
            for _ in range(6):
                z = 0.5 * z + 0.125 * (np.roll(z, 1, 0) + np.roll(z, -1, 0) + np.roll(z, 1, 1) + np.roll(z, -1, 1))
            x = z / z.std()
        x = x.astype(np.float32)
        w = (rng.standard_normal((Co, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
        ref = direct(x, w, "exact")
        den = np.sqrt((ref ** 2).mean())
        assert np.abs(winograd(x, w, "exact") - ref).max() < 1e-5 * den                                   # the identity itself
        line = f"{label}, C = {C}: rms error / rms result"
        for mode in ("bf16x3", "f16f6"):
            ed = np.sqrt(((direct(x, w, mode) - ref) ** 2).mean()) / den
            ew = np.sqrt(((winograd(x, w, mode) - ref) ** 2).mean()) / den
            line += f"   {mode}: direct {ed:.2e}, Winograd F(2x2,3x3) {ew:.2e} ({ew / ed:.1f}x)"
        print(line)


if __name__ == "__main__":
    main()
