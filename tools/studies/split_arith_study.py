"""CPU study (numpy): accuracy of split-operand schemes for an fp32 dot product on MFMA hardware, against float64.

  bf16x3        x = h + l (bf16 RNE), three bf16 MFMAs  h*h' + h*l' + l*h'                          (shipped; 3 bf16-rate MFMAs)
  f16+f6x2      main product fp16(x) * fp16(w) on the f16 MFMA, the two first-order corrections
                x * (w - fp16 w)  and  (x - fp16 x) * w  on the BLOCK-SCALED fp6 (e2m3, MX: one power-of-two scale per 32 k) MFMA,
                whose rate is 4 x bf16 on CDNA4: 1 + 2 * 0.25 = 1.5 bf16-rate MFMAs
  f16+f8x2      the same with e4m3 block-scaled corrections (2 x bf16 rate): 1 + 2 * 0.5 = 2
  f16           fp16(x) * fp16(w) alone: 1
  python tools/studies/split_arith_study.py
"""
import numpy as np

rng = np.random.default_rng(0)


def bf16(v):
    u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def f16(v):
    return v.astype(np.float16).astype(np.float32)


def mx_quant(v, grid, emax):
    """block-scaled quantisation along the last axis in blocks of 32: scale = 2^(floor(log2 max|v|) - emax), elements RNE onto +-grid"""
    s = v.shape
    b = v.reshape(-1, 32).astype(np.float64)
    m = np.abs(b).max(1, keepdims=True)
    e = np.floor(np.log2(np.where(m > 0, m, 1.0))) - emax
    sc = 2.0 ** e
    a = np.abs(b) / sc
    idx = np.clip(np.searchsorted(grid, a), 1, len(grid) - 1)
    lo, hi = grid[idx - 1], grid[idx]
    q = np.where(a - lo <= hi - a, lo, hi)
    q = np.where(a > grid[-1], grid[-1], q)
    return (np.sign(b) * q * sc).reshape(s).astype(np.float32)


E2M3 = np.array(sorted(set([i * 0.125 for i in range(8)] + [(1 + i / 8) * 2.0 ** e for e in range(3) for i in range(8)])))
E4M3 = np.array(sorted(set([i * 2.0 ** -9 for i in range(8)] + [(1 + i / 8) * 2.0 ** e for e in range(-6, 9) for i in range(8) if (1 + i / 8) * 2.0 ** e <= 448])))
E2M1 = np.array([0, 0.5, 1, 1.5, 2, 3, 4, 6.0])


def dots(x, w):
    """x [M, K], w [N, K] -> dict of [M, N] results (products exact, accumulation in float64: isolates the operand errors)"""
    X, W = x.astype(np.float64), w.astype(np.float64)
    ref = X @ W.T
    out = {}
    xh, wh = bf16(x), bf16(w)
    xl, wl = bf16(x - xh), bf16(w - wh)
    out["bf16x3"] = xh.astype(np.float64) @ wh.T + xh.astype(np.float64) @ wl.T + xl.astype(np.float64) @ wh.T
    out["bf16"] = xh.astype(np.float64) @ wh.T
    x1, w1 = f16(x), f16(w)
    x2, w2 = x - x1, w - w1
    main = x1.astype(np.float64) @ w1.T
    out["f16"] = main
    for name, grid, emax in (("f6", E2M3, 2), ("f8", E4M3, 8), ("f4", E2M1, 2)):
        q = lambda v: mx_quant(v, grid, emax).astype(np.float64)  # noqa: E731
        out[f"f16+{name}x2"] = main + q(x) @ q(w2).T + q(x2) @ q(w).T
    return ref, out


def main():
    for label, mkx in (("x ~ N(0,1)", lambda m, k: rng.standard_normal((m, k))),
                       ("x = relu(N(0,1)) (half zeros)", lambda m, k: np.maximum(rng.standard_normal((m, k)), 0)),
                       ("x heavy-tailed (N * lognormal)", lambda m, k: rng.standard_normal((m, k)) * np.exp(rng.standard_normal((m, k)))),
                       ("x ~ N(3, 0.1) (large mean: cancellation-free)", lambda m, k: 3 + 0.1 * rng.standard_normal((m, k)))):
        for K in (288, 4608):
            x = mkx(64, K).astype(np.float32)
            w = (rng.standard_normal((96, K)) / np.sqrt(K)).astype(np.float32)
            ref, out = dots(x, w)
            den = np.sqrt((ref ** 2).mean())
            print(f"{label}, K = {K}: rms error / rms result   " + "   ".join(f"{k} {np.sqrt(((v - ref) ** 2).mean()) / den:.2e}" for k, v in out.items()))


if __name__ == "__main__":
    main()
