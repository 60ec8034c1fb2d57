"""Depth-evaluation metrics and colour maps of the reference's Tester (host side, numpy).

compute_errors      estimator/utils/metric.py:11-50   (delta1-3, AbsRel, RMSE, log10, RMSE-log, SILog, SqRel)
soft_edge_error     estimator/utils/metric.py:53-72   (min |gt shifted - pred| over a (2r+1)^2 window)
get_boundaries      estimator/utils/metric.py:74-85   (disparity jumps > th; dilation needs cv2 -> dilation=0 only)
compute_metrics     estimator/utils/metric.py:87-149  (resize, clamp, valid / crop masks, optional SEE on gt edges)
colorize            estimator/utils/color.py:95-158   (percentile normalisation + matplotlib colour map, RGBA uint8)

Pinned by tests/golden/output_stage.npz (the reference functions imported by oracle/make_golden.py).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def compute_errors(gt: np.ndarray, pred: np.ndarray) -> dict:
    thresh = np.maximum(gt / pred, pred / gt)
    d = gt - pred
    err = np.log(pred) - np.log(gt)
    return dict(a1=(thresh < 1.25).mean(), a2=(thresh < 1.25 ** 2).mean(), a3=(thresh < 1.25 ** 3).mean(),
                abs_rel=np.mean(np.abs(d) / gt), rmse=np.sqrt((d ** 2).mean()),
                log_10=np.abs(np.log10(gt) - np.log10(pred)).mean(),
                rmse_log=np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean()),
                silog=np.sqrt(np.mean(err ** 2) - np.mean(err) ** 2) * 100, sq_rel=np.mean(d ** 2 / gt))


def _shift(data: np.ndarray, dx: int, dy: int, fill=0) -> np.ndarray:
    out = np.roll(data, dx, axis=1)
    if dx < 0:
        out[:, dx:] = fill
    elif dx > 0:
        out[:, :dx] = fill
    out = np.roll(out, dy, axis=0)
    if dy < 0:
        out[dy:, :] = fill
    elif dy > 0:
        out[:dy, :] = fill
    return out


def soft_edge_error(pred: np.ndarray, gt: np.ndarray, radius: int = 1) -> np.ndarray:
    diffs = [np.abs(_shift(gt, i, j, 0) - pred) for i in range(-radius, radius + 1) for j in range(-radius, radius + 1)]
    return np.minimum.reduce(diffs)


def get_boundaries(disp: np.ndarray, th: float = 1.0, dilation: int = 0) -> np.ndarray:
    if dilation > 0:
        raise NotImplementedError("get_boundaries(dilation > 0) uses cv2.dilate (un-vendored); every dataset branch of the "
                                  "reference calls it with dilation=0 (general_dataset.py:88-143)")
    dy = np.abs(disp[1:, :] - disp[:-1, :]) > th
    dx = np.abs(disp[:, 1:] - disp[:, :-1]) > th
    ey = np.logical_or(np.pad(dy, ((1, 0), (0, 0))), np.pad(dy, ((0, 1), (0, 0))))
    ex = np.logical_or(np.pad(dx, ((0, 0), (1, 0))), np.pad(dx, ((0, 0), (0, 1))))
    return np.logical_or(ey, ex).astype(np.float32)


def compute_metrics(gt, pred, interpolate=True, garg_crop=False, eigen_crop=True, dataset="nyu", min_depth_eval=0.1,
                    max_depth_eval=10, disp_gt_edges=None, additional_mask=None) -> dict:
    if gt.shape[-2:] != pred.shape[-2:] and interpolate:
        pred = F.interpolate(pred, gt.shape[-2:], mode="bilinear", align_corners=False).squeeze()
    pred = pred.squeeze().cpu().numpy().copy()
    pred[pred < min_depth_eval] = min_depth_eval
    pred[pred > max_depth_eval] = max_depth_eval
    pred[np.isinf(pred)] = max_depth_eval
    pred[np.isnan(pred)] = min_depth_eval
    gt_depth = gt.squeeze().cpu().numpy()
    valid = np.logical_and(gt_depth > min_depth_eval, gt_depth < max_depth_eval)
    eval_mask = np.ones(valid.shape)
    if garg_crop or eigen_crop:
        h, w = gt_depth.shape
        eval_mask = np.zeros(valid.shape)
        if garg_crop:
            eval_mask[int(0.40810811 * h):int(0.99189189 * h), int(0.03594771 * w):int(0.96405229 * w)] = 1
        elif dataset == "kitti":
            eval_mask[int(0.3324324 * h):int(0.91351351 * h), int(0.0359477 * w):int(0.96405229 * w)] = 1
        else:
            eval_mask[45:471, 41:601] = 1
    valid = np.logical_and(valid, eval_mask)
    if additional_mask is not None:
        valid = np.logical_and(valid, additional_mask.squeeze().detach().cpu().numpy())
    metrics = compute_errors(gt_depth[valid], pred[valid])
    if disp_gt_edges is not None:
        edges = disp_gt_edges.squeeze().numpy() if isinstance(disp_gt_edges, torch.Tensor) else np.squeeze(disp_gt_edges)
        mask = np.logical_and(valid.squeeze(), edges)
        see = torch.tensor([0])
        if mask.sum() > 0:
            see = soft_edge_error(pred, gt_depth)[mask].mean()
        metrics["see"] = see
    return metrics


@torch.no_grad()
def compute_metrics_device(gt: torch.Tensor, pred: torch.Tensor, interpolate=True, garg_crop=False, eigen_crop=True, dataset="nyu",
                           min_depth_eval=0.1, max_depth_eval=10, disp_gt_edges=None, additional_mask=None) -> dict:
    """``compute_metrics`` (estimator/utils/metric.py:87-149) on the tensors' device -- the prediction does not have to leave the
    GPU to be scored: same clamping, masks, error formulas and soft-edge error; the reductions accumulate in
    float64 (the numpy version sums float32 pairwise: agreement to ~1e-6 relative, tests/test_host_logic.py)."""
    dev = pred.device
    gt = gt.to(dev)
    if gt.shape[-2:] != pred.shape[-2:] and interpolate:
        pred = F.interpolate(pred, gt.shape[-2:], mode="bilinear", align_corners=False)
    p = pred.squeeze().float().clone()
    p = torch.where(torch.isnan(p), torch.full_like(p, min_depth_eval), p)  # (order as the reference: < min, > max, inf, nan)
    p = p.clamp(min_depth_eval, max_depth_eval)
    g = gt.squeeze().float()
    valid = (g > min_depth_eval) & (g < max_depth_eval)
    if garg_crop or eigen_crop:
        h, w = g.shape
        m = torch.zeros_like(valid)
        if garg_crop:
            m[int(0.40810811 * h):int(0.99189189 * h), int(0.03594771 * w):int(0.96405229 * w)] = True
        elif dataset == "kitti":
            m[int(0.3324324 * h):int(0.91351351 * h), int(0.0359477 * w):int(0.96405229 * w)] = True
        else:
            m[45:471, 41:601] = True
        valid &= m
    if additional_mask is not None:
        valid &= additional_mask.squeeze().to(dev).bool()
    gv, pv = g[valid].double(), p[valid].double()
    thresh = torch.maximum(gv / pv, pv / gv)
    d = gv - pv
    err = torch.log(pv) - torch.log(gv)
    out = dict(a1=(thresh < 1.25).double().mean(), a2=(thresh < 1.25 ** 2).double().mean(), a3=(thresh < 1.25 ** 3).double().mean(),
               abs_rel=(d.abs() / gv).mean(), rmse=(d ** 2).mean().sqrt(), log_10=(torch.log10(gv) - torch.log10(pv)).abs().mean(),
               rmse_log=(err ** 2).mean().sqrt(), silog=((err ** 2).mean() - err.mean() ** 2).sqrt() * 100, sq_rel=(d ** 2 / gv).mean())
    if disp_gt_edges is not None:
        edges = torch.as_tensor(disp_gt_edges).squeeze().to(dev) != 0
        mask = valid & edges
        see = torch.zeros((), device=dev, dtype=torch.float64)
        if bool(mask.any()):
            best = None
            for i in (-1, 0, 1):      # soft_edge_error(radius=1): min over the 3 x 3 shifts of gt (zero-filled borders)
                for j in (-1, 0, 1):
                    sh = torch.zeros_like(g)
                    ys, yd = (slice(0, g.shape[0] - j), slice(j, None)) if j >= 0 else (slice(-j, None), slice(0, g.shape[0] + j))
                    xs, xd = (slice(0, g.shape[1] - i), slice(i, None)) if i >= 0 else (slice(-i, None), slice(0, g.shape[1] + i))
                    sh[yd, xd] = g[ys, xs]
                    diff = (sh - p).abs()
                    best = diff if best is None else torch.minimum(best, diff)
            see = best[mask].double().mean()
        out["see"] = see
    keys = list(out)
    vals = torch.stack([out[k].double() for k in keys]).cpu().tolist()  # one D2H of ten scalars
    return dict(zip(keys, vals))


def evaluate(per_frame: list) -> dict:
    """mean of every metric over the frames (general_dataset.py evaluate / mmengine-style collect)."""
    keys = per_frame[0].keys()
    return {k: float(np.mean([float(m[k]) for m in per_frame])) for k in keys}


def colorize(value, vmin=None, vmax=None, cmap="turbo_r", invalid_val=-99, invalid_mask=None,
             background_color=(128, 128, 128, 255), gamma_corrected=False, value_transform=None, vminp=2, vmaxp=95):
    """[H,W] / [1,1,H,W] depth -> RGBA uint8 [H,W,4]."""
    import matplotlib
    if isinstance(value, torch.Tensor):
        value = value.detach().cpu().numpy()
    value = value.squeeze()
    if invalid_mask is None:
        invalid_mask = value == invalid_val
    mask = np.logical_not(invalid_mask)
    vmin = np.percentile(value[mask], vminp) if vmin is None else vmin
    vmax = np.percentile(value[mask], vmaxp) if vmax is None else vmax
    value = (value - vmin) / (vmax - vmin) if vmin != vmax else value * 0.0
    value[invalid_mask] = np.nan
    if value_transform:
        value = value_transform(value)
    img = matplotlib.colormaps[cmap](value, bytes=True)
    img[invalid_mask] = background_color
    if gamma_corrected:
        img = (np.power(img / 255, 2.2) * 255).astype(np.uint8)
    return img


# ------------------------------------------------------------------------------------------------------------------
# <name>_edge.png of the tester's output stage (estimator/tester/tester.py:99-106): Canny edges of the LOG depth
# (extract_edges(result, use_canny=True, preprocess='log'), estimator/utils/metric.py:169-207), widened by one pixel
# (kornia.filters.gaussian_blur2d(edges, (3, 3), ...) > 0 == a 3 x 3 binary dilation).
# skimage.feature.canny and kornia are not vendored in the reference nor installed here: the detector below restates
# skimage's published algorithm on scipy.ndimage (what skimage itself is built on) -- PARITY UNPINNED.
# ------------------------------------------------------------------------------------------------------------------
def canny(image: np.ndarray, sigma: float = 1.0, low_threshold: float = 0.1, high_threshold: float = 0.2) -> np.ndarray:
    """skimage.feature.canny(image, sigma) with its defaults (mode='constant', cval=0, thresholds 0.1 / 0.2, no mask):
    Gaussian smoothing corrected for the zero border ("bleed-over"), Sobel gradients, non-maximum suppression with
    bilinear interpolation along the gradient in four 45-degree sectors, hysteresis = 8-connected components of the
    low-threshold maxima that contain a high-threshold pixel."""
    from scipy import ndimage as ndi
    image = np.asarray(image, dtype=np.float32)
    ones = np.ones(image.shape, dtype=np.float32)
    bleed = ndi.gaussian_filter(ones, sigma, mode="constant", cval=0.0) + np.finfo(np.float32).eps
    smoothed = ndi.gaussian_filter(image, sigma, mode="constant", cval=0.0) / bleed
    eroded = np.ones(image.shape, dtype=bool)
    eroded[:1, :] = eroded[-1:, :] = False
    eroded[:, :1] = eroded[:, -1:] = False
    jsobel = ndi.sobel(smoothed, axis=1)
    isobel = ndi.sobel(smoothed, axis=0)
    magnitude = np.sqrt(isobel * isobel + jsobel * jsobel)
    ai, aj = np.abs(isobel), np.abs(jsobel)
    eroded = eroded & (magnitude >= low_threshold)
    local_max = np.zeros(image.shape, dtype=bool)

    def sector(pts, a_slice, b_slice, pa, pb, c_slice, d_slice, pc, pd, w_num, w_den):
        pts = eroded & pts
        m = magnitude[pts]
        w = w_num[pts] / w_den[pts]
        c1, c2 = magnitude[a_slice][pts[pa]], magnitude[b_slice][pts[pb]]
        plus = c2 * w + c1 * (1 - w) <= m
        c1, c2 = magnitude[c_slice][pts[pc]], magnitude[d_slice][pts[pd]]
        minus = c2 * w + c1 * (1 - w) <= m
        local_max[pts] = plus & minus

    s = np.s_
    same = ((isobel >= 0) & (jsobel >= 0)) | ((isobel <= 0) & (jsobel <= 0))
    opp = ((isobel <= 0) & (jsobel >= 0)) | ((isobel >= 0) & (jsobel <= 0))
    # 0 - 45 degrees: neighbours (i+1, j) / (i+1, j+1) and (i-1, j) / (i-1, j-1)
    sector(same & (ai >= aj), s[1:, :], s[1:, 1:], s[:-1, :], s[:-1, :-1], s[:-1, :], s[:-1, :-1], s[1:, :], s[1:, 1:], aj, ai)
    # 45 - 90: (i, j+1) / (i+1, j+1) and (i, j-1) / (i-1, j-1)
    sector(same & (ai <= aj), s[:, 1:], s[1:, 1:], s[:, :-1], s[:-1, :-1], s[:, :-1], s[:-1, :-1], s[:, 1:], s[1:, 1:], ai, aj)
    # 90 - 135: (i, j+1) / (i-1, j+1) and (i, j-1) / (i+1, j-1)
    sector(opp & (ai <= aj), s[:, 1:], s[:-1, 1:], s[:, :-1], s[1:, :-1], s[:, :-1], s[1:, :-1], s[:, 1:], s[:-1, 1:], ai, aj)
    # 135 - 180: (i-1, j) / (i-1, j+1) and (i+1, j) / (i+1, j-1)
    sector(opp & (ai >= aj), s[:-1, :], s[:-1, 1:], s[1:, :], s[1:, :-1], s[1:, :], s[1:, :-1], s[:-1, :], s[:-1, 1:], aj, ai)
    low_mask = local_max & (magnitude >= low_threshold)
    labels, count = ndi.label(low_mask, np.ones((3, 3), bool))
    if count == 0:
        return low_mask
    high_mask = low_mask & (magnitude >= high_threshold)
    good = np.zeros((count + 1,), bool)
    good[np.unique(labels[high_mask])] = True
    good[0] = False
    return good[labels]


def depth_edges(depth) -> np.ndarray:
    """the boolean map behind <name>_edge.png: canny(log depth) dilated 3 x 3 (tester.py:99-105)"""
    from scipy import ndimage as ndi
    d = torch.as_tensor(depth).detach().cpu().float().squeeze()
    d = (d > 0) * d.clamp(min=torch.finfo(torch.float32).eps).log()  # to_log, metric.py:157-161
    return ndi.binary_dilation(canny(d.numpy(), sigma=1.0), structure=np.ones((3, 3), bool))
