"""Evaluation metrics of the LINF-LP harness on the device (LINF-LP/test.py:172-225): PSNR with the benchmark / div2k
conventions, SSIM, LR-consistency PSNR through the MATLAB-style bicubic `imresize`, uint8 formatting.  The reference moves
every SR image to the host for these (numpy / cv2); here the images stay in HBM and only scalars come back.
LPIPS (a pretrained AlexNet) is outside the path."""
from math import ceil

import numpy as np
import torch


def _cubic(x):
    a = np.abs(x)
    a2, a3 = a * a, a * a * a
    return (1.5 * a3 - 2.5 * a2 + 1) * (a <= 1) + (-0.5 * a3 + 2.5 * a2 - 4 * a + 2) * ((1 < a) & (a <= 2))


def imresize_tables(in_length, out_length, scale, k_width=4.0):
    """Tap tables of MATLAB-style bicubic resizing (imresize.py:64-88): float64 weights [out,P], int32 indices [out,P];
    for scale < 1 the kernel is stretched by 1/scale (antialiasing), borders are symmetric."""
    scale = float(scale)
    kw = k_width / scale if scale < 1 else k_width
    u = np.arange(1, out_length + 1, dtype=np.float64) / scale + 0.5 * (1 - 1 / scale)
    left = np.floor(u - kw / 2)
    P = int(ceil(kw)) + 2
    ind = (left[:, None] + np.arange(P) - 1).astype(np.int32)
    t = u[:, None] - ind - 1
    w = scale * _cubic(scale * t) if scale < 1 else _cubic(t)
    w = w / w.sum(1, keepdims=True)
    aux = np.concatenate((np.arange(in_length), np.arange(in_length - 1, -1, -1))).astype(np.int32)
    ind = aux[np.mod(ind, aux.size)]
    keep = np.nonzero(np.any(w, axis=0))[0]
    return w[:, keep], ind[:, keep]


_TABLES = {}


def imresize(ops, img, scale):
    """img [B,C,H,W] on the device -> [B,C,ceil(scale*H),ceil(scale*W)], rows first then columns (imresize.py:157-171)."""
    B, C, H, W = img.shape
    oh, ow = int(ceil(scale * H)), int(ceil(scale * W))
    key = (H, W, float(scale), str(img.device))
    if key not in _TABLES:
        tabs = []
        for n_in, n_out in ((H, oh), (W, ow)):
            w, i = imresize_tables(n_in, n_out, scale)
            tabs.append((torch.from_numpy(np.ascontiguousarray(i)).to(img.device),
                         torch.from_numpy(np.ascontiguousarray(w.astype(np.float32))).to(img.device)))
        _TABLES[key] = tabs
    (ih, wh), (iw, ww) = _TABLES[key]
    mid = ops.resample_taps(img, ops.empty(B, C, oh, W), ih, wh, 0)
    return ops.resample_taps(mid, ops.empty(B, C, oh, ow), iw, ww, 1)


def psnr(ops, sr, hr, dataset=None, scale=1, rgb_range=1):
    """calc_psnr (utils.py:132-149) with the reduction on the device; returns a python float (mean over the batch like the
    reference's `.mean()` over all elements)."""
    B, C, H, W = sr.shape
    shave = scale if dataset is not None else 0
    if dataset not in (None, "benchmark", "div2k"):
        raise NotImplementedError
    luma = dataset == "benchmark" and C > 1
    s = ops.sqdiff_sum(sr, hr, shave=shave, luma=luma, rgb_range=rgb_range)
    count = B * (1 if luma else C) * (H - 2 * shave) * (W - 2 * shave)
    return float(-10.0 * torch.log10(s.sum() / count))


_WIN = {}


def ssim(ops, img1, img2):
    """calculate_ssim (utils.py:174-193) for [0,1] images [B,3,H,W]: per-image mean over channels of the mean SSIM map."""
    dev = str(img1.device)
    if dev not in _WIN:
        g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
        g = g / g.sum()
        _WIN[dev] = torch.from_numpy(np.outer(g, g).reshape(-1)).to(img1.device)
    B, C, H, W = img1.shape
    s = ops.ssim_sum(img1, img2, _WIN[dev], 255.0)
    return (s / float((H - 10) * (W - 10))).mean(dim=1)          # [B]


def lr_consistency_psnr(ops, pred01, inp01, scale, dataset=None):
    """LR-consistency (test.py:183-187,197-200): `psnr_fn(imresize(pred, 1/scale), batch['inp'])` -- `psnr_fn` is the SAME
    partial the HR PSNR uses (test.py:66-75), so with an eval_type it shaves `scale` border pixels of the LR image and, for
    'benchmark', compares luma."""
    return psnr(ops, imresize(ops, pred01, 1.0 / scale), inp01, dataset=dataset, scale=scale if dataset is not None else 1)
