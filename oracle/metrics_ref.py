"""ORACLE (test infrastructure only -- never imported by the product): numpy restatement of the reference's evaluation
metrics right after the LINF-LP hot path.

  imresize / contributions / cubic : LINF-LP/imresize.py:54-174 (MATLAB-style antialiased bicubic; MIT-licensed
      fatheral/matlab_imresize algorithm) -- PINNED: checked against the genuine module on seeded inputs
      (tests/golden/make_golden_metrics.py -> metrics.npz, MANIFEST.json "metrics").
  calc_psnr                        : LINF-LP/utils.py:132-149 -- PINNED the same way.
  ssim / calculate_ssim            : LINF-LP/utils.py:152-193.  The reference evaluates it with cv2 (getGaussianKernel,
      filter2D), which this image does not have, so it cannot be run here: PARITY UNPINNED for SSIM; the restatement follows
      the published formula (11x11 Gaussian sigma 1.5, 'valid' region, C1/C2 of the [0,255] range) and is held to
      known-answer properties in the tests.
  skimage_ssim                     : SRFlow-LP/code/Measure.py:45-48 calls skimage.metrics.structural_similarity(imgA, imgB, full=True,
      multichannel=True) on uint8 images -- a third-party dependency (scikit-image, unpinned in the reference's requirements; the
      algorithm below is the one of scikit-image 0.16-0.19, `skimage/metrics/_structural_similarity.py`) that is NOT in this image:
      PARITY UNPINNED; restated from the published algorithm and held to known-answer properties in the tests.
"""
from math import ceil

import numpy as np


def cubic(x):
    """imresize.py:54-61."""
    x = np.asarray(x, dtype=np.float64)
    a = np.abs(x)
    a2, a3 = a * a, a * a * a
    return (1.5 * a3 - 2.5 * a2 + 1) * (a <= 1) + (-0.5 * a3 + 2.5 * a2 - 4 * a + 2) * ((1 < a) & (a <= 2))


def contributions(in_length, out_length, scale, k_width=4.0):
    """imresize.py:64-88 (kernel = cubic): -> weights [out, P'] float64, indices [out, P'] int32 (symmetric boundary)."""
    if scale < 1:
        h = lambda t: scale * cubic(scale * t)
        kernel_width = 1.0 * k_width / scale
    else:
        h = cubic
        kernel_width = k_width
    x = np.arange(1, out_length + 1).astype(np.float64)
    u = x / scale + 0.5 * (1 - 1 / scale)
    left = np.floor(u - kernel_width / 2)
    P = int(ceil(kernel_width)) + 2
    ind = np.expand_dims(left, 1) + np.arange(P) - 1
    indices = ind.astype(np.int32)
    weights = h(np.expand_dims(u, 1) - indices - 1)
    weights = weights / np.expand_dims(weights.sum(1), 1)
    aux = np.concatenate((np.arange(in_length), np.arange(in_length - 1, -1, -1))).astype(np.int32)
    indices = aux[np.mod(indices, aux.size)]
    keep = np.nonzero(np.any(weights, axis=0))[0]
    return weights[:, keep], indices[:, keep]


def imresize(img, scalar_scale):
    """imresize.py:136-174 for an HxWxC float image and one scalar scale (both dims equal => rows first, :157-171)."""
    H, W = img.shape[:2]
    oh, ow = int(ceil(scalar_scale * H)), int(ceil(scalar_scale * W))
    wh, ih = contributions(H, oh, float(scalar_scale))
    ww, iw = contributions(W, ow, float(scalar_scale))
    B = img.astype(np.float64)
    B = np.sum(wh[:, :, None, None] * B[ih], axis=1)                      # rows
    B = np.sum(ww[None, :, :, None] * B[:, iw], axis=2)                   # cols
    return B


def calc_psnr(sr, hr, dataset=None, scale=1, rgb_range=1):
    """utils.py:132-149 on numpy arrays [B,C,H,W]."""
    diff = (sr - hr) / rgb_range
    if dataset is not None:
        shave = scale
        if dataset == "benchmark" and diff.shape[1] > 1:
            diff = (diff * (np.array([65.738, 129.057, 25.064], dtype=diff.dtype).reshape(1, 3, 1, 1) / 256)).sum(1)
        elif dataset not in ("benchmark", "div2k"):
            raise NotImplementedError
        diff = diff[..., shave:-shave, shave:-shave]
    return -10 * np.log10(np.mean(diff ** 2))


def gaussian_window():
    """cv2.getGaussianKernel(11, 1.5) outer product (utils.py:158-159): exp(-(i-5)^2 / (2*1.5^2)) normalised."""
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    g = g / g.sum()
    return np.outer(g, g)


def ssim(img1, img2):
    """utils.py:152-171 for one [0,255] plane (valid region only, so the filter2D border mode is irrelevant)."""
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a, b = img1.astype(np.float64), img2.astype(np.float64)
    win = gaussian_window()
    H, W = a.shape

    def filt(t):
        out = np.zeros((H - 10, W - 10))
        for dy in range(11):
            for dx in range(11):
                out += win[dy, dx] * t[dy:dy + H - 10, dx:dx + W - 10]
        return out
    mu1, mu2 = filt(a), filt(b)
    s1, s2, s12 = filt(a * a) - mu1 ** 2, filt(b * b) - mu2 ** 2, filt(a * b) - mu1 * mu2
    return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s1 + s2 + C2))).mean()


def calculate_ssim(img1, img2):
    """utils.py:174-193 for HxWx3 [0,255] images: mean over the 3 channels."""
    return float(np.mean([ssim(img1[:, :, i], img2[:, :, i]) for i in range(img1.shape[2])]))


def skimage_ssim(imgA, imgB):
    """skimage.metrics.structural_similarity(imgA, imgB, multichannel=True) for uint8 HxWxC images with its defaults (Measure.py:45-48):
    win_size 7, uniform filter, use_sample_covariance (cov_norm = 49/48), K1 = 0.01, K2 = 0.03, data_range 255 (the uint8 dtype range),
    images cast to float64 unscaled; per channel the SSIM map is averaged over the image cropped by (win_size - 1) // 2 = 3 pixels, then
    the channel means are averaged."""
    from scipy.ndimage import uniform_filter
    a, b = np.asarray(imgA), np.asarray(imgB)
    assert a.shape == b.shape and a.dtype == np.uint8 and b.dtype == np.uint8 and a.ndim == 3 and min(a.shape[:2]) >= 7
    C1, C2, cov_norm = (0.01 * 255) ** 2, (0.03 * 255) ** 2, 49.0 / 48.0
    vals = []
    for c in range(a.shape[2]):
        x, y = a[:, :, c].astype(np.float64), b[:, :, c].astype(np.float64)
        ux, uy = uniform_filter(x, size=7), uniform_filter(y, size=7)
        uxx, uyy, uxy = uniform_filter(x * x, size=7), uniform_filter(y * y, size=7), uniform_filter(x * y, size=7)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
        vals.append(S[3:-3, 3:-3].mean())
    return float(np.mean(vals))
