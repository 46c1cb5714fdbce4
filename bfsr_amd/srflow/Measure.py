"""Counterpart of the reference's `Measure` (SRFlow-LP/code/Measure.py:31-53) for the metrics that can run without pretrained
networks, computed on the device from uint8 HWC images: `psnr` (skimage's `peak_signal_noise_ratio` = 10 log10(255^2 / mse) for uint8
inputs) and `ssim` (skimage's 7x7 uniform-window `structural_similarity`; the restatement it is tested against is unpinned -- scikit-image
is not in this image, oracle/metrics_ref.py says so).  LPIPS needs the pretrained AlexNet weights (a download): `measure` keeps the
reference's three-element return value with NaN in the LPIPS slot (INTEGRATION.md lists the deviation); `lpips` itself raises."""
import warnings

import numpy as np
import torch

_warned_lpips = False


class Measure(object):
    def __init__(self, ops, net='alex', use_gpu=True):
        self.ops = ops

    def psnr(self, imgA, imgB):
        """imgA, imgB: uint8 HWC numpy arrays (or uint8 tensors [H,W,C]) -> float, Measure.py:50-52."""
        ops = self.ops
        f = lambda im: ops.to_device(torch.as_tensor(np.asarray(im)).permute(2, 0, 1).unsqueeze(0).to(torch.float32).contiguous())
        a, b = f(imgA), f(imgB)
        s = ops.sqdiff_sum(a, b, shave=0, luma=False, rgb_range=1.0)          # float64 sum of squared differences
        mse = float(s.sum()) / a.numel()
        if mse == 0.0:                      # identical images: skimage's peak_signal_noise_ratio returns inf (divide warning), not an error
            return float("inf")
        return float(10.0 * np.log10(255.0 ** 2 / mse))

    def ssim(self, imgA, imgB):
        """imgA, imgB: uint8 HWC arrays -> float, Measure.py:45-48: skimage.metrics.structural_similarity(imgA, imgB, full=True,
        multichannel=True) for uint8 inputs = uniform 7x7 filter, sample covariance (x 49/48), K1 = 0.01, K2 = 0.03, data_range 255, the SSIM
        map averaged over the image cropped by 3 pixels, then over the channels.  On the device: bfsr_ssim_sum_w (metrics.hip)."""
        ops = self.ops
        f = lambda im: ops.to_device(torch.as_tensor(np.asarray(im)).permute(2, 0, 1).unsqueeze(0).to(torch.float32).contiguous())
        a, b = f(imgA), f(imgB)
        if a.shape != b.shape or a.shape[2] < 7 or a.shape[3] < 7:
            raise ValueError("Measure.ssim: images must have the same shape and be at least 7x7 (skimage raises for smaller images)")
        win = torch.full((49,), 1.0 / 49.0, dtype=torch.float64, device=a.device)
        s = ops.ssim_sum_w(a, b, win, cov_norm=49.0 / 48.0, scale=1.0)                 # float64 [1, C] sums over the (H-6) x (W-6) region
        return float(s.sum()) / (a.shape[1] * (a.shape[2] - 6) * (a.shape[3] - 6))

    def lpips(self, imgA, imgB, model=None):
        raise NotImplementedError("Measure.lpips needs the pretrained AlexNet LPIPS weights (lpips package): outside the path, not in this image")

    def measure(self, imgA, imgB, with_lpips=True):
        """Measure.py:37-38: [psnr, ssim, lpips] -- the reference's return value; the LPIPS entry is NaN (no pretrained network offline),
        so drop-in callers that unpack three values keep working.  with_lpips=False returns [psnr, ssim]."""
        out = [float(self.psnr(imgA, imgB)), float(self.ssim(imgA, imgB))]
        if with_lpips:
            global _warned_lpips
            if not _warned_lpips:           # once per process: a NaN column in the caller's CSV must not pass for a computed metric
                _warned_lpips = True
                warnings.warn("bfsr_amd Measure.measure: LPIPS is NOT computed (pretrained AlexNet weights are a download) -- the third value is NaN; "
                              "pass with_lpips=False for [psnr, ssim]", RuntimeWarning, stacklevel=2)
            out.append(float("nan"))
        return out
