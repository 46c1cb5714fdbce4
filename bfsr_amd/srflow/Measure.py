"""Counterpart of the reference's `Measure` (SRFlow-LP/code/Measure.py:31-53) for the metrics that can run without pretrained
networks: PSNR on uint8 HWC images, computed on the device.  `Measure.psnr` there is skimage's
`peak_signal_noise_ratio(imgA, imgB)` = 10 log10(255^2 / mse) for uint8 inputs.  SSIM (skimage's 7x7 uniform-window variant)
and LPIPS (pretrained AlexNet) are outside the accelerated path."""
import numpy as np
import torch


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

    def measure(self, imgA, imgB):
        raise NotImplementedError("Measure.measure also needs SSIM (skimage) and LPIPS (pretrained AlexNet): outside the path")
