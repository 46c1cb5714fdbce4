"""Evaluation metrics after the hot path: oracle vs the genuine reference's goldens (imresize,
calc_psnr), the product's device metrics (on the CPU double here, on the HIP kernels under -m gpu) vs the same goldens, and
SSIM pinned to the reference's own `utils.calculate_ssim` / `eval_psnr(detail=True)` (tests/golden/linf_detail.npz, generated
with numpy/scipy restatements of the two OpenCV calls the reference makes, see tests/golden/ref_import.py), and the SRFlow
harness' `Measure.psnr` (tests/golden/srflow_measure.npz)."""
import os

import numpy as np
import pytest
import torch

import oracle.metrics_ref as O
from bfsr_amd.linf import metrics
from cpu_ops import CpuOps

T = torch.from_numpy


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "metrics.npz"))


def test_oracle_vs_reference_golden(golden_dir):
    g = _g(golden_dir)
    for tag in "abc":
        assert np.abs(O.imresize(g["img_" + tag], 1.0 / int(g["scale_" + tag])) - g["lr_" + tag]).max() <= 1e-12
    sr, hr = g["psnr_sr"], g["psnr_hr"]
    assert abs(O.calc_psnr(sr, hr) - float(g["psnr_plain"])) <= 1e-5
    assert abs(O.calc_psnr(sr, hr, dataset="div2k", scale=4) - float(g["psnr_div2k4"])) <= 1e-5
    assert abs(O.calc_psnr(sr, hr, dataset="benchmark", scale=3) - float(g["psnr_bench3"])) <= 1e-5


def _check_product(ops, g):
    dev = ops.to_device
    for tag in "abc":
        img = T(g["img_" + tag]).permute(2, 0, 1).unsqueeze(0).contiguous()
        out = metrics.imresize(ops, dev(img), 1.0 / int(g["scale_" + tag]))
        ref = T(g["lr_" + tag]).permute(2, 0, 1).unsqueeze(0).float()
        assert tuple(out.shape) == tuple(ref.shape)
        assert (out.cpu() - ref).abs().max() <= 2e-6, tag
    sr, hr = dev(T(g["psnr_sr"])), dev(T(g["psnr_hr"]))
    assert abs(metrics.psnr(ops, sr, hr) - float(g["psnr_plain"])) <= 1e-4
    assert abs(metrics.psnr(ops, sr, hr, dataset="div2k", scale=4) - float(g["psnr_div2k4"])) <= 1e-4
    assert abs(metrics.psnr(ops, sr, hr, dataset="benchmark", scale=3) - float(g["psnr_bench3"])) <= 1e-4
    # SSIM: identical images -> 1; vs the oracle's restatement on a noisy pair; symmetric
    a, b = T(g["psnr_sr"]), T(g["psnr_hr"])
    s_same = metrics.ssim(ops, dev(a), dev(a)).cpu()
    assert (s_same - 1.0).abs().max() <= 1e-9
    s_ab = metrics.ssim(ops, dev(a), dev(b)).cpu()
    s_ba = metrics.ssim(ops, dev(b), dev(a)).cpu()
    assert (s_ab - s_ba).abs().max() <= 1e-9
    for i in range(a.shape[0]):
        ref = O.calculate_ssim(a[i].permute(1, 2, 0).numpy() * 255.0, b[i].permute(1, 2, 0).numpy() * 255.0)
        assert abs(float(s_ab[i]) - ref) <= 1e-6
    # uint8 formatting: round half to even like numpy
    x = torch.tensor([0.0, 0.5 / 255, 1.5 / 255, 2.5 / 255, 0.999, 1.2, -0.3, 100.4 / 255]).view(1, 1, 2, 4)
    q = ops.to_uint8(dev(x)).cpu()
    assert q.flatten().tolist() == np.round(np.clip(x.numpy().astype(np.float32), 0, 1) * np.float32(255.0)).astype(np.uint8).flatten().tolist()
    # LR consistency of an exact bicubic-downscale pair is (numerically) perfect
    img = dev(T(g["img_a"]).permute(2, 0, 1).unsqueeze(0).contiguous())
    lr = metrics.imresize(ops, img, 0.5)
    assert metrics.lr_consistency_psnr(ops, img, lr, 2) > 120


def test_ssim_oracle_pinned_to_reference(golden_dir):
    """oracle/metrics_ref.calculate_ssim == the genuine LINF-LP/utils.py:152-193 on a random pair and on a real prediction"""
    d = np.load(os.path.join(golden_dir, "linf_detail.npz"))
    assert abs(O.calculate_ssim(d["rand_a"], d["rand_b"]) - float(d["ssim_rand"])) <= 1e-9
    man = __import__("json").load(open(os.path.join(golden_dir, "MANIFEST.json")))
    assert man["extra_r2"]["ssim_oracle_vs_reference"] <= 1e-8


def _check_ssim_and_measure(ops, golden_dir):
    d = np.load(os.path.join(golden_dir, "linf_detail.npz"))
    f = lambda a: ops.to_device(T(np.ascontiguousarray(a.transpose(2, 0, 1)[None] / 255.0)).float())
    s = float(metrics.ssim(ops, f(d["rand_a"]), f(d["rand_b"]))[0])
    assert abs(s - float(d["ssim_rand"])) <= 1e-5               # inputs pass through fp32 [0,1] images on the product side
    from bfsr_amd.srflow.Measure import Measure
    m = np.load(os.path.join(golden_dir, "srflow_measure.npz"))
    assert abs(Measure(ops).psnr(m["a"], m["b"]) - float(m["psnr"])) <= 1e-6
    # Measure.ssim (skimage's structural_similarity, restated in oracle/metrics_ref.skimage_ssim: scikit-image is not in this image, parity unpinned)
    me = Measure(ops)
    a, b = m["a"], m["b"]
    assert abs(me.ssim(a, b) - O.skimage_ssim(a, b)) <= 1e-9
    assert abs(me.ssim(a, a) - 1.0) <= 1e-12 and abs(me.ssim(a, b) - me.ssim(b, a)) <= 1e-12
    g = np.random.Generator(np.random.PCG64(9))
    r1 = g.integers(0, 256, size=(23, 31, 3), dtype=np.uint8)
    r2 = np.clip(r1.astype(np.int32) + g.integers(-20, 21, size=r1.shape), 0, 255).astype(np.uint8)
    assert abs(me.ssim(r1, r2) - O.skimage_ssim(r1, r2)) <= 1e-9
    flat = np.full((9, 9, 3), 100, np.uint8)                   # constant images: means only, SSIM = (2 m1 m2 + C1) / (m1^2 + m2^2 + C1)
    flat2 = np.full((9, 9, 3), 110, np.uint8)
    assert abs(me.ssim(flat, flat2) - (2 * 100 * 110 + 6.5025) / (100 ** 2 + 110 ** 2 + 6.5025)) <= 1e-12
    # (the device sums are accumulated with double atomics: two evaluations agree to the last bits, not bit for bit)
    near = lambda u, v: all(abs(x - y) <= 1e-12 * max(1.0, abs(y)) for x, y in zip(u, v))
    two = me.measure(a, b, with_lpips=False)
    assert len(two) == 2 and near(two, [me.psnr(a, b), me.ssim(a, b)])
    full = me.measure(a, b)                                # the reference's three-element return value; LPIPS (pretrained AlexNet) is NaN here
    assert len(full) == 3 and near(full[:2], [me.psnr(a, b), me.ssim(a, b)]) and full[2] != full[2]
    with pytest.raises(NotImplementedError):
        me.lpips(a, b)


def test_ssim_and_measure_on_cpu_double(golden_dir):
    _check_ssim_and_measure(CpuOps(), golden_dir)


@pytest.mark.gpu
def test_ssim_and_measure_on_hip(golden_dir):
    from bfsr_amd.ops import HipOps
    _check_ssim_and_measure(HipOps("cuda:0"), golden_dir)


def test_product_metrics_on_cpu_double(golden_dir):
    _check_product(CpuOps(), _g(golden_dir))


@pytest.mark.gpu
def test_product_metrics_on_hip(golden_dir):
    from bfsr_amd.ops import HipOps
    _check_product(HipOps("cuda:0"), _g(golden_dir))
