"""Round-2 golden vectors from the GENUINE reference (build container only; see make_golden.py / ref_import.py):

  linf_e2e_rrdb_s6.npz   rrdb-linf-LP at the out-of-distribution x6 scale (BASELINE config 5 in miniature)
  linf_sampling.npz      the stochastic LINF path: `query_rgb(zmap=None, temperature)` (linf.py:397-398) with the reference's
                         `torch.randn` call replaced by a recorded noise tensor
  linf_detail.npz        the reference's own `eval_psnr(detail=True)` dict (psnr / ssim / LR recon) for eval_type None, 'div2k-4',
                         'benchmark-4' and `utils.calculate_ssim` -- with cv2.getGaussianKernel / cv2.filter2D supplied by
                         ref_import.py (numpy/scipy restatements of the two OpenCV calls, cv2 is not installed) and LPIPS stubbed
  srflow_sampling.npz    SRFlowNet reverse with a given z and Split2d's eps sampled through a patched
                         GaussianDiag.sample_eps (Split.py:66-70, flow.py:113-119) = the `get_sr_with_z` / tau path
  srflow_measure.npz     Measure.psnr (Measure.py:50-52) on uint8 images (skimage's peak_signal_noise_ratio restated in ref_import.py)
Usage:  python tests/golden/make_golden_extra.py
"""
import importlib
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_import as R  # noqa: E402
from make_golden_linf import MODEL_SPECS, PRIOR_SPEC, maxdiff, save  # noqa: E402

torch.set_grad_enabled(False)


class Pair(torch.utils.data.Dataset):
    def __init__(self, lr, hr):
        self.lr, self.hr = lr, hr

    def __len__(self):
        return 1

    def __getitem__(self, i):
        return self.lr, self.hr


def gen_linf(man):
    from bfsr_amd import synth
    from bfsr_amd.linf import spec as lspec
    import oracle.linf_ref as O
    import oracle.metrics_ref as MO
    R.use_linf()
    models = importlib.import_module("models")
    wrappers = importlib.import_module("datasets.wrappers")
    test_mod = importlib.import_module("test")
    utils = importlib.import_module("utils")

    prior = models.make(PRIOR_SPEC).eval()
    psd = synth.state_dict_from_schema(lspec.linf_prior_schema(27), 777)
    prior.load_state_dict(psd, strict=True)

    # ---- rrdb x6 (BASELINE config 5 in miniature): same recipe as make_golden_linf.py's cases
    mspec = MODEL_SPECS["rrdb"]
    model = models.make(mspec).eval()
    sd = synth.state_dict_from_schema(lspec.linf_schema(mspec["args"]["encoder_spec"]), 2024)
    model.load_state_dict(sd, strict=True)
    s, (h, w) = 6, (10, 8)
    lr = synth.smooth_lr_batch(2024 + s, 1, h, w)[0]
    H, W = s * h, s * w
    item = wrappers.SRImplicitPairedFastPatch(Pair(lr, torch.rand(3, H, W)), patch_size=3)[0]
    batch = {k: v.unsqueeze(0) for k, v in item.items()}
    inp = (batch["inp"] - 0.5) / 0.5
    z_lr = test_mod.batched_predict_log_p(model, inp, batch["coord"], batch["cell"], batch["gt_lr_up"]).contiguous()
    z_learned = prior(z_lr, inp)
    if z_learned.shape != z_lr.shape:
        z_learned = F.interpolate(z_learned, size=z_lr.shape[-2:], mode="bilinear", align_corners=False)
    pred = test_mod.batched_predict(model, inp, batch["coord"], batch["cell"], 0, z_learned)[..., :H, :W]
    pred = pred + F.interpolate(inp, pred.shape[-2:], mode="bilinear", align_corners=False)
    out = torch.clamp(pred * 0.5 + 0.5, 0, 1)
    rt = test_mod.batched_predict(model, inp, batch["coord"], batch["cell"], 0, z_lr)      # decode of the un-modified latent
    o = O.lp_pipeline({k: batch[k] for k in ("inp", "coord", "cell", "gt_lr_up")}, sd, psd, mspec, (H, W), return_all=True)
    man["e2e_rrdb_s6"] = dict(z_lr=maxdiff(o["z_lr"], z_lr), z_learned=maxdiff(o["z_learned"], z_learned),
                              pred_raw=maxdiff(o["pred_raw"], pred), pred=maxdiff(o["pred"], out))
    save("linf_e2e_rrdb_s6.npz", lr=lr.unsqueeze(0), scale=np.int64(s), coord=batch["coord"], cell=batch["cell"],
         gt_lr_up=batch["gt_lr_up"], z_lr=z_lr, z_learned=z_learned, pred_raw=pred, pred=out, roundtrip_fold=rt, weights_seed=np.int64(2024),
         prior_seed=np.int64(777), weights_sha256=np.frombuffer(synth.digest(sd).encode(), dtype=np.uint8))

    # ---- stochastic path: temperature sampling with recorded noise (edsr-baseline, x4, 12x10 LR)
    mspec_e = MODEL_SPECS["edsr"]
    model_e = models.make(mspec_e).eval()
    sd_e = synth.state_dict_from_schema(lspec.linf_schema(mspec_e["args"]["encoder_spec"]), 2025)
    model_e.load_state_dict(sd_e, strict=True)
    lr = synth.smooth_lr_batch(77, 1, 12, 10)[0]
    H, W = 48, 40
    item = wrappers.SRImplicitPairedFastPatch(Pair(lr, torch.rand(3, H, W)), patch_size=3)[0]
    batch = {k: v.unsqueeze(0) for k, v in item.items()}
    inp = (batch["inp"] - 0.5) / 0.5
    qh, qw = batch["coord"].shape[1:3]
    g = np.random.Generator(np.random.PCG64(4242))
    noise = torch.from_numpy(g.standard_normal((qh * qw, 27)).astype(np.float32))       # the shape linf.py:398 asks torch.randn for
    real_randn = torch.randn
    torch.randn = lambda *a, **k: noise.clone()
    try:
        pred = test_mod.batched_predict(model_e, inp, batch["coord"], batch["cell"], 0.8)[..., :H, :W]
    finally:
        torch.randn = real_randn
    pred = pred + F.interpolate(inp, pred.shape[-2:], mode="bilinear", align_corners=False)
    out = torch.clamp(pred * 0.5 + 0.5, 0, 1)
    zmap = (noise * 0.8).view(1, qh, qw, 27).permute(0, 3, 1, 2).contiguous()
    feat = O.encoder(inp, sd_e, mspec_e["args"]["encoder_spec"])
    mine = O.query_rgb(feat, batch["coord"], batch["cell"], zmap, sd_e)[..., :H, :W] + F.interpolate(inp, (H, W), mode="bilinear", align_corners=False)
    man["sampling_edsr"] = maxdiff(mine, pred)
    save("linf_sampling.npz", lr=lr.unsqueeze(0), coord=batch["coord"], cell=batch["cell"], noise=noise, temperature=np.float32(0.8),
         pred_raw=pred, pred=out, weights_seed=np.int64(2025))

    # ---- eval_psnr(detail=True) of the reference itself + calculate_ssim
    lr = synth.smooth_lr_batch(31, 1, 48, 48)[0]
    hr = F.interpolate(lr.unsqueeze(0), scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)[0]
    item = wrappers.SRImplicitPairedFastPatch(Pair(lr, hr), patch_size=3)[0]
    res = {}
    norm = {"inp": {"sub": [0.5], "div": [0.5]}, "gt": {"sub": [0.5], "div": [0.5]}}
    for et in (None, "div2k-4", "benchmark-4"):
        batch = {k: v.unsqueeze(0) for k, v in item.items()}
        kw = dict(prior_model=prior, data_norm=norm, eval_type=et, eval_bsize=300000, patch=True, detail=True)
        if et is None:
            kw["scale_max"] = 4
        # eval_psnr reads `scale` only inside the eval_type branches (test.py:69-75); for eval_type=None the LR-consistency
        # imresize(…, 1/scale) would hit an unbound name, so the None case is run through the pieces instead
        if et is None:
            continue
        d = test_mod.eval_psnr([dict(batch)], model_e, **kw)
        res[et] = {k: float(v) for k, v in d.items() if k != "lpips"}
    o = O.lp_pipeline({k: item[k].unsqueeze(0) for k in ("inp", "coord", "cell", "gt_lr_up")}, sd_e, psd, mspec_e, (192, 192))
    pred_np = o[0].permute(1, 2, 0).numpy()
    hr_np = hr.permute(1, 2, 0).numpy()
    ssim_ref = float(utils.calculate_ssim(pred_np * 255.0, hr_np * 255.0))
    man["ssim_oracle_vs_reference"] = abs(MO.calculate_ssim(pred_np.astype(np.float64) * 255.0, hr_np.astype(np.float64) * 255.0) - ssim_ref)
    a, b = torch.rand(21, 34, 3, generator=torch.Generator().manual_seed(5)).numpy() * 255, torch.rand(21, 34, 3, generator=torch.Generator().manual_seed(6)).numpy() * 255
    save("linf_detail.npz", lr=lr.unsqueeze(0), hr=hr.unsqueeze(0), ssim_pred_hr=np.float64(ssim_ref),
         rand_a=a, rand_b=b, ssim_rand=np.float64(utils.calculate_ssim(a, b)),
         **{("%s_%s" % (et.replace("-", ""), k)).replace(" ", "_"): np.float64(v) for et, d in res.items() for k, v in d.items()})
    man["detail"] = res


def gen_srflow(man):
    from bfsr_amd import synth
    from bfsr_amd.srflow import options, spec
    import oracle.srflow_ref as O
    opt_ref = R.srflow_opt(4)
    opt = options.load(options.DEFAULT_CONF)
    net = R.build_srflownet(opt_ref)
    sd = synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234)
    net.load_state_dict(sd, strict=True)
    flow_mod = importlib.import_module("models.modules.flow")
    lr = synth.smooth_lr_batch(91, 2, 16, 12)
    g = np.random.Generator(np.random.PCG64(808))
    heat = 0.8
    z = torch.from_numpy((g.standard_normal((2, 96, 8, 6)) * heat).astype(np.float32))           # get_z (SRFlow_model.py:224-237)
    eps = torch.from_numpy((g.standard_normal((2, 6, 32, 24)) * heat).astype(np.float32))        # what sample_eps would draw
    real = flow_mod.GaussianDiag.sample_eps
    calls = []

    def fake(shape, eps_std, seed=None):
        calls.append((tuple(shape), eps_std))
        return eps.clone()
    flow_mod.GaussianDiag.sample_eps = staticmethod(fake)
    try:
        sr, logdet = net(lr=lr, z=z, eps_std=heat, reverse=True, epses=None, reverse_with_grad=True)
    finally:
        flow_mod.GaussianDiag.sample_eps = real
    assert calls == [((2, 6, 32, 24), heat)], calls
    mine = O.srflow_reverse_flow(lr, [eps, z], sd, opt, 23)
    mine = mine[0] if isinstance(mine, tuple) else mine
    man["sampling_srflow"] = maxdiff(mine, sr)
    save("srflow_sampling.npz", lr=lr, z=z, eps=eps, heat=np.float32(heat), sr=sr, logdet=logdet)

    # Measure.psnr on uint8 HWC images (Measure.py:50-52 -> skimage.metrics.peak_signal_noise_ratio, restated in ref_import.py)
    measure = importlib.import_module("Measure")
    rng = np.random.Generator(np.random.PCG64(3))
    a = rng.integers(0, 256, (37, 52, 3), dtype=np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-20, 21, a.shape), 0, 255).astype(np.uint8)
    m = measure.Measure.__new__(measure.Measure)
    save("srflow_measure.npz", a=a, b=b, psnr=np.float64(m.psnr(a, b)))


def main():
    path = os.path.join(HERE, "MANIFEST.json")
    man = json.load(open(path))
    extra = man.setdefault("extra_r2", {})
    gen_linf(extra)
    gen_srflow(extra)
    json.dump(man, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(extra, indent=1))


if __name__ == "__main__":
    main()
