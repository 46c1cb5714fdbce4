"""LINF-LP evaluation harness -- counterpart of the reference's `test.py` (LINF-LP/test.py:20-236).

`lp_infer` is the LP branch of `eval_psnr` (test.py:94-171, 217) for `--patch` models with a prior:
normalise -> encode (`query_log_p`) -> prior (+ bilinear resize if shapes differ) -> decode (`query_rgb`) ->
crop -> `+= bilinear(inp)` -> clamp(0.5 x + 0.5).  The 256-row chunk loops of `batched_predict(_log_p)` are
result-preserving and are replaced by the kernel grid; `gen_feat` and the per-point conditioning are computed once.
Datasets, PNG I/O, SSIM/LPIPS are outside the accelerated path."""
import math

import torch

from ..ops import MODE_BILINEAR
from . import prep


def batched_predict_log_p(model, inp, coord, cell, gt):
    """-> z (the reference's `model.query_log_p(...)[1]`, LINF-LP/test.py:36-52 + 145).  The LP branch discards the log-density the reference computes
    alongside (270 logf + the Gaussian term per query point), so the engine is asked for z alone: same z up to the rounding of the scale's sigmoid
    (the flow kernel then evaluates it with v_exp / v_rcp, see linf_ops.hip); `model("query_log_p", ...)` still returns the reference's pair."""
    feat = model("gen_feat", inp=inp)
    eng = model.engine() if hasattr(model, "engine") else None
    if eng is None or not hasattr(eng, "query_log_p"):
        return model("query_log_p", inp=inp, feat=feat, coord=coord, cell=cell, gt=gt)[1]
    d = eng.ops.to_device
    return eng.query_log_p(d(feat), d(coord), d(cell), d(gt), with_logp=False)


def batched_predict(model, inp, coord, cell, temperature, zmap=None):
    feat = model("gen_feat", inp=inp)
    return model("query_rgb", inp=inp, feat=feat, coord=coord, cell=cell, temperature=temperature, zmap=zmap)


def lp_infer(model, prior_model, batch, hr_hw, temperature=0, return_all=False):
    """batch: dict(inp [B,3,h,w] in [0,1], coord, cell, gt_lr_up) as delivered by the patch wrappers.
    Runs under the range guard of the two-term fp16 split: an overflow re-runs the pass under the bf16x3 split (guard.run_guarded)."""
    from ..guard import run_guarded
    return run_guarded([model, prior_model], lambda: _lp_infer(model, prior_model, batch, hr_hw, temperature, return_all))


def _lp_infer(model, prior_model, batch, hr_hw, temperature, return_all):
    eng = model.engine()
    ops = eng.ops
    d = ops.to_device
    H, W = hr_hw
    with torch.no_grad():
        inp01 = d(batch['inp'])
        B, _, h, w = inp01.shape
        inp = ops.axpb_clamp(inp01, ops.empty(B, 3, h, w), 2.0, -1.0)            # (inp - 0.5) / 0.5  (test.py:98)
        coord, cell, gt = d(batch['coord']), d(batch['cell']), d(batch['gt_lr_up'])
        z_lr = batched_predict_log_p(model, inp, coord, cell, gt)                # test.py:145
        z_learned = prior_model(z_lr, inp)                                       # test.py:147
        if z_learned.shape != z_lr.shape:                                        # test.py:148-149
            t = ops.empty(*z_lr.shape)
            ops.resize(z_learned, t, MODE_BILINEAR, float(z_learned.shape[2]) / t.shape[2], float(z_learned.shape[3]) / t.shape[3])
            z_learned = t
        if model.patch_size != 1 and prep._fused(ops) and hasattr(eng, "query_rgb"):
            # fold + crop + `+= bilinear(inp)` + clamp(0.5 x + 0.5) as ONE launch over the inverse flow's output (test.py:165-171, 217; round 6: the folded
            # image, the skip image and pred_raw were three more trips of the HR batch through HBM); the same bits as the launches below
            feat = model("gen_feat", inp=inp)
            p = eng.query_rgb(d(feat), coord, cell, z_learned, fold=False)
            pred_raw = ops.empty(B, 3, H, W) if return_all else None
            _, out = ops.linf_fold_skip(p, inp, H, W, eng.ps, raw=pred_raw, out=ops.empty(B, 3, H, W))
            if return_all:
                return dict(z_lr=z_lr, z_learned=z_learned, pred_raw=pred_raw, pred=out)
            return out
        full = batched_predict(model, inp, coord, cell, temperature, z_learned)  # test.py:165
        if model.patch_size == 1:        # pixel-wise LINF: the skip is already inside query_rgb, no fold (test.py:168, 217)
            pred_raw = full[..., :H, :W].contiguous()
            out = ops.axpb_clamp(pred_raw, ops.empty(B, 3, H, W), 0.5, 0.5, 0.0, 1.0)
            if return_all:
                return dict(z_lr=z_lr, z_learned=z_learned, pred_raw=pred_raw, pred=out)
            return out
        pred = full[..., :H, :W]                                                 # test.py:168 (a view; planes stay contiguous only if W is full)
        if pred.shape[-1] != full.shape[-1]:
            c = ops.empty(B, 3, H, W)
            ops.patch_fold(eng.ws.bufs["flow_out"], c, eng.ps)                   # fold again with the crop applied
            pred = c
        elif pred.shape[-2] != full.shape[-2]:
            pred = pred.contiguous()
        skip = ops.resize(inp, ops.empty(B, 3, H, W), MODE_BILINEAR, float(h) / H, float(w) / W)   # test.py:171
        pred_raw = ops.axpb_clamp(pred, ops.empty(B, 3, H, W), 1.0, 0.0, r=skip)
        out = ops.axpb_clamp(pred_raw, ops.empty(B, 3, H, W), 0.5, 0.5, 0.0, 1.0)                  # test.py:217
    if return_all:
        return dict(z_lr=z_lr, z_learned=z_learned, pred_raw=pred_raw, pred=out)
    return out


def infer_from_lr(model, prior_model, inp01, scale, always_pad=True, **kw):
    """LR tensor in -> HR tensor out: device-side input prep (prep.prepare_batch) + lp_infer."""
    ops = model.engine().ops
    inp01 = ops.to_device(inp01)
    h, w = inp01.shape[-2:]
    H, W = round(h * scale), round(w * scale)
    if model.patch_size == 1:
        batch = prep.prepare_batch_pixelwise(ops, inp01, (H, W))
    else:
        batch = prep.prepare_batch(ops, inp01, (H, W), model.patch_size, always_pad)
    return lp_infer(model, prior_model, batch, (H, W), **kw)


def calc_psnr(sr, hr, dataset=None, scale=1, rgb_range=1):
    """LINF-LP/utils.py:132-149."""
    diff = (sr - hr) / rgb_range
    if dataset is not None:
        if dataset == 'benchmark':
            shave = scale
            if diff.size(1) > 1:
                gray = diff.new_tensor([65.738, 129.057, 25.064]).view(1, 3, 1, 1) / 256
                diff = diff.mul(gray).sum(dim=1)
        elif dataset == 'div2k':
            shave = scale
        else:
            raise NotImplementedError
        valid = diff[..., shave:-shave, shave:-shave]
    else:
        valid = diff
    return -10 * torch.log10(valid.pow(2).mean())


def eval_psnr(loader, model, prior_model=None, eval_type=None, patch=True, temperature=0, randomness=False, n_samples=5,
              detail=False):
    """Average PSNR over an iterable of batch dicts (LINF-LP/test.py:50-236, `eval_bsize` branch).
    prior_model given: the LP path (encode -> prior -> decode).  prior_model None: the stochastic path, z ~ N(0, tau^2)
    sampled on the device (linf.py:398).  randomness=True reproduces the `--randomness` loop (test.py:151-162, 203-208):
    `n_samples` predictions per batch, mean PSNR and the diversity score (std over samples of the uint8 images)."""
    from . import metrics
    ops = model.engine().ops
    tot, div, n = 0.0, 0.0, 0
    tot_ssim, tot_lr = 0.0, 0.0

    def one(batch, H, W):
        if prior_model is not None:
            return lp_infer(model, prior_model, batch, (H, W), temperature)
        from ..guard import run_guarded
        return run_guarded([model], lambda: tau(batch, H, W))

    def tau(batch, H, W):
        ops = model.engine().ops
        d = ops.to_device
        inp01 = d(batch['inp'])
        B, _, h, w = inp01.shape
        inp = ops.axpb_clamp(inp01, ops.empty(B, 3, h, w), 2.0, -1.0)
        full = batched_predict(model, inp, d(batch['coord']), d(batch['cell']), temperature)
        pred = full[..., :H, :W].contiguous()
        skip = ops.resize(inp, ops.empty(B, 3, H, W), MODE_BILINEAR, float(h) / H, float(w) / W)
        raw = ops.axpb_clamp(pred, ops.empty(B, 3, H, W), 1.0, 0.0, r=skip)
        return ops.axpb_clamp(raw, ops.empty(B, 3, H, W), 0.5, 0.5, 0.0, 1.0)

    def psnr(pred, gt):
        if eval_type is None:
            return calc_psnr(pred, gt)
        kind, s = eval_type.split('-')
        return calc_psnr(pred, gt, dataset=kind, scale=int(s))

    for batch in loader:
        H, W = batch['gt'].shape[-2:]
        preds = [one(batch, H, W) for _ in range(n_samples if randomness else 1)]
        gt = batch['gt'].to(preds[0].device)
        bsz = preds[0].shape[0]
        tot += sum(float(psnr(p, gt)) for p in preds) / len(preds) * bsz
        if randomness:
            q = torch.stack([torch.round(p * 255.0) for p in preds], 1)
            div += float(torch.std(q, dim=1).mean()) * bsz
        if detail:      # test.py:172-200: SSIM on [0,255] images and LR-consistency PSNR through imresize(pred, 1/scale), on device
            sc = int(eval_type.split('-')[1]) if eval_type else round(H / batch['inp'].shape[-2])
            inp01 = ops.to_device(batch['inp'])
            tot_ssim += sum(float(metrics.ssim(ops, p, gt).mean()) for p in preds) / len(preds) * bsz
            kind = eval_type.split('-')[0] if eval_type else None
            tot_lr += sum(metrics.lr_consistency_psnr(ops, p, inp01, sc, dataset=kind) for p in preds) / len(preds) * bsz
        n += bsz
    res = {'psnr': tot / max(n, 1)}
    if detail:
        res.update({'ssim': tot_ssim / max(n, 1), 'LR recon': tot_lr / max(n, 1)})      # 'lpips' (pretrained AlexNet) is outside the path
    if randomness:
        res['diversity'] = div / max(n, 1)
    if randomness or detail:
        return res
    return tot / max(n, 1)
