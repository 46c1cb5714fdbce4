"""Latent-module training step -- counterpart of the body of the reference's `train()` (LINF-LP/train.py:118-160).

The LINF model stays frozen on the HIP engine: `gen_feat`, `query_log_p` (the encodes that produce `z_lr` / `z_hr`) and `query_rgb`
run through the kernels; `query_rgb` is differentiable w.r.t. `zmap` (linf.py::_QueryRGB: the inverse flow is affine in z for a fixed
conditioning, its backward is a transposed flow, `bfsr_linf_flow` mode 2).  The latent module (`prior_model`) is whatever
`torch.nn.Module` the caller trains -- its forward/backward are ordinary PyTorch autograd; only the gradient that has to cross the
frozen model comes from this library.  The reference's perceptual term uses a pretrained VGG19 (`train.py:306-307`), which is a
download; `feat_fn` is the hook for it (identity = a plain image-space L1).  Data loading, LR schedules, checkpoint writing and
logging are outside the accelerated path."""
import torch
import torch.nn.functional as F


def train_step(prior_model, linf_model, batch, optimizer=None, latent_weight=1.0, image_weight=0.0, feat_fn=None, patch=True):
    """One iteration of train.py:118-160 on a batch dict (inp, gt, coord, cell, gt_lr_up, gt_patch | gt_pixel, [interpolate_coord]),
    tensors in the reference's normalisation (inp, gt in [0,1]; sub = div = 0.5).  `image_weight` / `feat_fn` are train.py's
    `vgg_weight` / `vgg`.  Returns dict(loss, latent, image) of floats;
    with `optimizer` the step is applied."""
    eng = linf_model.engine()
    d = eng.ops.to_device
    inp = (d(batch["inp"]) - 0.5) / 0.5                                           # train.py:122
    coord, cell = d(batch["coord"]), d(batch["cell"])
    feat = linf_model("gen_feat", inp=inp)                                        # train.py:125 (frozen: no graph needed)
    z_lr = linf_model("query_log_p", inp=inp, feat=feat, coord=coord, cell=cell, gt=d(batch["gt_lr_up"]))[1]     # train.py:129/133
    z_hr = None
    if latent_weight > 0:
        z_hr = linf_model("query_log_p", inp=inp, feat=feat, coord=coord, cell=cell,
                          gt=d(batch["gt_patch" if patch else "gt_pixel"]))[1]    # train.py:131/135
    with torch.enable_grad():
        z_learned = prior_model(z_lr.detach().contiguous(), inp)                  # train.py:137-138
        if not z_learned.requires_grad:
            raise RuntimeError("train_step: prior_model returned a tensor without a graph -- the latent module must be a trainable "
                               "torch.nn.Module (the engine-backed `unet` of this package is inference-only)")
        latent_l = F.l1_loss(z_learned, z_hr.detach()) if latent_weight > 0 else z_learned.new_zeros(())      # train.py:146
        image_l = z_learned.new_zeros(())
        if image_weight > 0:
            pred = linf_model("query_rgb", inp=inp, feat=feat, coord=coord, cell=cell, zmap=z_learned)           # train.py:152
            gt = d(batch["gt"])
            if patch:
                H, W = gt.shape[-2:]
                if "interpolate_coord" in batch:
                    # train.py:154: + F.grid_sample(inp, interpolate_coord.flip(-1), bilinear, border) -- the LR image sampled at
                    # the coordinates of the random out_size x out_size HR sub-crop the training wrapper cut (wrappers.py:741-783).
                    # `inp` carries no graph, so the skip runs on the resampling kernel outside autograd.
                    ic = d(batch["interpolate_coord"]).contiguous()
                    assert tuple(ic.shape[1:3]) == tuple(pred.shape[-2:]), "interpolate_coord must cover the folded prediction"
                    with torch.no_grad():
                        skip = eng.ops.grid_sample_add(inp, ic, torch.zeros_like(pred), torch.empty_like(pred))
                    pred = pred + skip
                else:
                    # full-image batches of the eval wrapper (no sub-crop): the same skip is the plain bilinear resize of
                    # LINF-LP/test.py:171; the reference's train() itself always receives interpolate_coord
                    pred = pred[..., :H, :W] + F.interpolate(inp, (H, W), mode="bilinear", align_corners=False)
            img = torch.clamp(pred * 0.5 + 0.5, 0, 1)
            f = feat_fn if feat_fn is not None else (lambda t: t)
            image_l = F.l1_loss(f(img), f(gt))                                    # train.py:155/157 with vgg := feat_fn
        loss = image_l * image_weight + latent_l * latent_weight                  # train.py:163
        if optimizer is not None:
            optimizer.zero_grad()
        loss.backward()
        if optimizer is not None:
            optimizer.step()
    return dict(loss=float(loss.detach()), latent=float(latent_l.detach()), image=float(image_l.detach()))
