"""Golden vector of ONE latent-module training iteration, produced by the GENUINE reference's `train()` (LINF-LP/train.py:88-172)
run in the build container (CPU; `.cuda()` is a no-op through ref_import.py):

  linf_train_step.npz   a batch of two items of the reference's own training wrapper `sr-implicit-downsampled-fast-crop-patch`
                        (datasets/wrappers.py:686-783: random scale, random out_size x out_size HR sub-crop, `interpolate_coord`),
                        edsr-baseline linf-patch (frozen, seeded synthetic weights), a small conv latent module and a fixed seeded
                        2-layer conv in place of the pretrained VGG19 (`train.vgg`, a download); both loss weights > 0, patch=True.
                        Stored: the batch, the latent module's / feature net's weights, the two loss terms `train()` returns and the
                        latent module's parameter gradients left behind by `loss.backward()` (optimizer = SGD with lr 0).
                        MANIFEST "train_step" records the difference of the same objective written on the oracle restatement.
Usage:  python tests/golden/make_golden_train.py
"""
import importlib
import json
import os
import random
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_import as R  # noqa: E402
from make_golden_linf import MODEL_SPECS, maxdiff, save  # noqa: E402

VGG_W, LATENT_W = 1.3, 0.7


class Tiny(torch.nn.Module):
    """The latent module of the fixture: z + 0.1 conv3x3(z) + 0.05 conv1x1(bilinear(inp))."""

    def __init__(self):
        super().__init__()
        self.c = torch.nn.Conv2d(27, 27, 3, padding=1)
        self.i = torch.nn.Conv2d(3, 27, 1)

    def forward(self, z, inp):
        return z + 0.1 * self.c(z) + 0.05 * self.i(F.interpolate(inp, z.shape[-2:], mode="bilinear", align_corners=False))


class Feat(torch.nn.Module):
    """Stand-in for the VGG19 feature extractor (train.py:306-307)."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Conv2d(3, 8, 3, padding=1)
        self.b = torch.nn.Conv2d(8, 8, 3, padding=1)

    def forward(self, x):
        return self.b(F.relu(self.a(x)))


class Images(torch.utils.data.Dataset):
    def __init__(self, imgs):
        self.imgs = imgs

    def __len__(self):
        return len(self.imgs)

    def __getitem__(self, i):
        return self.imgs[i]


class _Writer(object):
    def add_scalars(self, *a, **k):
        pass


def main():
    from bfsr_amd import synth
    from bfsr_amd.linf import spec as lspec
    import oracle.linf_ref as O
    R.use_linf()
    models = importlib.import_module("models")
    wrappers = importlib.import_module("datasets.wrappers")
    train_mod = importlib.import_module("train")
    mspec = MODEL_SPECS["edsr"]
    model = models.make(mspec).eval()
    sd = synth.state_dict_from_schema(lspec.linf_schema(mspec["args"]["encoder_spec"]), 2025)
    model.load_state_dict(sd, strict=True)
    for prm in model.parameters():
        prm.requires_grad_(False)

    def resize_fn(img, size):
        """wrappers.py:241-244 (ToPILImage -> Resize(BICUBIC) -> ToTensor) without torchvision: only makes the LR crop, i.e. INPUT data."""
        from PIL import Image
        a = img.mul(255).byte().permute(1, 2, 0).numpy()
        r = Image.fromarray(a).resize((size[1], size[0]), Image.BICUBIC)
        return torch.from_numpy(np.asarray(r).copy()).permute(2, 0, 1).float().div(255)

    wrappers.resize_fn = resize_fn
    random.seed(4242)
    torch.manual_seed(4242)
    imgs = [synth.smooth_lr_batch(500 + i, 1, 80, 80)[0] for i in range(2)]
    ds = wrappers.SRImplicitDownsampledFastCropPatch(Images(imgs), inp_size=6, scale_max=4, augment=True, patch_size=3)
    items = [ds[i] for i in range(2)]
    batch = {k: torch.stack([it[k] for it in items]) for k in items[0]}

    torch.manual_seed(11)
    prior = Tiny()
    torch.manual_seed(12)
    feat_net = Feat().eval()
    for prm in feat_net.parameters():
        prm.requires_grad_(False)
    opt = torch.optim.SGD(prior.parameters(), lr=0.0)
    train_mod.config = {"loss_weight": {"vgg": VGG_W, "latent": LATENT_W},
                        "data_norm": {"inp": {"sub": [0.5], "div": [0.5]}, "gt": {"sub": [0.5], "div": [0.5]}},
                        "train_dataset": {"batch_size": 2, "dataset": {"args": {"repeat": 1}}}}
    train_mod.writer = _Writer()
    train_mod.vgg = feat_net
    with torch.enable_grad():
        vgg_l, latent_l = train_mod.train([dict(batch)], prior, model, opt, 1, patch=True)       # the genuine loop body
    grads = {n: p.grad.detach().clone() for n, p in prior.named_parameters()}

    # the same objective on the oracle restatement, end to end under autograd
    torch.manual_seed(11)
    ref_prior = Tiny()
    espec = mspec["args"]["encoder_spec"]
    inp = (batch["inp"] - 0.5) / 0.5
    feat = O.encoder(inp, sd, espec)
    z_lr = O.query_log_p(feat, batch["coord"], batch["cell"], batch["gt_lr_up"], sd)
    z_hr = O.query_log_p(feat, batch["coord"], batch["cell"], batch["gt_patch"], sd)
    with torch.enable_grad():
        zl = ref_prior(z_lr, inp)
        pred = O.query_rgb(feat, batch["coord"], batch["cell"], zl, sd)
        pred = pred + F.grid_sample(inp, batch["interpolate_coord"].flip(-1), mode="bilinear", padding_mode="border", align_corners=False)
        o_vgg = F.l1_loss(feat_net(torch.clamp(pred * 0.5 + 0.5, 0, 1)), feat_net(batch["gt"]))
        o_lat = F.l1_loss(zl, z_hr)
        (o_vgg * VGG_W + o_lat * LATENT_W).backward()
    man_path = os.path.join(HERE, "MANIFEST.json")
    man = json.load(open(man_path))
    man["train_step"] = dict(vgg=abs(float(o_vgg) - vgg_l), latent=abs(float(o_lat) - latent_l),
                             grads=max(maxdiff(p.grad, grads[n]) for n, p in ref_prior.named_parameters()))
    json.dump(man, open(man_path, "w"), indent=1, sort_keys=True)
    out = {k: v for k, v in batch.items()}
    out.update({"prior." + n: p.detach() for n, p in prior.named_parameters()})
    out.update({"feat." + n: p.detach() for n, p in feat_net.named_parameters()})
    out.update({"grad." + n: g for n, g in grads.items()})
    save("linf_train_step.npz", vgg_loss=np.float64(vgg_l), latent_loss=np.float64(latent_l), vgg_weight=np.float64(VGG_W),
         latent_weight=np.float64(LATENT_W), weights_seed=np.int64(2025), **out)
    print("train_step:", man["train_step"], "vgg %.6f latent %.6f" % (vgg_l, latent_l),
          {k: tuple(v.shape) for k, v in batch.items()}, "|grad|max", {n: float(g.abs().max()) for n, g in grads.items()})


if __name__ == "__main__":
    main()
