"""Golden vector for the backward of `query_rgb` into the latent (LINF-LP/train.py:143: the image-space loss of the latent module
back-propagates through the frozen model's inverse flow), from the GENUINE reference's autograd (build container only):

  linf_vjp.npz   edsr-baseline linf-patch, x4, 12x10 LR: zmap, a random cotangent R of the folded prediction, pred = query_rgb(zmap)
                 and d/dzmap sum(pred * R) computed by torch.autograd through the reference modules; MANIFEST "vjp" records the
                 difference of the oracle restatement's autograd gradient (0.0 = bit-identical).
Usage:  python tests/golden/make_golden_vjp.py
"""
import importlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_import as R  # noqa: E402
from make_golden_linf import MODEL_SPECS, maxdiff, save  # noqa: E402
from make_golden_extra import Pair  # noqa: E402


def main():
    from bfsr_amd import synth
    from bfsr_amd.linf import spec as lspec
    import oracle.linf_ref as O
    R.use_linf()
    models = importlib.import_module("models")
    wrappers = importlib.import_module("datasets.wrappers")
    mspec = MODEL_SPECS["edsr"]
    model = models.make(mspec).eval()
    sd = synth.state_dict_from_schema(lspec.linf_schema(mspec["args"]["encoder_spec"]), 2025)
    model.load_state_dict(sd, strict=True)
    for prm in model.parameters():
        prm.requires_grad_(False)
    lr = synth.smooth_lr_batch(78, 1, 12, 10)[0]
    H, W = 48, 40
    item = wrappers.SRImplicitPairedFastPatch(Pair(lr, torch.rand(3, H, W)), patch_size=3)[0]
    batch = {k: v.unsqueeze(0) for k, v in item.items()}
    inp = (batch["inp"] - 0.5) / 0.5
    qh, qw = batch["coord"].shape[1:3]
    g = np.random.Generator(np.random.PCG64(9090))
    z0 = torch.from_numpy(g.standard_normal((1, 27, qh, qw)).astype(np.float32))
    with torch.enable_grad():
        zmap = z0.clone().requires_grad_(True)
        feat = model("gen_feat", inp=inp)
        pred = model("query_rgb", inp=inp, feat=feat, coord=batch["coord"], cell=batch["cell"], zmap=zmap)     # train.py:143
        cot = torch.from_numpy(g.standard_normal(tuple(pred.shape)).astype(np.float32))
        (pred * cot).sum().backward()
        grad = zmap.grad.detach().clone()
        # the oracle restatement under autograd
        zo = z0.clone().requires_grad_(True)
        po = O.query_rgb(O.encoder(inp, sd, mspec["args"]["encoder_spec"]), batch["coord"], batch["cell"], zo, sd)
        (po * cot).sum().backward()
    man_path = os.path.join(HERE, "MANIFEST.json")
    man = json.load(open(man_path))
    man["vjp"] = dict(pred=maxdiff(po.detach(), pred.detach()), grad_z=maxdiff(zo.grad, grad))
    json.dump(man, open(man_path, "w"), indent=1, sort_keys=True)
    save("linf_vjp.npz", lr=lr.unsqueeze(0), coord=batch["coord"], cell=batch["cell"], zmap=z0, cotangent=cot, pred=pred.detach(),
         grad_z=grad, weights_seed=np.int64(2025))
    print("vjp:", man["vjp"], "pred", tuple(pred.shape), "|grad|max %.3e" % grad.abs().max().item())


if __name__ == "__main__":
    main()
