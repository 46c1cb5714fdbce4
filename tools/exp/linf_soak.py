"""LINF-LP reproducibility soak: N LP passes (input prep included) at a config-5-like and a config-3-like shape, every output compared bit for bit with the first pass.
GPU box: python tools/exp/linf_soak.py [passes]"""
import contextlib, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd import synth
from bfsr_amd.ops import HipOps
from bfsr_amd.linf import spec as lspec
from bfsr_amd.linf.models import make
from bfsr_amd.linf.test import infer_from_lr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ops = HipOps("cuda:0")
mspec = {"name": "linf-patch", "args": {"encoder_spec": {"name": "rrdb", "args": {"no_upsampling": True}},
                                         "imnet_spec": {"name": "flow", "args": {"name": "flow"}}, "flow_layers": 10, "num_layer": 3, "hidden_dim": 256}}
for (B, h, scale, precision) in ((32, 128, 6.0, "fp16"), (4, 256, 4.0, "fp32"), (16, 128, 6.0, "fp16")):
    with contextlib.redirect_stdout(sys.stderr):
        model = make(mspec, args={"ops": ops, "precision": precision}).eval()
        model.load_state_dict(synth.state_dict_from_schema(lspec.linf_schema(mspec["args"]["encoder_spec"]), 2024))
        prior = make({"name": "unet", "args": {"in_chans": 27, "depth": 3, "dim": 64, "bilinear": True}}, args={"ops": ops, "precision": precision}).eval()
        prior.load_state_dict(synth.state_dict_from_schema(lspec.linf_prior_schema(27), 777))
    x = ops.to_device(synth.lr_batch(3, B, h, h))
    t0 = time.time()
    ref, bad = None, 0
    for i in range(N):
        x.add_(0.0)
        out = infer_from_lr(model, prior, x, scale)
        if ref is None:
            ref = out.clone()
        elif not torch.equal(out, ref):
            bad += 1
    torch.cuda.synchronize()
    print("LINF soak B=%d %dx%d x%g %s: %d passes, %d differ from the first (%.0f s), fallbacks %d" % (B, h, h, scale, precision, N, bad, time.time() - t0, ops.fallbacks), flush=True)
    del model, prior
    torch.cuda.empty_cache()
