"""N eager LINF-LP passes (BASELINE config 5 / 3 shapes) for timeline profiling.  Usage: python tools/exp/linf_loop.py [--batch 16] [--passes 6]"""
import argparse, contextlib, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--config", type=int, default=5)
    ap.add_argument("--passes", type=int, default=6)
    a = ap.parse_args()
    from bfsr_amd import synth
    from bfsr_amd.ops import HipOps
    from bfsr_amd.linf import spec as lspec
    from bfsr_amd.linf.models import make
    from bfsr_amd.linf.test import infer_from_lr
    B, h, scale, precision = (a.batch, 256, 4.0, "fp32") if a.config == 3 else (a.batch, 128, 6.0, "fp16")
    ops = HipOps("cuda:0")
    mspec = {"name": "linf-patch", "args": {"encoder_spec": {"name": "rrdb", "args": {"no_upsampling": True}},
                                             "imnet_spec": {"name": "flow", "args": {"name": "flow"}}, "flow_layers": 10, "num_layer": 3, "hidden_dim": 256}}
    with contextlib.redirect_stdout(sys.stderr):
        model = make(mspec, args={"ops": ops, "precision": precision}).eval()
        model.load_state_dict(synth.state_dict_from_schema(lspec.linf_schema(mspec["args"]["encoder_spec"]), 2024))
        prior = make({"name": "unet", "args": {"in_chans": 27, "depth": 3, "dim": 64, "bilinear": True}}, args={"ops": ops, "precision": precision}).eval()
        prior.load_state_dict(synth.state_dict_from_schema(lspec.linf_prior_schema(27), 777))
    x = ops.to_device(synth.lr_batch(1, B, h, h))
    for _ in range(3):
        x.add_(0.0)
        infer_from_lr(model, prior, x, scale)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.passes):
        x.add_(0.0)
        infer_from_lr(model, prior, x, scale)
    torch.cuda.synchronize()
    print("%d passes: %.1f ms per pass" % (a.passes, (time.time() - t0) / a.passes * 1e3))


main()
