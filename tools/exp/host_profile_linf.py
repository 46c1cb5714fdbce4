"""cProfile of the host side of one LINF-LP pass (where does the enqueue time go?).  Usage: python tools/exp/host_profile_linf.py [--batch 16]"""
import argparse, contextlib, cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--config", type=int, default=5)
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
    for _ in range(5):
        x.add_(0.0)
        infer_from_lr(model, prior, x, scale)
    th = time.time() - t0
    torch.cuda.synchronize()
    print("5 passes: host enqueue %.1f ms, wall %.1f ms per pass" % (th / 5 * 1e3, (time.time() - t0) / 5 * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        x.add_(0.0)
        infer_from_lr(model, prior, x, scale)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(25)


main()
