"""Per-launch-shape time table of one LINF-LP pass (BASELINE config 3 or 5) from in-situ HIP events on every launch.
Usage (GPU box): python tools/exp/linf_keys.py [--config 5] [--top 40]"""
import argparse, contextlib, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=5, choices=[3, 5])
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--batch", type=int, default=None)
    a = ap.parse_args()
    from bfsr_amd import synth
    from bfsr_amd.ops import HipOps
    from bfsr_amd.linf import spec as lspec
    from bfsr_amd.linf.models import make
    from bfsr_amd.linf.test import infer_from_lr
    B, h, scale, precision = (16, 256, 4.0, "fp32") if a.config == 3 else (128, 128, 6.0, "fp16")
    if a.batch:
        B = a.batch
    ops = HipOps("cuda:0")
    mspec = {"name": "linf-patch", "args": {"encoder_spec": {"name": "rrdb", "args": {"no_upsampling": True}},
                                             "imnet_spec": {"name": "flow", "args": {"name": "flow"}}, "flow_layers": 10, "num_layer": 3, "hidden_dim": 256}}
    with contextlib.redirect_stdout(sys.stderr):
        model = make(mspec, args={"ops": ops, "precision": precision}).eval()
        model.load_state_dict(synth.state_dict_from_schema(lspec.linf_schema(mspec["args"]["encoder_spec"]), 2024))
        prior = make({"name": "unet", "args": {"in_chans": 27, "depth": 3, "dim": 64, "bilinear": True}}, args={"ops": ops, "precision": precision}).eval()
        prior.load_state_dict(synth.state_dict_from_schema(lspec.linf_prior_schema(27), 777))
    x = ops.to_device(synth.lr_batch(1, B, h, h))
    for _ in range(2):
        x.add_(0.0)
        infer_from_lr(model, prior, x, scale)
    torch.cuda.synchronize()
    ops.profile_keys, ops.profile = "ALL", {}
    x.add_(0.0)
    infer_from_lr(model, prior, x, scale)
    torch.cuda.synchronize()
    import time
    t0 = time.time()
    for _ in range(5):
        x.add_(0.0)
        infer_from_lr(model, prior, x, scale)
    th = time.time() - t0
    torch.cuda.synchronize()
    print("5 passes without events: host enqueue %.1f ms, wall %.1f ms per pass" % (th / 5 * 1e3, (time.time() - t0) / 5 * 1e3))
    rows = sorted(((sum(s.elapsed_time(e) for s, e in ev), len(ev), k) for k, ev in ops.profile.items()), reverse=True)
    print("total event time %.1f ms over %d launches" % (sum(r[0] for r in rows), sum(r[1] for r in rows)))
    for t, n, k in rows[:a.top]:
        print("  %8.2f ms %4d x %9.1f us  %s" % (t, n, t / n * 1e3, k))


main()
