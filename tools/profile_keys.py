"""Per-launch-shape time table of one SRFlow-LP pass (cfg2) from in-situ HIP events on every launch.
Usage (GPU box): python tools/profile_keys.py [--batch 8] [--lr 160] [--scale 4|8] [--top 40]   (config 4: --scale 8 --batch 64 --lr 96)"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--lr", type=int, default=160)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--scale", type=int, default=4, choices=[4, 8])
    a = ap.parse_args()
    from bfsr_amd import synth
    from bfsr_amd.ops import HipOps
    from bfsr_amd.srflow import options, spec
    from bfsr_amd.srflow.models import create_model, models as registry
    from bfsr_amd.srflow.test import lp_infer
    ops = HipOps("cuda:0")
    opt = options.load(options.DEFAULT_CONF)
    if a.scale != 4:
        opt = options.derive_scale(opt, a.scale)
    m = create_model(opt, ops=ops)
    m.load_network(synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234))
    prior = registry.make({"name": "unet", "args": {"depth": 3, "dim": 64, "bilinear": True, "ops": ops},
                           "sd": synth.state_dict_from_schema(spec.srflow_prior_schema(), 4321)}, load_sd=True).eval()
    x = ops.to_device(synth.lr_batch(1, a.batch, a.lr, a.lr))
    lp_infer(m, prior, x)
    x.add_(0.0)
    ops.profile_keys, ops.profile = "ALL", {}
    lp_infer(m, prior, x)
    torch.cuda.synchronize()
    ops.profile_keys = None
    rows = []
    tot = 0.0
    for k, ev in ops.profile.items():
        t = sum(s.elapsed_time(e) for s, e in ev)
        tot += t
        rows.append((t, len(ev), k))
    rows.sort(reverse=True)
    print("total event time %.1f ms over %d launches" % (tot, sum(r[1] for r in rows)))
    for t, n, k in rows[:a.top]:
        extra = ""
        if k[0] == "conv":
            _, KS, mt, Cin, Cout, B, H, W = k
            extra = " %6.1f TF" % (2.0 * Cin * KS * KS * Cout * B * H * W * n / (t * 1e-3) / 1e12)
        if k[0] == "conv_up2":
            _, mt, Cin, Cout, B, H, W, c2 = k
            extra = " %6.1f TF" % (2.0 * (Cin * 4 + c2 * 9) * Cout * B * H * W * n / (t * 1e-3) / 1e12)
        print("%8.2f ms %4d x %8.1f us  %s%s" % (t, n, t / n * 1e3, k, extra))


if __name__ == "__main__":
    main()
