"""Interleaved sustained A/B of one environment knob on the SRFlow-LP pass (the knob must be read at call time, e.g. BFSR_PRIOR, BFSR_LANES,
BFSR_PRIOR_OVERLAP, BFSR_HOIST).  Boxes differ by +-3 % and bursts run 10-15 % faster than steady state, so: both variants on ONE box, alternating
blocks of `--block` passes, medians per variant.
Usage (GPU box): python tools/env_ab.py VAR A B [--scale 4|8] [--batch 8] [--lr 160] [--rounds 6] [--block 8]"""
import argparse
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("var")
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--lr", type=int, default=160)
    ap.add_argument("--scale", type=int, default=4, choices=[4, 8])
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--block", type=int, default=8)
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
    outs = {}
    for v in (a.a, a.b):
        os.environ[a.var] = v
        outs[v] = lp_infer(m, prior, x).clone()
        x.add_(0.0)
    print("max |A - B| on sr: %.3e" % float((outs[a.a] - outs[a.b]).abs().max()))
    times = {a.a: [], a.b: []}
    for r in range(a.rounds):
        for v in ((a.a, a.b) if r % 2 == 0 else (a.b, a.a)):
            os.environ[a.var] = v
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.block):
                x.add_(0.0)
                lp_infer(m, prior, x)
            torch.cuda.synchronize()
            times[v].append((time.perf_counter() - t0) * 1e3 / a.block)
    for v in (a.a, a.b):
        t = times[v]
        print("%s=%s: median %.2f ms  (min %.2f max %.2f)  %s" % (a.var, v, statistics.median(t), min(t), max(t), " ".join("%.1f" % q for q in t)))
    print("fallbacks", ops.fallbacks)


if __name__ == "__main__":
    main()
