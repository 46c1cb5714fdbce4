"""Host-enqueue vs GPU time of one cfg2 LP pass, and whether the pass can be captured in a HIP graph.
Usage (GPU box): python tools/graph_probe.py [--batch 8] [--lr 160]"""
import argparse
import contextlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--lr", type=int, default=160)
    a = ap.parse_args()
    from bfsr_amd import synth
    from bfsr_amd.ops import HipOps
    from bfsr_amd.srflow import options, spec
    from bfsr_amd.srflow.models import create_model, models as registry
    from bfsr_amd.srflow.test import lp_infer
    ops = HipOps("cuda:0")
    opt = options.load(options.DEFAULT_CONF)
    sd = synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234)
    psd = synth.state_dict_from_schema(spec.srflow_prior_schema(), 4321)
    model = create_model(opt, ops=ops)
    model.load_network(sd)
    with contextlib.redirect_stdout(sys.stderr):
        prior = registry.make({"name": "unet", "args": {"depth": 3, "dim": 64, "bilinear": True, "ops": ops}, "sd": psd},
                              load_sd=True).eval()
    x = ops.to_device(synth.lr_batch(0, a.batch, a.lr, a.lr))
    for _ in range(2):
        x.add_(0.0)
        sr = lp_infer(model, prior, x)
    torch.cuda.synchronize()
    for _ in range(3):
        x.add_(0.0)
        t0 = time.perf_counter()
        sr = lp_infer(model, prior, x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("eager: host enqueue %.1f ms, total %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
    ref = sr.clone()
    try:
        g = torch.cuda.CUDAGraph()
        x.add_(0.0)
        with torch.cuda.graph(g):
            sr_g = lp_infer(model, prior, x, check_range=False)
        torch.cuda.synchronize()
        for _ in range(3):
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            print("graph replay: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
        print("graph == eager:", float((sr_g - ref).abs().max()))
    except Exception as e:  # noqa
        print("graph capture failed:", repr(e)[:500])


if __name__ == "__main__":
    main()
