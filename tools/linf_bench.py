"""LINF-LP throughput on one MI355X (BASELINE configs 3/5: rrdb-linf-LP, arbitrary scale).
Usage (GPU box): python tools/linf_bench.py [--batch 16] [--lr 256] [--scales 2,3,4] [--steps 3] [--encoder rrdb]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--lr", type=int, default=256)
    ap.add_argument("--scales", default="2,3,4")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--encoder", default="rrdb")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16"], help="fp16 = BASELINE config 5 MFMA path")
    ap.add_argument("--cpu", action="store_true", help="also time the oracle on one image")
    args = ap.parse_args()
    from bfsr_amd import synth
    from bfsr_amd.linf import spec as lspec
    from bfsr_amd.linf.models import make
    from bfsr_amd.linf.test import infer_from_lr
    from bfsr_amd.ops import HipOps
    ops = HipOps("cuda:0")
    mspec = {"name": "linf-patch", "args": {"encoder_spec": {"name": args.encoder, "args": {"no_upsampling": True}},
                                             "imnet_spec": {"name": "flow", "args": {"name": "flow"}},
                                             "flow_layers": 10, "num_layer": 3, "hidden_dim": 256}}
    sd = synth.state_dict_from_schema(lspec.linf_schema(mspec["args"]["encoder_spec"]), 2024)
    psd = synth.state_dict_from_schema(lspec.linf_prior_schema(27), 777)
    m = make(mspec, args={"ops": ops, "precision": args.precision}).eval()
    m.load_state_dict(sd)
    prior = make({"name": "unet", "args": {"in_chans": 27, "depth": 3, "dim": 64, "bilinear": True}}, args={"ops": ops, "precision": args.precision}).eval()
    prior.load_state_dict(psd)
    B, h = args.batch, args.lr
    xs = [ops.to_device(synth.lr_batch(100 + i, B, h, h)) for i in range(2)]
    for s in [float(v) for v in args.scales.split(",")]:
        H = round(h * s)
        infer_from_lr(m, prior, xs[0], s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            x = xs[i % 2]
            x.add_(0.0)
            out = infer_from_lr(m, prior, x, s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        line = {"workload": "LINF-LP %s-linf-LP x%g, batch=%d %dx%d LR -> %dx%d" % (args.encoder, s, B, h, h, H, H),
                "precision": args.precision, "ms_per_step": round(dt * 1e3, 2), "HR_MPix_per_s": round(B * H * H / 1e6 / dt, 3)}
        if args.cpu:
            import oracle.linf_ref as O
            lr1 = synth.lr_batch(7, 1, h, h)
            torch.set_num_threads(min(16, os.cpu_count() or 1))
            t1 = time.perf_counter()
            prep = O.batch_prep(lr1, (H, H))
            O.lp_pipeline(prep, sd, psd, mspec, (H, H))
            cdt = time.perf_counter() - t1
            line["cpu_oracle_MPix_per_s_16thr"] = round(H * H / 1e6 / cdt, 4)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
