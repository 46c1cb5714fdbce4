"""Which buffer of the prior's branch 0 first differs run to run when it overlaps the main stream (8x model, B >= 16)?  GPU box: python tools/exp/nondet_prior.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd import synth
from bfsr_amd.ops import HipOps
from bfsr_amd.srflow import options, spec
from bfsr_amd.srflow.models import create_model, models as registry
from bfsr_amd.srflow.test import lp_infer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ops = HipOps("cuda:0")
opt = options.derive_scale(options.load(options.DEFAULT_CONF), 8)
m = create_model(opt, ops=ops)
m.load_network(synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234))
prior = registry.make({"name": "unet", "args": {"depth": 3, "dim": 64, "bilinear": True, "ops": ops}, "sd": synth.state_dict_from_schema(spec.srflow_prior_schema(), 4321)}, load_sd=True).eval()
x = ops.to_device(synth.lr_batch(1, B, 96, 96))
pe = prior.engine()
ref, order = None, None
for it in range(4):
    x.add_(0.0)
    out = lp_infer(m, prior, x, return_all=True)
    torch.cuda.synchronize()
    snap = {}
    for k, t in pe.ws.bufs.items():
        snap["ws:" + k] = t
    for k, (want, t) in pe._hb.bufs.items():
        snap["hb:" + k] = t
    snap["epsn0"] = out["epses_norm"][0]
    snap["epsl0"] = out["epses_learned"][0]
    snap["epsl1"] = out["epses_learned"][1]
    if ref is None:
        ref = {k: v.clone() for k, v in snap.items()}
        print("buffers:", len(ref), "mem %.1f GB" % (torch.cuda.memory_allocated() / 1e9), flush=True)
        continue
    bad = []
    for k, v in snap.items():
        a, b = v.float() if v.dtype == torch.float16 else v, ref[k].float() if ref[k].dtype == torch.float16 else ref[k]
        d = (a - b).abs()
        d = torch.nan_to_num(d, nan=1e30)
        n = int((d > 0).sum())
        if n:
            smp = sorted(set((d.flatten(1).max(1).values > 0).nonzero().flatten().tolist()))
            bad.append("%s: %d el, max %.1e, samples %s" % (k, n, float(d.max()), smp[:6]))
            if n < 2000 and d.dim() == 4:
                idx = (d > 0).nonzero()
                print("   ", k, tuple(d.shape), "channels", sorted(set(idx[:, 1].tolist()))[:12], "rows", sorted(set(idx[:, 2].tolist()))[:12], "cols", sorted(set(idx[:, 3].tolist()))[:16])
                print("    first", idx[:6].tolist(), "values now/ref", [(float(a[tuple(i)]), float(b[tuple(i)])) for i in idx[:4].tolist()])
    print("pass %d: %s" % (it, "; ".join(bad) if bad else "identical"), flush=True)
