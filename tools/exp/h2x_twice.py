"""Every conv_h2x launch of an encode run twice on the same inputs; report launches whose two outputs differ (debug aid)."""
import os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from bfsr_amd import synth
from test_srflow_gpu import build
from bfsr_amd.ops import HipOps, MODE_BILINEAR
hip = HipOps("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = int(sys.argv[2]) if len(sys.argv) > 2 else 160
m, prior, opt, sd, psd = build(hip, S)
eng = m.netG.module.engine()
f = hip.conv_h2x
n = [0, 0]
def g(x, pw, out, **k):
    f(x, pw, out, **k)
    a = out.clone()
    f(x, pw, out, **k)
    n[0] += 1
    if not torch.equal(a.view(torch.int16) if a.dtype == torch.float16 else a.view(torch.int32), out.view(torch.int16) if out.dtype == torch.float16 else out.view(torch.int32)):
        n[1] += 1
        d = (a.float() - out.float()).abs()
        idx = (d > 0).nonzero() if not torch.isnan(d).any() else torch.isnan(d).nonzero()
        if n[1] <= 12:
            print("launch %d differs: x %s out %s %s res1 %s; %d elements, first %s last %s, max %.3e, vals %s vs %s" % (
                n[0], tuple(x.shape), tuple(out.shape), out.dtype, k.get("res1") is not None, idx.shape[0], idx[0].tolist(), idx[-1].tolist(), float(torch.nan_to_num(d).max()),
                a[tuple(idx[0].tolist())].item(), out[tuple(idx[0].tolist())].item()), flush=True)
            if idx.shape[0] < 64:
                print("    all:", idx.tolist(), flush=True)
    return out
hip.conv_h2x = g
lr = hip.to_device(synth.smooth_lr_batch(61, 8, L, L))
lr_up = hip.resize(lr, hip.empty(8, 3, S * L, S * L), MODE_BILINEAR, 1.0 / S, 1.0 / S)
for r in range(3):
    eng._cond = None if hasattr(eng, "_cond") else None
    ep = eng.encode(lr_up + 0.001 * r, lr + 0.001 * r)
    print("round", r, "launches", n[0], "differing", n[1], "encode nan:", [int(torch.isnan(e).sum()) for e in ep], flush=True)
