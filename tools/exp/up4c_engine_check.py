"""The 8x engine's level-1 hoisted tensors with the compact x4 path (BFSR_UP4C=1) vs the pre_add path (=0): must be bit-identical.  GPU box."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from bfsr_amd import synth
from bfsr_amd.ops import HipOps
from test_srflow_gpu import build
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
hip = HipOps("cuda:0")
m, prior, opt, sd, psd = build(hip, 8)
eng = m.netG.module.engine()
lr = hip.to_device(synth.smooth_lr_batch(21, B, 96, 96))
res = {}
for v in ("1", "0", "1"):
    os.environ["BFSR_UP4C"] = v
    lr.add_(0.0)
    cond = eng.conditioning(lr)
    c1 = eng._await(cond[1])
    torch.cuda.synchronize()
    cur = {k: c1[k].clone() for k in ("pre_aff", "h_ft")}
    if v in res:
        for k in cur:
            print("run-to-run (UP4C=%s) %s: %s" % (v, k, torch.equal(cur[k], res[v][k])))
    res[v] = cur
for k in ("pre_aff", "h_ft"):
    d = (res["1"][k] - res["0"][k]).abs()
    print(k, tuple(d.shape), "max diff %.3e, differing %d of %d" % (float(d.max()), int((d > 0).sum()), d.numel()))
    if float(d.max()) > 0 and k == "pre_aff":
        q = d.view(d.shape[0], d.shape[1] // 4, d.shape[2], d.shape[3], 4)
        idx = (q > 0).nonzero()
        print("  first:", idx[:4].tolist(), "last:", idx[-2:].tolist(), "samples:", sorted(set(idx[:, 0].tolist())), "rows%4:", sorted(set((idx[:, 2] % 4).tolist())), "quads:", len(set(idx[:, 1].tolist())))
