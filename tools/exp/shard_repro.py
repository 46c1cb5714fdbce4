"""Reproducer for an intermittent difference between encode(B=2)[1] and encode(B=1) (tests/test_srflow_gpu.py::
test_roundtrip_and_batch_invariance_at_bench_size): N rounds, reports which of the two calls is not reproducible and where."""
import os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from bfsr_amd import synth
from bfsr_amd.ops import HipOps, MODE_BILINEAR
from test_srflow_gpu import build

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SCALE, LR = int(os.environ.get("REPRO_SCALE", "4")), int(os.environ.get("REPRO_LR", "160"))      # e.g. REPRO_SCALE=8 REPRO_LR=96: BASELINE config 4's model
hip = HipOps("cuda:0")
m, prior, opt, sd, psd = build(hip, SCALE)
eng = m.netG.module.engine()
lr = hip.to_device(synth.smooth_lr_batch(21, 2, LR, LR))
lr_up = hip.resize(lr, hip.empty(2, 3, LR * SCALE, LR * SCALE), MODE_BILINEAR, 1.0 / SCALE, 1.0 / SCALE)
lr1, lr_up1 = lr[1:2].clone(), lr_up[1:2].clone()
ref2 = ref1 = None
for it in range(N):
    ep = [e.clone() for e in eng.encode(lr_up, lr)]
    if os.environ.get("REPRO_DECODE", "1") == "1":
        rt = eng.decode(lr, epses=[e.clone() for e in ep])
        err = (rt - lr_up).abs().max().item()
    ep1 = [e.clone() for e in eng.encode(lr_up1, lr1)]
    torch.cuda.synchronize()
    if ref2 is None:
        ref2, ref1 = ep, ep1
    msg = []
    for lvl, (a, b, c, d_) in enumerate(zip(ep, ref2, ep1, ref1)):
        if not torch.equal(a, b):
            df = (a - b).abs(); idx = torch.nonzero(df > 0)
            msg.append("B=2 eps%d differs from round 0: max %.3e, %d elements, samples %s, first %s" % (lvl, float(df.max()), idx.shape[0], sorted(set(idx[:, 0].tolist())), idx[0].tolist()))
        if not torch.equal(c, d_):
            df = (c - d_).abs(); idx = torch.nonzero(df > 0)
            msg.append("B=1 eps%d differs from round 0: max %.3e, %d elements, first %s" % (lvl, float(df.max()), idx.shape[0], idx[0].tolist()))
        if not torch.equal(a[1:2], c):
            df = (a[1:2] - c).abs(); idx = torch.nonzero(df > 0)
            msg.append("eps%d: B=2[1] != B=1: max %.3e, %d elements" % (lvl, float(df.max()), idx.shape[0]))
    print("round %d: %s" % (it, "; ".join(msg) if msg else "ok"), flush=True)
