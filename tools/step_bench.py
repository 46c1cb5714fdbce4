"""Micro-benchmark of the coupled FlowStep remainder at BASELINE config 2's level shapes: fused kernel (coupling_step) vs the
head + tail pair, A/B interleaved in one process.  Usage (GPU box): python tools/step_bench.py [rounds]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bfsr_amd.ops import HipOps  # noqa: E402

ops = HipOps("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = np.random.Generator(np.random.PCG64(1))
r = lambda *s, scale=1.0: torch.from_numpy((g.standard_normal(s) * scale).astype(np.float32))


def timed(f, n=20):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for C, hw, B in ((12, 320, 8), (24, 160, 8), (12, 384, 8), (24, 192, 8)):
    cn, cc2 = C // 2, 2 * (C - C // 2)
    w0, w2 = r(64, cn, 3, 3, scale=0.1), r(64, 64, 1, 1, scale=0.1)
    s0, c0, s2, c2 = r(64, scale=0.1), torch.exp(r(64, scale=0.1)), r(64, scale=0.1), torch.exp(r(64, scale=0.1))
    w4, b4, ps = r(cc2, 64, 3, 3, scale=0.02), r(cc2, scale=0.2), torch.exp(r(cc2, scale=0.2))
    Wm = torch.from_numpy(np.linalg.qr(g.standard_normal((C, C)))[0].astype(np.float32))
    bias, es = ops.vec(r(C, scale=0.1)), ops.vec(torch.exp(r(C, scale=0.1)))
    z = torch.randn(B, C, hw, hw, device="cuda")
    z2 = torch.empty_like(z)
    pre = torch.randn(B, 64, hw, hw, device="cuda") * 0.5
    hf = torch.randn(B, 2 * C, hw, hw, device="cuda") * 0.5
    hid = torch.empty(B, 64, hw, hw, device="cuda")
    spk = ops.pack_coupling_step(w0, w2, s0, c0, s2, c2, w4, b4, ps)
    hpk = ops.pack_coupling_head(w0, w2, s0, c0, s2, c2)
    tpk = ops.pack_coupling_tail(w4, b4, ps)
    wv = ops.vec(Wm)
    for rev in (1, 0):
        fused = lambda: ops.coupling_step(z, z2, spk, pre, rev, h_ft=hf, w=wv, an_bias=bias, an_escale=es)

        def pair():
            ops.coupling_head(z, hpk, pre, hid, hid_fmt=1)
            ops.coupling_tail(hid, tpk, z, z2, rev, h_ft=hf, w=wv, an_bias=bias, an_escale=es, hid_fmt=1)
        tf, tp, th = [], [], []
        for _ in range(rounds):
            tf.append(timed(fused)); tp.append(timed(pair)); th.append(timed(lambda: ops.coupling_head(z, hpk, pre, hid, hid_fmt=1)))
        px = B * hw * hw
        alg = px * (256 + 4 * C * 4) / 1e6          # MB: pre_aff + z in + h_ft (2C) + z out
        mf, mp = float(np.median(tf)), float(np.median(tp))
        print("C=%d %dx%d B=%d rev=%d: fused %.1f us (min %.1f) = %.2f TB/s algorithmic (%.0f MB)   head+tail %.1f us (head %.1f)   x%.2f"
              % (C, hw, hw, B, rev, mf, min(tf), alg / mf, alg, mp, float(np.median(th)), mp / mf), flush=True)
