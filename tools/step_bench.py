"""Micro-benchmark of the coupled FlowStep remainder (coupling_head -> coupling_tail) at BASELINE config 2 / 4 level shapes, every kernel
timed alone with HIP events, NCHW and quad-major hand-over of pre_aff / h_ft.  Usage (GPU box): python tools/step_bench.py [rounds]
Algorithmic bytes per step: 20*C*B*h*w (z, h_aff, h_ft in; z out); the pair's own traffic adds pre_aff in and hid
out + in (256 B/px each)."""
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
    hid = ops.h2_empty(B, 64, hw, hw)
    hpk = ops.pack_coupling_head(w0, w2, s0, c0, s2, c2)
    tpk = ops.pack_coupling_tail(w4, b4, ps)
    wv = ops.vec(Wm)
    px = B * hw * hw
    for rev in (1, 0):
        for fmt in (0, 1):
            head = lambda: ops.coupling_head(z, hpk, pre, hid, pre_fmt=fmt)
            tail = lambda: ops.coupling_tail(hid, tpk, z, z2, rev, h_ft=hf, w=wv, an_bias=bias, an_escale=es, h_ft_fmt=fmt)
            th, tt = [], []
            for _ in range(rounds):
                th.append(timed(head)); tt.append(timed(tail))
            mh, mt = float(np.median(th)), float(np.median(tt))
            bh, bt = px * (4 * cn + 256 + 256) / 1e6, px * (256 + 4 * C * 4) / 1e6      # MB moved by head / tail
            print("C=%d %dx%d B=%d rev=%d quad-major=%d: head %.1f us (%.2f TB/s of its %.0f MB)  tail %.1f us (%.2f TB/s of its %.0f MB; %.2f TB/s on 20*C B/px)  step %.1f us"
                  % (C, hw, hw, B, rev, fmt, mh, bh / mh, bh, mt, bt / mt, bt, px * 20 * C / 1e6 / mt, mh + mt), flush=True)
ops.check_range()
