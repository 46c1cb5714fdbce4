"""Level-2 hoist of config 2 (3x3 conv 320 -> 1024 at 8 x 160^2, quad-major fp32 out): register-staged split conv on the fp32 tensor against
h2_pack + conv_h2x.  GPU box: python tools/exp/hoist2_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator().manual_seed(0)
def timed(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for B, hw in ((8, 160), (8, 80)):
    x = torch.randn(B, 320, hw, hw, device="cuda")
    w = torch.randn(1024, 320, 3, 3, generator=g) * 0.02
    out = ops.empty(B, 1024, hw, hw)
    p2 = ops.pack_conv_x3(w, 2)
    t0 = timed(lambda: ops.conv_x3(x, p2, out, y_fmt=1))
    xh = ops.h2_empty(B, 320, hw, hw)
    tp = timed(lambda: ops.h2_pack(x, xh))
    p1 = ops.pack_conv_x3(w, 1, lazy=True)
    t1 = timed(lambda: ops.conv_h2x(xh, p1, out, y_fmt=1))
    print("%dx%d^2 320->1024: conv_x3 (register-staged) %.3f ms; h2_pack %.3f + conv_h2x %.3f ms" % (B, hw, t0, tp, t1), flush=True)
