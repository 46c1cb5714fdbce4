"""conv_f16 (round-1 register-staged fp16 kernel, one product) against conv_x3 (conv_bf16x3_kernel<PL=2>, three products) on the cfg5 shapes that still
run on conv_f16: how much of conv_f16's time is its structure rather than its MFMAs?  GPU box: python tools/exp/f16_vs_f2.py"""
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
for B, cin, cout, hw in ((128, 64, 64, 257), (128, 128, 64, 257), (128, 64, 512, 128), (128, 155, 32, 257), (128, 256, 128, 128)):
    x = torch.randn(B, cin, hw, hw, device="cuda")
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.03
    y = ops.empty(B, cout, hw, hw)
    pf, px = ops.pack_conv_f16(w, 2 if cout > 32 else None), ops.pack_conv_x3(w, 2 if cout > 32 else None)
    t1 = timed(lambda: ops.conv_f16(x, pf, y))
    t2 = timed(lambda: ops.conv_x3(x, px, y))
    fl = 2.0 * cin * 9 * cout * B * hw * hw
    print("B%d %d->%d @%d^2: conv_f16 %.3f ms (%.0f TFLOP/s), conv_x3 f16x2 %.3f ms (%.0f TFLOP/s-eq)" % (B, cin, cout, hw, t1, fl / t1 * 1e-9, t2, fl / t2 * 1e-9), flush=True)
