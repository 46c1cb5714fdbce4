"""Micro-benchmark of conv_h2s at the RDB shapes of BASELINE config 5 (B=128 x 128x128 LR): per-shape time, TFLOP/s, algorithmic GB/s.
Usage: python tools/exp/h2s_bench.py [B H W] [tune]"""
import sys
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (128, 128, 128)
tune = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ops = HipOps()
D = ops.h2_empty(B, 192, H, W)
D.copy_(torch.randn(D.shape, device=D.device).half())
N = ops.h2_empty(B, 192, H, W)
g = torch.Generator().manual_seed(0)
tot = 0.0
for cin, cout in ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64)):
    pw = ops.pack_conv_h2s(torch.randn(cout, cin, 3, 3, generator=g) * 0.03)
    epi = ops.pack_epilogue(cout, bias=torch.zeros(cout))
    if cout == 32:
        run = lambda: ops.conv_h2s(D[:, :cin // 8], pw, D[:, cin // 8:cin // 8 + 4], epi=epi, act=2, hi_only=True, tune=tune)
    else:
        run = lambda: ops.conv_h2s(D, pw, N[:, :8], epi=epi, res1=D[:, :8], alpha1=0.2, tune=tune)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * 9 * cin * cout * B * H * W
    by = B * H * W * (cin * 2 + cout * (2 if cout == 32 else 4 + 4))
    tot += ms
    print("cin %3d cout %2d: %.3f ms  %.0f TFLOP/s  %.2f TB/s (algorithmic)" % (cin, cout, ms, fl / ms * 1e-9, by / ms * 1e-9))
print("one RDB: %.3f ms -> x69 = %.1f ms" % (tot, tot * 69))
