"""The level-1 taps conv of config 2 (256 -> 1024 over the x2-upsampled 160^2 taps, B = 8): conv_up2_h2t (h2 input, LDS-DMA, four parities per
item) against the register-staged conv_up2_bf16x3 kernel in its f16x2 mode, both with quad-major pre_add in place.  GPU box: python tools/exp/taps_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
B, Cin, Cout, h = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 256, 1024, 160)
g = torch.Generator().manual_seed(0)
w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.02
x = torch.randn(B, Cin, h, h, device="cuda")
xh = ops.h2_pack(x, ops.h2_empty(B, Cin, h, h))
out = torch.randn(B, Cout, 2 * h, 2 * h, device="cuda")
def timed(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
flop = 2.0 * 16 * Cin * Cout * B * h * h                      # 16 pre-summed taps per source pixel (all four parities)
pk = ops.pack_conv_up2_h2t(w)
t = timed(lambda: ops.conv_up2_h2t(xh, pk, out, pre_add=out))
print("conv_up2_h2t      %d->%d @%dx%d->x2 B%d: %.3f ms  %.0f TFLOP/s fp32-equivalent (x3 products on the fp16 pipe)" % (Cin, Cout, h, h, B, t, flop / t * 1e-9), flush=True)
t0 = timed(lambda: ops.conv_up2_h2t(xh, pk, out))
print("   without pre_add: %.3f ms" % t0, flush=True)
po = ops.pack_conv_up2_x3(w)
t = timed(lambda: ops.conv_up2_x3(x, po, out, pre_add=out, y_fmt=1))
print("conv_up2_x3 (f2)  %d->%d @%dx%d->x2 B%d: %.3f ms  %.0f TFLOP/s fp32-equivalent" % (Cin, Cout, h, h, B, t, flop / t * 1e-9), flush=True)
