"""A/B of the XCD-aware workgroup order in conv_bf16x3 / conv_up2_bf16x3 (tune bit 30 = off)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
OFF = 0x40000000


def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


B = 8
taps = torch.randn(B, 256, 160, 160, device="cuda")
w = torch.randn(1024, 256, 3, 3) * 0.02
out = ops.empty(B, 1024, 320, 320)
ptx = ops.pack_conv_up2_x3(w)
flop = 2.0 * 256 * 4 * 1024 * B * 320 * 320
for rep in range(2):
    for name, t in (("xcd", 0), ("plain", OFF)):
        us = timeit(lambda: ops.conv_up2_x3(taps, ptx, out, pre_add=out, tune=t))
        print("taps kernel %-5s %8.0f us %6.1f TF" % (name, us, flop / us / 1e6), flush=True)
for (Cin, Cout, H, mt) in ((64, 1024, 320, 2), (320, 1024, 160, 2), (320, 1024, 80, 2), (64, 64, 320, 2), (128, 128, 160, 2)):
    x = torch.randn(B, Cin, H, H, device="cuda")
    pw = ops.pack_conv_x3(torch.randn(Cout, Cin, 3, 3) * 0.05, mt)
    y = ops.empty(B, Cout, H, H)
    fl = 2.0 * Cin * 9 * Cout * B * H * H
    for rep in range(2):
        for name, t in (("xcd", 0), ("plain", OFF)):
            us = timeit(lambda: ops.conv_x3(x, pw, y, tune=t))
            print("conv %d->%d @%d %-5s %8.0f us %6.1f TF" % (Cin, Cout, H, name, us, fl / us / 1e6), flush=True)
