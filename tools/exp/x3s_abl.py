"""Ablation timing of conv3x3_x3s_kernel variants (tools/exp/libabl.so, built with -DBFSR_X3S_ABL)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bfsr_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libabl.so")
from bfsr_amd.ops import HipOps, ACT_LRELU
ops = HipOps("cuda:0")


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


NAMES = {0: "full", 16: "2 loaders", 48: "3 loaders", 80: "4 loaders", 1: "noDMA"}
for B in (8, 32):
    for Cin, Cout in ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64)):
        H = W = 160
        x = torch.randn(B, Cin, H, W, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3) * 0.05
        pw = ops.pack_conv_x3(w, None)
        x3 = ops.x3_pack(x, ops.x3_empty(B, Cin, H, W))
        y3 = ops.x3_empty(B, Cout, H, W)
        flop = 2.0 * Cin * 9 * Cout * B * H * W
        row = "B%-2d %3d->%2d:" % (B, Cin, Cout)
        ref = ops.conv_x3s(x3, pw, ops.x3_empty(B, Cout, H, W), act=ACT_LRELU, tune=0).clone()
        for abl in (16, 48, 80):
            got = ops.conv_x3s(x3, pw, ops.x3_empty(B, Cout, H, W), act=ACT_LRELU, tune=-abl)
            torch.cuda.synchronize()
            if not torch.equal(got.view(torch.int16), ref.view(torch.int16)):
                row += " [variant %d WRONG]" % abl
        for abl, nm in NAMES.items():
            t = timeit(lambda: ops.conv_x3s(x3, pw, y3, act=ACT_LRELU, tune=-abl))
            row += " %s %.0fus %.0fTF |" % (nm, t, flop / t / 1e6)
        print(row, flush=True)
