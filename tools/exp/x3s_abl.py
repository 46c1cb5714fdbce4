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


ops.split = "bf16x3"          # this tool ablates conv3x3_x3s_kernel (the 3xBF16 split)
NAMES = {16: "product", 2064: "with per-chunk vmcnt(0)", 1040: "deferred epilogue", 144: "no epilogue"}
for B, H, W in ((8, 160, 160), (16, 256, 256)):
    tot = {k: 0.0 for k in NAMES}
    for Cin, Cout in ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64)):
        x = torch.randn(B, Cin, H, W, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3) * 0.05
        pw = ops.pack_conv_x3(w, None)
        epi = ops.pack_epilogue(Cout, bias=torch.zeros(Cout))
        x3 = ops.x3_pack(x, ops.x3_empty(B, Cin, H, W))
        y3 = ops.x3_empty(B, Cout, H, W)
        r3 = ops.x3_pack(torch.randn(B, Cout, H, W, device="cuda"), ops.x3_empty(B, Cout, H, W))
        kw = dict(epi=epi, act=ACT_LRELU) if Cout == 32 else dict(epi=epi, res1=r3, alpha1=0.2)
        flop = 2.0 * Cin * 9 * Cout * B * H * W
        row = "B%-2d %dx%d %3d->%2d:" % (B, H, W, Cin, Cout)
        for abl, nm in NAMES.items():
            t = timeit(lambda: ops.conv_x3s(x3, pw, y3, tune=-abl, **kw))
            tot[abl] += t
            row += " %s %.0fus %.0fTF |" % (nm, t, flop / t / 1e6)
        print(row, flush=True)
    print("   one RDB: " + ", ".join("%s %.0f us" % (NAMES[k], v) for k, v in tot.items()), flush=True)
