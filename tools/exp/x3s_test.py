"""GPU check + micro-benchmark of conv_x3s (LDS-DMA conv over x3 tensors) against conv_x3 (register-staged fp32 input)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bfsr_amd.ops import HipOps, ACT_LRELU  # noqa: E402

ops = HipOps("cuda:0")
torch.manual_seed(0)


def check(B, Cin, Cout, H, W, res=False, fp32_out=False):
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, 3, 3) * 0.05
    b = torch.randn(Cout) * 0.1
    r1 = torch.randn(B, Cout, H, W)
    r2 = torch.randn(B, Cout, H, W)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if res:
        ref = 0.2 * (0.2 * ref + r1.double()) + r2.double()
    else:
        ref = F.leaky_relu(ref, 0.2)
    pw = ops.pack_conv_x3(w, 1)
    epi = ops.pack_epilogue(Cout, bias=b)
    x3 = ops.x3_pack(x.cuda(), ops.x3_empty(B, Cin, H, W))
    back = ops.x3_unpack(x3, ops.empty(B, Cin, H, W))
    assert torch.equal(back.cpu(), x), "x3 pack/unpack is not lossless"
    if res:
        kw = dict(res1=ops.x3_pack(r1.cuda(), ops.x3_empty(B, Cout, H, W)), alpha1=0.2,
                  res2=ops.x3_pack(r2.cuda(), ops.x3_empty(B, Cout, H, W)), alpha2=0.2)
    else:
        kw = dict(act=ACT_LRELU, slope=0.2)
    if fp32_out:
        y = ops.conv_x3s(x3, pw, ops.empty(B, Cout, H, W), epi=epi, **kw)
    else:
        y3 = ops.conv_x3s(x3, pw, ops.x3_empty(B, Cout, H, W), epi=epi, **kw)
        y = ops.x3_unpack(y3, ops.empty(B, Cout, H, W))
    torch.cuda.synchronize()
    err = (y.cpu().double() - ref).abs().max().item()
    print("check B%d %d->%d %dx%d res=%d fp32out=%d: max-abs err vs fp64 %.3e (|ref| %.2f)" % (B, Cin, Cout, H, W, res, fp32_out, err, ref.abs().max()), flush=True)
    return err < 2e-5 * max(1.0, ref.abs().max().item())


if "--perf-only" not in sys.argv:
    ok = True
    ok &= check(1, 16, 32, 8, 32)
    ok &= check(2, 64, 32, 19, 45)
    ok &= check(1, 32, 64, 40, 70, res=True)
    ok &= check(2, 48, 24, 9, 33, fp32_out=True)
    ok &= check(1, 192, 64, 33, 65, res=True, fp32_out=True)
    ok &= check(3, 64, 32, 160, 160)
    print("ALL CHECKS", "PASSED" if ok else "FAILED", flush=True)


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for B in (8, 32):
    for Cin, Cout in ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64)):
        H = W = 160
        x = torch.randn(B, Cin, H, W, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3) * 0.05
        pw = ops.pack_conv_x3(w, None)
        x3 = ops.x3_pack(x, ops.x3_empty(B, Cin, H, W))
        y3 = ops.x3_empty(B, Cout, H, W)
        y = ops.empty(B, Cout, H, W)
        flop = 2.0 * Cin * 9 * Cout * B * H * W
        t_old = timeit(lambda: ops.conv_x3(x, pw, y, act=ACT_LRELU))
        t_new = timeit(lambda: ops.conv_x3s(x3, pw, y3, act=ACT_LRELU))
        row = "B%-2d %3d->%2d @160: conv_x3 %6.0f us %6.1f TF | conv_x3s %6.0f us %6.1f TF" % (B, Cin, Cout, t_old, flop / t_old / 1e6, t_new, flop / t_new / 1e6)
        for g in (128, 512):
            try:
                t = timeit(lambda: ops.conv_x3s(x3, pw, y3, act=ACT_LRELU, tune=g))
                row += " | grid%d %6.0f us %6.1f TF" % (g, t, flop / t / 1e6)
            except RuntimeError:
                row += " | grid%d n/a" % g
        print(row, flush=True)
