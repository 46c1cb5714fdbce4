"""GPU check + micro-benchmark of the producer/consumer x3 conv (conv_bf16x3_ps.hip) vs the default x3 kernel."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bfsr_amd.ops import HipOps, ACT_LRELU
ops = HipOps("cuda:0")
torch.manual_seed(0)


def check(B, Cin, Cout, H, W, tune, res=False):
    x = torch.randn(B, Cin, H, W); w = torch.randn(Cout, Cin, 3, 3) * 0.05; b = torch.randn(Cout) * 0.1
    r1 = torch.randn(B, Cout, H, W); r2 = torch.randn(B, Cout, H, W)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = (0.2 * (0.2 * ref + r1.double()) + r2.double()) if res else F.leaky_relu(ref, 0.2)
    pw = ops.pack_conv_x3(w, 1)
    kw = dict(res1=r1.cuda(), alpha1=0.2, res2=r2.cuda(), alpha2=0.2) if res else dict(act=ACT_LRELU, slope=0.2)
    y = ops.conv_x3(x.cuda(), pw, ops.empty(B, Cout, H, W), epi=ops.pack_epilogue(Cout, bias=b), tune=tune, **kw)
    torch.cuda.synchronize()
    err = (y.cpu().double() - ref).abs().max().item()
    ok = err < 2e-5 * max(1.0, ref.abs().max().item())
    print("check tune %d B%d %d->%d %dx%d res=%d: err %.3e %s" % (tune, B, Cin, Cout, H, W, res, err, "ok" if ok else "FAIL"), flush=True)
    return ok


ok = True
for tune in (9004, 9008):
    ok &= check(1, 16, 32, 8, 32, tune)
    ok &= check(2, 64, 32, 19, 45, tune)
    ok &= check(1, 40, 64, 40, 70, tune, res=True)
    ok &= check(2, 48, 24, 9, 33, tune)
    ok &= check(3, 192, 64, 160, 160, tune, res=True)
print("ALL CHECKS", "PASSED" if ok else "FAILED", flush=True)


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for B in (8, 32):
    for Cin, Cout in ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64)):
        H = W = 160
        x = torch.randn(B, Cin, H, W, device="cuda")
        pw = ops.pack_conv_x3(torch.randn(Cout, Cin, 3, 3) * 0.05, 1)
        y = ops.empty(B, Cout, H, W)
        flop = 2.0 * Cin * 9 * Cout * B * H * W
        row = "B%-2d %3d->%2d:" % (B, Cin, Cout)
        for tune in (0, 9004, 9008, 9808):
            t = timeit(lambda: ops.conv_x3(x, pw, y, act=ACT_LRELU, tune=tune))
            row += " tune%d %.0fus %.0fTF |" % (tune, t, flop / t / 1e6)
        print(row, flush=True)
