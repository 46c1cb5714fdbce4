"""GPU micro-benchmark of bfsr_conv2d variants (tune = NR*100+CK) on the shapes of the bench workload.
Usage (GPU box): python tools/conv_bench.py [--quick]   -> table of us / TFLOP/s per (shape, variant)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bfsr_amd.ops import HipOps  # noqa: E402

SHAPES = [
    # name, B, Cin, Cout, H, W, KS, mtile
    ("rdb.conv1 64->32 @160", 8, 64, 32, 160, 160, 3, 1),
    ("rdb.conv1 B16", 16, 64, 32, 160, 160, 3, 1),
    ("rdb.conv1 B32", 32, 64, 32, 160, 160, 3, 1),
    ("rdb.conv1 B64", 64, 64, 32, 160, 160, 3, 1),
    ("rdb.conv1 B2", 2, 64, 32, 160, 160, 3, 1),
    ("lat Cin8 B2", 2, 8, 32, 160, 160, 3, 1),
    ("lat Cin16 B2", 2, 16, 32, 160, 160, 3, 1),
    ("lat Cin128 B2", 2, 128, 32, 160, 160, 3, 1),
    ("lat Cin256 B2", 2, 256, 32, 160, 160, 3, 1),
    ("lat Cin8 B8", 8, 8, 32, 160, 160, 3, 1),
    ("lat Cin256 B8", 8, 256, 32, 160, 160, 3, 1),
    ("lat Cin8 1WG", 1, 8, 32, 8, 32, 3, 1),
    ("lat Cin256 1WG", 1, 256, 32, 8, 32, 3, 1),
    ("rdb.conv4 160->32 @160", 8, 160, 32, 160, 160, 3, 1),
    ("rdb.conv4 B32", 32, 160, 32, 160, 160, 3, 1),
    ("rdb.conv5 B32", 32, 192, 64, 160, 160, 3, 2),
    ("rdb.conv5 mt1", 8, 192, 64, 160, 160, 3, 1),
    ("rdb.conv4 mt1 chk", 8, 160, 32, 160, 160, 3, 1),
    ("rdb.conv5 192->64 @160", 8, 192, 64, 160, 160, 3, 2),
    ("hoist L1 320->1024 @320", 8, 320, 1024, 320, 320, 3, 2),
    ("hoist L2 320->1024 @160", 8, 320, 1024, 160, 160, 3, 2),
    ("hoist L3 320->1024 @80", 8, 320, 1024, 80, 80, 3, 2),
    ("L1 convA 6->64 @320", 8, 6, 64, 320, 320, 3, 2),
    ("L1 1x1 64->64 @320", 8, 64, 64, 320, 320, 1, 2),
    ("L1 convC 64->12 @320", 8, 64, 12, 320, 320, 3, 1),
    ("L3 convA 48->64 @80", 8, 48, 64, 80, 80, 3, 2),
    ("L3 convC 64->96 @80", 8, 64, 96, 80, 80, 3, 3),
    ("unet 64->64 @320", 8, 64, 64, 320, 320, 3, 2),
    ("unet 128->128 @160", 8, 128, 128, 160, 160, 3, 2),
    ("unet 256->256 @40", 8, 256, 256, 40, 40, 3, 2),
]
TUNES = [0, 208, 408, 216, 416]


def up2_bench(ops):
    """the dominant launch: conv over cat[64 key @320^2, up2(256 taps @160^2)] -> 1024, B=8"""
    B, h = 8, 160
    taps, key = torch.randn(B, 256, h, h, device="cuda"), torch.randn(B, 64, 2 * h, 2 * h, device="cuda")
    w = torch.randn(1024, 320, 3, 3) * 0.02
    out = ops.empty(B, 1024, 2 * h, 2 * h)
    flop_t, flop_k = 2.0 * 256 * 4 * 1024 * B * 4 * h * h, 2.0 * 64 * 9 * 1024 * B * 4 * h * h

    def timeit(fn, n=5):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    pt, pk = ops.pack_conv_up2(w[:, 64:].contiguous()), ops.pack_conv(w[:, :64].contiguous(), 2)
    us = timeit(lambda: ops.conv_up2(taps, pt, out, key=(key, pk)))
    print("conv_up2 fp32 (taps+key)      %8.0fus %6.1fTF" % (us, (flop_t + flop_k) / us / 1e6))
    ptx, pkx = ops.pack_conv_up2_x3(w[:, 64:].contiguous()), ops.pack_conv_x3(w[:, :64].contiguous(), 2)
    usk = timeit(lambda: ops.conv_x3(key, pkx, out))
    print("x3 key conv 64->1024 @320     %8.0fus %6.1fTF" % (usk, flop_k / usk / 1e6))
    for t in (402, 401, 801):
        us = timeit(lambda: ops.conv_up2_x3(taps, ptx, out, pre_add=out, tune=t))
        print("x3 taps kernel tune %d       %8.0fus %6.1fTF   (pair: %.0fus %.1fTF)" % (t, us, flop_t / us / 1e6, us + usk,
                                                                                   (flop_t + flop_k) / (us + usk) / 1e6), flush=True)


def main():
    ops = HipOps("cuda:0")
    if "--up2" in sys.argv:
        return up2_bench(ops)
    quick = "--quick" in sys.argv
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    tunes = [int(t) for a in sys.argv if a.startswith("--tunes=") for t in a.split("=", 1)[1].split(",")] or TUNES
    extra = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--shape=")]       # --shape=name,B,Cin,Cout,H,W,KS,mt
    shapes = [(e[0],) + tuple(int(v) for v in e[1:]) for e in extra] or SHAPES
    for name, B, Cin, Cout, H, W, KS, mt in shapes:
        if only and not any(o in name for o in only):
            continue
        x = torch.randn(B, Cin, H, W, device="cuda")
        w = torch.randn(Cout, Cin, KS, KS) * 0.05
        pw = ops.pack_conv(w, mt)
        y = ops.empty(B, Cout, H, W)
        flop = 2.0 * Cin * KS * KS * Cout * B * H * W
        row = []
        for t in tunes:
            if KS == 1 and t % 100 == 8:
                row.append("   -   ")
                continue
            try:
                ops.conv(x, pw, y, tune=t)
                torch.cuda.synchronize()
            except RuntimeError:
                row.append("  n/a  ")
                continue
            n = 3 if quick else 10
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(n):
                ops.conv(x, pw, y, tune=t)
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) / n * 1e3
            row.append("%7.0fus %5.1fTF" % (us, flop / us / 1e6))
        if "--f16" in sys.argv:
            pf = ops.pack_conv_f16(w, min(mt, 2))
            for t in ([int(v) for a in sys.argv if a.startswith('--f16tunes=') for v in a.split('=')[1].split(',')] or [0]):
                ops.conv_f16(x, pf, y, tune=t)
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10):
                    ops.conv_f16(x, pf, y, tune=t)
                e.record()
                torch.cuda.synchronize()
                us = s.elapsed_time(e) / 10 * 1e3
                row.append("f16/%d %7.0fus %5.1fTF" % (t, us, flop / us / 1e6))
        if "--x3" in sys.argv:
            px = ops.pack_conv_x3(w, min(mt, 2))
            for t in ([int(v) for a in sys.argv if a.startswith('--x3tunes=') for v in a.split('=')[1].split(',')] or ([0] if KS == 1 else [200, 400])):
                ops.conv_x3(x, px, y, tune=t)
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10):
                    ops.conv_x3(x, px, y, tune=t)
                e.record()
                torch.cuda.synchronize()
                us = s.elapsed_time(e) / 10 * 1e3
                row.append("x3/%d %7.0fus %5.1fTF" % (t, us, flop / us / 1e6))
        if "--wide" in sys.argv and KS == 1:
            for x3 in (True, False):
                pw1 = ops.pack_conv1x1(w, x3=x3)
                ops.conv1x1(x, pw1, y, x3=x3)
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10):
                    ops.conv1x1(x, pw1, y, x3=x3)
                e.record()
                torch.cuda.synchronize()
                us = s.elapsed_time(e) / 10 * 1e3
                row.append("wide/%s %7.0fus %5.1fTF" % ("x3" if x3 else "f16", us, flop / us / 1e6))
        print("%-26s | %s" % (name, " | ".join(row)), flush=True)
    print("tunes:", TUNES)


if __name__ == "__main__":
    main()
