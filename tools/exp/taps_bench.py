"""The level-1 conditioning conv of config 2 (cat[64 key @320^2, 256 taps @160^2 upsampled] -> 1024, B = 8): conv_up2_h2t with the key channels
folded in as space-to-depth chunks, conv_up2_h2t (taps only) + pre_add, and the round-3 form (conv_h2x key conv, then the register-staged
conv_up2_bf16x3 kernel in its f16x2 mode with pre_add).  GPU box: python tools/exp/taps_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
B, Ct, Ck, Cout, h = (int(v) for v in sys.argv[1:6]) if len(sys.argv) >= 6 else (8, 256, 64, 1024, 160)
g = torch.Generator().manual_seed(0)
w = torch.randn(Cout, Ck + Ct, 3, 3, generator=g) * 0.02
x = torch.randn(B, Ct, h, h, device="cuda")
key = torch.randn(B, Ck, 2 * h, 2 * h, device="cuda")
xh = ops.h2_empty(B, Ct + 4 * Ck, h, h)
out = torch.randn(B, Cout, 2 * h, 2 * h, device="cuda")
def timed(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
ft, fk = 2.0 * 16 * Ct * Cout * B * h * h, 2.0 * 36 * Ck * Cout * B * h * h      # taps: 16 pre-summed taps per source pixel; key: 9 taps x 4 output pixels
tp = timed(lambda: (ops.h2_pack(x, xh[:, :Ct // 8]), ops.h2_pack_s2d(key, xh[:, Ct // 8:])))
print("h2_pack + h2_pack_s2d: %.3f ms" % tp, flush=True)
pk = ops.pack_conv_up2_h2t(w[:, Ck:].contiguous(), w[:, :Ck].contiguous())
t = timed(lambda: ops.conv_up2_h2t(xh, pk, out))
print("conv_up2_h2t taps + key fused   B%d %d+%d->%d @%d^2: %.3f ms  %.0f TFLOP/s fp32-equivalent (x3 products on the fp16 pipe)" % (B, Ct, Ck, Cout, h, t, (ft + fk) / t * 1e-9), flush=True)
pt = ops.pack_conv_up2_h2t(w[:, Ck:].contiguous())
t1 = timed(lambda: ops.conv_up2_h2t(xh[:, :Ct // 8], pt, out, pre_add=out))
t0 = timed(lambda: ops.conv_up2_h2t(xh[:, :Ct // 8], pt, out))
print("conv_up2_h2t taps only: %.3f ms with pre_add in place (%.0f TFLOP/s), %.3f ms without" % (t1, ft / t1 * 1e-9, t0), flush=True)
kh = ops.h2_pack(key, ops.h2_empty(B, Ck, 2 * h, 2 * h))
pkk = ops.pack_conv_x3(w[:, :Ck].contiguous(), 1)
tk = timed(lambda: ops.conv_x3s(kh, pkk, out, y_fmt=1))
po = ops.pack_conv_up2_x3(w[:, Ck:].contiguous())
t2 = timed(lambda: ops.conv_up2_x3(x, po, out, pre_add=out, y_fmt=1))
print("round 3: conv_h2x key conv %.3f ms + conv_up2_x3 (f2) %.3f ms (%.0f TFLOP/s)" % (tk, t2, ft / t2 * 1e-9), flush=True)
