"""Randomised shape sweep of the 16-bit conv kernels (3xBF16 / fp16 conv, x2 / x4 parity kernels, wide 1x1) against the CPU test
double: odd sizes, channel-slice views, 1..3 batches.  Usage (GPU box): python tools/fuzz_kernels.py  -> "mismatches: 0"."""
import sys, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bfsr_amd.ops import HipOps
from cpu_ops import CpuOps
hip, CPU = HipOps("cuda:0"), CpuOps()
g = np.random.Generator(np.random.PCG64(2026))
def rnd(*shape, scale=1.0): return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))
bad = 0
for it in range(70):
    B = int(g.integers(1, 4)); Cin = int(g.integers(1, 200)); Cout = int(g.integers(1, 150))
    H = int(g.integers(1, 70)); W = int(g.integers(1, 90)); mt = int(g.integers(1, 3))
    x = rnd(B, Cin, H, W); w = rnd(Cout, Cin, 3, 3, scale=1 / np.sqrt(Cin * 9)); b = rnd(Cout, scale=0.1)
    # channel-slice views on both sides
    xbig = rnd(B, Cin + 5, H, W); xbig[:, 3:3 + Cin] = x
    obig = hip.empty(B, Cout + 7, H, W)
    out = hip.conv_x3(hip.to_device(xbig)[:, 3:3 + Cin], hip.pack_conv_x3(w, mt), obig[:, 2:2 + Cout], epi=hip.pack_epilogue(Cout, bias=b), act=2)
    ref = CPU.conv(x, CPU.pack_conv(w, mt), torch.empty(B, Cout, H, W), bias=b, act=2)
    e = (out.cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    o16 = hip.conv_f16(hip.to_device(x), hip.pack_conv_f16(w, mt), hip.empty(B, Cout, H, W), epi=hip.pack_epilogue(Cout, bias=b), act=2)
    r16 = CPU.conv_f16(x, CPU.pack_conv_f16(w, mt), torch.empty(B, Cout, H, W), bias=b, act=2)
    e16 = (o16.cpu() - r16).abs().max().item() / max(1.0, r16.abs().max().item())
    if e > 1e-5 or e16 > 3e-5:
        bad += 1; print("CONV MISMATCH", (B, Cin, Cout, H, W, mt), e, e16)
for it in range(40):
    B = int(g.integers(1, 3)); Ct = int(g.integers(1, 130)); Cout = int(g.integers(1, 100)); h = int(g.integers(1, 40)); w_ = int(g.integers(1, 50))
    taps = rnd(B, Ct, h, w_); w = rnd(Cout, Ct, 3, 3, scale=1 / np.sqrt(Ct * 9))
    for f, fn, pk in ((2, hip.conv_up2_x3, hip.pack_conv_up2_x3), (4, hip.conv_up4_x3, hip.pack_conv_up4_x3)):
        pre = rnd(B, Cout, f * h, f * w_)
        out = fn(hip.to_device(taps), pk(w), hip.empty(B, Cout, f * h, f * w_), pre_add=hip.to_device(pre), act=1)
        ref = torch.relu(torch.nn.functional.conv2d(torch.nn.functional.interpolate(taps, scale_factor=f, mode="nearest"), w, padding=1) + pre)
        e = (out.cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        if e > 1e-5:
            bad += 1; print("UP%d MISMATCH" % f, (B, Ct, Cout, h, w_), e)
for it in range(30):
    B = int(g.integers(1, 3)); Cin = int(g.integers(1, 300)); Cout = int(g.integers(1, 600)); H = int(g.integers(1, 30)); W = int(g.integers(1, 40))
    x = rnd(B, Cin, H, W); w = rnd(Cout, Cin, 1, 1, scale=1 / np.sqrt(Cin)); b = rnd(Cout, scale=0.1)
    for x3 in (True, False):
        out = hip.conv1x1(hip.to_device(x), hip.pack_conv1x1(w, x3=x3), hip.empty(B, Cout, H, W), x3=x3, epi=hip.pack_epilogue(Cout, bias=b), act=1)
        ref = CPU.conv1x1(x, CPU.pack_conv1x1(w, x3=x3), torch.empty(B, Cout, H, W), x3=x3, bias=b, act=1)
        e = (out.cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        if e > (1e-5 if x3 else 3e-5):
            bad += 1; print("1x1 MISMATCH", (B, Cin, Cout, H, W, x3), e)
print("fuzz done, mismatches:", bad)
