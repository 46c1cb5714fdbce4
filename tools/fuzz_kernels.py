"""Randomised shape sweep of the 16-bit conv kernels (3xBF16 / fp16 conv, x2 / x4 parity kernels, wide 1x1; the LDS-DMA family over h2 tensors: conv_h2x,
conv_chain, conv_up2_h2t, conv_up4_h2t against fp64 convs of the same 22-bit inputs) against the CPU test
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
# ---- the LDS-DMA family over h2 tensors: conv_h2x, a conv_chain of random length (bit-identical to the launches), conv_up2_h2t, conv_up4_h2t
F = torch.nn.functional
for it in range(40):
    B = int(g.integers(1, 4)); Cin = 16 * int(g.integers(1, 13)); Cout = 32 * int(g.integers(1, 5)); H = int(g.integers(1, 70)); W = int(g.integers(1, 90))
    x = rnd(B, Cin, H, W); w = rnd(Cout, Cin, 3, 3, scale=1 / np.sqrt(Cin * 9)); b = rnd(Cout, scale=0.1)
    xh = hip.h2_pack(hip.to_device(x), hip.h2_empty(B, Cin, H, W))
    x22 = hip.h2_unpack(xh, hip.empty(B, Cin, H, W)).cpu().double()
    ref = F.leaky_relu(F.conv2d(x22, w.double(), b.double(), 1, 1), 0.2)
    pw, epi = hip.pack_conv_x3(w, 1, lazy=True), hip.pack_epilogue(Cout, bias=b)
    out = hip.conv_h2x(xh, pw, hip.empty(B, Cout, H, W), epi=epi, act=2, slope=0.2)
    e = (out.cpu().double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    if e > 4e-6:
        bad += 1; print("H2X MISMATCH", (B, Cin, Cout, H, W), e)
for it in range(25):
    B = int(g.integers(1, 4)); H = int(g.integers(8, 70)); W = int(g.integers(8, 90)); n = int(g.integers(2, 7)); gc = 32
    D = hip.h2_empty(B, 64 + gc * n, H, W)
    hip.h2_pack(hip.to_device(rnd(B, 64, H, W, scale=0.5)), D[:, :8])
    specs = []
    for i in range(n):
        Cin = 64 + gc * i
        specs.append(dict(x=D[:, :Cin // 8], pw=hip.pack_conv_x3(rnd(gc, Cin, 3, 3, scale=1 / np.sqrt(Cin * 9)), 1, lazy=True), out=D[:, Cin // 8:(Cin + gc) // 8],
                          epi=hip.pack_epilogue(gc, bias=rnd(gc, scale=0.1)), act=2, slope=0.2))
    last = hip.empty(B, 64, H, W)
    specs.append(dict(x=D, pw=hip.pack_conv_x3(rnd(64, 64 + gc * n, 3, 3, scale=0.02), 1, lazy=True), out=last, epi=hip.pack_epilogue(64, bias=rnd(64, scale=0.1))))
    for sp in specs:
        hip.conv_h2x(sp["x"], sp["pw"], sp["out"], epi=sp["epi"], act=sp.get("act", 0), slope=0.2)
    want_d, want = D.clone(), last.clone()
    D[:, 8:].zero_(); last.zero_()
    hip.conv_chain(specs).run()
    hip.check_range()
    if not (torch.equal(D, want_d) and torch.equal(last, want)):
        bad += 1; print("CHAIN MISMATCH", (B, H, W, n))
for it in range(30):
    B = int(g.integers(1, 3)); Ct = 16 * int(g.integers(1, 9)); Ck = 16 * int(g.integers(0, 3)); Cout = 32 * int(g.integers(1, 4)); h = int(g.integers(1, 40)); w_ = int(g.integers(1, 50))
    taps, key = rnd(B, Ct, h, w_), rnd(B, max(Ck, 1), 2 * h, 2 * w_)[:, :Ck]
    wt = rnd(Cout, Ct + Ck, 3, 3, scale=1 / np.sqrt((Ct + Ck) * 9))
    xh = hip.h2_empty(B, Ct + 4 * Ck, h, w_)
    hip.h2_pack(hip.to_device(taps), xh[:, :Ct // 8])
    if Ck:
        hip.h2_pack_s2d(hip.to_device(key.contiguous()), xh[:, Ct // 8:])
    t22 = hip.h2_unpack(xh[:, :Ct // 8], hip.empty(B, Ct, h, w_)).cpu().double()
    ref = F.conv2d(F.interpolate(t22, scale_factor=2, mode="nearest"), wt[:, Ck:].double(), None, 1, 1)
    if Ck:
        k22 = hip.h2_unpack(hip.h2_pack(hip.to_device(key.contiguous()), hip.h2_empty(B, Ck, 2 * h, 2 * w_)), hip.empty(B, Ck, 2 * h, 2 * w_)).cpu().double()
        ref = ref + F.conv2d(k22, wt[:, :Ck].double(), None, 1, 1)
    out = hip.conv_up2_h2t(xh, hip.pack_conv_up2_h2t(wt[:, Ck:].contiguous(), wt[:, :Ck].contiguous() if Ck else None), hip.empty(B, Cout, 2 * h, 2 * w_))
    e = (CPU.quads(out.cpu(), inverse=True).double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    if e > 4e-6:
        bad += 1; print("UP2_H2T MISMATCH", (B, Ct, Ck, Cout, h, w_), e)
    w4 = wt[:, Ck:].contiguous()
    pre = rnd(B, Cout, 4 * h, 4 * w_)
    ref4 = F.conv2d(F.interpolate(t22, scale_factor=4, mode="nearest"), w4.double(), None, 1, 1) + pre.double()
    out4 = hip.conv_up4_h2t(xh[:, :Ct // 8], hip.pack_conv_up4_h2t(w4), hip.empty(B, Cout, 4 * h, 4 * w_), pre_add=hip.to_device(CPU.quads(pre)))
    e = (CPU.quads(out4.cpu(), inverse=True).double() - ref4).abs().max().item() / max(1.0, ref4.abs().max().item())
    if e > 4e-6:
        bad += 1; print("UP4_H2T MISMATCH", (B, Ct, Cout, h, w_), e)
print("fuzz done, mismatches:", bad)
