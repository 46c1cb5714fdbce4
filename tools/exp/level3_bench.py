"""Level-3 coupled step (C = 96, 8 x 80^2 at config 2; 64 x 96^2 at config 4): the current launches (fused 3x3 -> 1x1 on the fp32 MFMA, Conv2dZeros on the
register-staged split conv) against a chain on the fp16-split kernels that exist (split 3x3 48->64 raw + pre_add, the 1x1-only coupling_head,
conv_h2x 64->96 over the h2 hid).  GPU box: python tools/exp/level3_bench.py [B hw]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps, ACT_RELU
ops = HipOps("cuda:0")
B, hw = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 80)
g = np.random.Generator(np.random.PCG64(1))
r = lambda *s, scale=1.0: torch.from_numpy((g.standard_normal(s) * scale).astype(np.float32))
def timed(f, n=50):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
z = torch.randn(B, 96, hw, hw, device="cuda")
pre = torch.randn(B, 64, hw, hw, device="cuda") * 0.5
w0, w2, w4 = r(64, 48, 3, 3, scale=0.1), r(64, 64, 1, 1, scale=0.1), r(96, 64, 3, 3, scale=0.02)
s0, c0, s2, c2 = r(64, scale=0.1), torch.exp(r(64, scale=0.1)), r(64, scale=0.1), torch.exp(r(64, scale=0.1))
b4, ps = r(96, scale=0.2), torch.exp(r(96, scale=0.2))
hid, haff = ops.empty(B, 64, hw, hw), ops.empty(B, 96, hw, hw)
pa, pb = ops.pack_conv(w0, 2), ops.pack_conv(w2, 2)
ea, eb = ops.pack_epilogue(64, aff_shift=s0, aff_scale=c0), ops.pack_epilogue(64, aff_shift=s2, aff_scale=c2)
t_a = timed(lambda: ops.conv(z[:, :48], pa, hid, epi=ea, pre_add=pre, act=ACT_RELU, stage2=(pb, eb, ACT_RELU)))
pc, ec = ops.pack_conv_x3(w4, 1), ops.pack_epilogue(96, bias=b4, post_scale=ps)
t_b = timed(lambda: ops.conv_x3(hid, pc, haff, epi=ec))
print("current: conv 3x3 48->64 + fused 1x1 (fp32 MFMA) %.1f us, Conv2dZeros 64->96 (conv_f16x2) %.1f us = %.1f us" % (t_a, t_b, t_a + t_b), flush=True)
raw = ops.empty(B, 64, hw, hw)
p0 = ops.pack_conv_x3(w0, 2)
t1 = timed(lambda: ops.conv_x3(z[:, :48], p0, raw, pre_add=pre))
hp = ops.pack_coupling_head(None, w2, s0, c0, s2, c2)
h2 = ops.h2_empty(B, 64, hw, hw)
t2 = timed(lambda: ops.coupling_head(None, hp, raw, h2, pre_fmt=0))
t3 = timed(lambda: ops.conv_h2x(h2, pc, haff, epi=ec))
print("chain:   conv_x3 48->64 raw + pre_add %.1f us, 1x1-only head %.1f us, conv_h2x 64->96 %.1f us = %.1f us" % (t1, t2, t3, t1 + t2 + t3), flush=True)
# fFeatures of the level: 1x1 (native fp32) + Conv2dZeros 64->192 against 1x1-only head + conv_h2x 64->192
w4f, b4f, psf = r(192, 64, 3, 3, scale=0.02), r(192, scale=0.2), torch.exp(r(192, scale=0.2))
hf = ops.empty(B, 192, hw, hw)
pf1 = ops.pack_conv(w2, 2)
t_f1 = timed(lambda: ops.conv(hid, pf1, hid, epi=eb, act=ACT_RELU))
pf, ef = ops.pack_conv_x3(w4f, 1), ops.pack_epilogue(192, bias=b4f, post_scale=psf)
t_f2 = timed(lambda: ops.conv_x3(hid, pf, hf, epi=ef))
t_f3 = timed(lambda: ops.conv_h2x(h2, pf, hf, epi=ef))
print("fFeatures: 1x1 (fp32 MFMA) %.1f + conv_f16x2 64->192 %.1f = %.1f us; 1x1-only head %.1f + conv_h2x 64->192 %.1f = %.1f us" % (t_f1, t_f2, t_f1 + t_f2, t2, t_f3, t2 + t_f3), flush=True)
