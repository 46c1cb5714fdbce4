"""Run-to-run determinism of the MFMA kernels under load: every kernel is launched N times on the same inputs and each result is
compared BITWISE with the first.  Motivation (round 3): the fused step kernel (tools/exp/kernels/coupling_step.hip) produced wrong columns in ~1 of 3 launches until its
MFMA operand registers were kept allocated for a chunk behind their last use (a dead SrcA/SrcB register that hipcc recycles at
once can be overwritten before a queued MFMA has read it when two waves share a SIMD's matrix pipe); such a fault is
intermittent and bf16-sized, i.e. a single parity run can miss it.  Usage (GPU box): python tools/determinism_stress.py [N] [noise]
`noise`: a second stream keeps the chip busy with other MFMA kernels while each kernel is stressed (the engine's side stream does
that to the flow steps): co-resident waves of ANOTHER kernel change what a wave's matrix pipe is doing behind its back."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bfsr_amd.ops import ACT_LRELU, ACT_RELU, HipOps  # noqa: E402

ops = HipOps("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
NOISE = len(sys.argv) > 2 and sys.argv[2] == "noise"
_side = torch.cuda.Stream() if NOISE else None
_noise = {}


def _noise_burst():
    """Enqueue ~10 ms of unrelated MFMA work on the side stream (small launches that share CUs with the kernel under test)."""
    if not _noise:
        x = torch.randn(4, 64, 96, 96, device="cuda")
        _noise["x"] = x
        _noise["pw"] = ops.pack_conv_x3(torch.randn(64, 64, 3, 3) * 0.05, 1)
        _noise["y"] = ops.empty(4, 64, 96, 96)
        _noise["pw32"] = ops.pack_conv(torch.randn(64, 64, 1, 1) * 0.1, 2)
    with torch.cuda.stream(_side):
        for _ in range(40):
            ops.conv_x3(_noise["x"], _noise["pw"], _noise["y"])
            ops.conv(_noise["x"], _noise["pw32"], _noise["y"])
g = np.random.Generator(np.random.PCG64(5))
r = lambda *s, scale=1.0: torch.from_numpy((g.standard_normal(s) * scale).astype(np.float32))
d = ops.to_device


def stress(name, fn):
    ref = fn().clone()
    torch.cuda.synchronize()
    bad = 0
    worst = 0.0
    for i in range(N):
        if NOISE and i % 4 == 0:
            _noise_burst()
        out = fn()
        if not torch.equal(out, ref):
            bad += 1
            worst = max(worst, float((out.float() - ref.float()).abs().max()))
    print("%-52s %4d / %d launches differ from the first%s" % (name, bad, N, "  (max |diff| %.3e)" % worst if bad else ""), flush=True)
    return bad


def main():
    total = 0
    B, hw = 8, 160
    # --- RRDB dense-block convs on split tensors: conv_h2x (h2 tensors, the default split) and conv_x3s (x3 tensors, BFSR_SPLIT=bf16x3)
    default_split = ops.split
    for split, name in (("f16x2", "conv_h2x"), ("bf16x3", "conv_x3s")):
        ops.split = split
        for Cin, Cout in ((64, 32), (128, 32), (192, 64)):
            x3 = ops.x3_pack(torch.randn(B, Cin, hw, hw, device="cuda"), ops.x3_empty(B, Cin, hw, hw))
            pw = ops.pack_conv_x3(r(Cout, Cin, 3, 3, scale=0.05), 1)
            epi = ops.pack_epilogue(Cout, bias=r(Cout, scale=0.1))
            y3 = ops.x3_empty(B, Cout, hw, hw)
            total += stress("%s %d->%d @%dx%d" % (name, Cin, Cout, hw, hw), lambda: ops.conv_x3s(x3, pw, y3, epi=epi, act=ACT_LRELU))
    ops.split = default_split
    # --- the quad-major fp32 outputs (16-byte buffer stores: the store-data hazard of launch_util.h's store_b128) and the Cout <= 32 ring kernel
    xh = ops.h2_pack(torch.randn(B, 64, hw, hw, device="cuda"), ops.h2_empty(B, 64, hw, hw))
    pw, epi, yq = ops.pack_conv_x3(r(64, 64, 3, 3, scale=0.05), 1), ops.pack_epilogue(64, bias=r(64, scale=0.1)), ops.empty(B, 64, hw, hw)
    total += stress("conv_h2x 64->64 quad-major fp32 out @%dx%d" % (hw, hw), lambda: ops.conv_h2x(xh, pw, yq, epi=epi, y_fmt=1))
    pr, er, yr = ops.pack_coupling_tail(r(24, 64, 3, 3, scale=0.05), r(24, scale=0.1), torch.ones(24)), ops.pack_epilogue(24, bias=r(24, scale=0.1)), ops.empty(B, 24, hw, hw)
    total += stress("conv_h2r 64->24 quad-major fp32 out @%dx%d" % (hw, hw), lambda: ops.conv_h2r(xh, pr, yr, epi=er, y_fmt=1))
    # --- fp16 dense-block convs on h2 tensors (conv_h2s)
    for Cin, Cout in ((64, 32), (192, 64)):
        xh = ops.h2_pack(torch.randn(B, Cin, 128, 128, device="cuda"), ops.h2_empty(B, Cin, 128, 128))
        pw = ops.pack_conv_h2s(r(Cout, Cin, 3, 3, scale=0.05))
        epi = ops.pack_epilogue(Cout, bias=r(Cout, scale=0.1))
        yh = ops.h2_empty(B, Cout, 128, 128)
        total += stress("conv_h2s %d->%d @128x128" % (Cin, Cout), lambda: ops.conv_h2s(xh, pw, yh, epi=epi, act=ACT_LRELU))
    # --- register-staged 3xBF16 convs (hoists) and the parity-decomposed taps kernel
    x = torch.randn(B, 64, hw, hw, device="cuda")
    pw = ops.pack_conv_x3(r(256, 64, 3, 3, scale=0.05), 2)
    y = ops.empty(B, 256, hw, hw)
    total += stress("conv_bf16x3 64->256 @%dx%d" % (hw, hw), lambda: ops.conv_x3(x, pw, y))
    taps = torch.randn(B, 256, hw // 2, hw // 2, device="cuda")
    pu = ops.pack_conv_up2_x3(r(256, 256, 3, 3, scale=0.02))
    total += stress("conv_up2_bf16x3 256->256 @%dx%d" % (hw, hw), lambda: ops.conv_up2_x3(taps, pu, y))
    # --- the coupled FlowStep remainder: head + tail pair, both levels
    for C, h2 in ((12, 160), (24, 96)):
        cn, cc2 = C // 2, 2 * (C - C // 2)
        w0, w2 = r(64, cn, 3, 3, scale=0.1), r(64, 64, 1, 1, scale=0.1)
        s0, c0, s2, c2 = r(64, scale=0.1), torch.exp(r(64, scale=0.1)), r(64, scale=0.1), torch.exp(r(64, scale=0.1))
        w4, b4, ps = r(cc2, 64, 3, 3, scale=0.02), r(cc2, scale=0.2), torch.exp(r(cc2, scale=0.2))
        z, pre, hf = torch.randn(B, C, h2, h2, device="cuda"), torch.randn(B, 64, h2, h2, device="cuda") * 0.5, torch.randn(B, 2 * C, h2, h2, device="cuda") * 0.5
        hid, zo = ops.h2_empty(B, 64, h2, h2), ops.empty(B, C, h2, h2)
        hpk, tpk = ops.pack_coupling_head(w0, w2, s0, c0, s2, c2), ops.pack_coupling_tail(w4, b4, ps)
        total += stress("coupling_head C=%d @%dx%d" % (C, h2, h2), lambda: ops.coupling_head(z, hpk, pre, hid))
        total += stress("coupling_tail C=%d @%dx%d" % (C, h2, h2), lambda: ops.coupling_tail(hid, tpk, z, zo, 1, h_ft=hf))
    # --- level 3 (C = 96): fused 3x3 -> 1x1 on the fp32 MFMA, Conv2dZeros 64 -> 96, and the MFMA pointwise kernel
    z = torch.randn(B, 96, 80, 80, device="cuda")
    pre3, hid3 = torch.randn(B, 64, 80, 80, device="cuda") * 0.5, ops.empty(B, 64, 80, 80)
    pa = ops.pack_conv(r(64, 48, 3, 3, scale=0.1), 2)
    pb = ops.pack_conv(r(64, 64, 1, 1, scale=0.1), 2)
    ea, eb = ops.pack_epilogue(64, aff_shift=r(64, scale=0.1), aff_scale=torch.exp(r(64, scale=0.1))), ops.pack_epilogue(64, aff_shift=r(64, scale=0.1), aff_scale=torch.exp(r(64, scale=0.1)))
    total += stress("conv 3x3 48->64 + fused 1x1 (fp32 MFMA) @80x80", lambda: ops.conv(z[:, :48], pa, hid3, epi=ea, pre_add=pre3, act=ACT_RELU, stage2=(pb, eb, ACT_RELU)))
    pc = ops.pack_conv_x3(r(96, 64, 3, 3, scale=0.02), 1)
    ha = ops.empty(B, 96, 80, 80)
    ec = ops.pack_epilogue(96, bias=r(96, scale=0.1))
    total += stress("conv_bf16x3 64->96 @80x80", lambda: ops.conv_x3(hid3, pc, ha, epi=ec))
    Wm = torch.from_numpy(np.linalg.qr(g.standard_normal((96, 96)))[0].astype(np.float32))
    hf3, zo3 = torch.randn(B, 192, 80, 80, device="cuda") * 0.5, ops.empty(B, 96, 80, 80)
    ab, ae = ops.vec(r(96, scale=0.1)), ops.vec(torch.exp(r(96, scale=0.1)))
    total += stress("flow_pointwise_mfma C=96 @80x80", lambda: ops.flow_pointwise(z, zo3, True, h_aff=ha, h_ft=hf3, w=ops.vec(Wm), wt=ops.vec(Wm.t().contiguous()), an_bias=ab, an_escale=ae))
    # --- the LDS-DMA taps kernels (x2 with key chunks, x4) and a dense-block chain in one persistent launch
    th = ops.h2_pack(torch.randn(B, 256, 80, 80, device="cuda"), ops.h2_empty(B, 256, 80, 80))
    tk = ops.h2_empty(B, 256 + 4 * 64, 80, 80)
    ops.h2_pack(torch.randn(B, 256, 80, 80, device="cuda"), tk[:, :32])
    ops.h2_pack_s2d(torch.randn(B, 64, 160, 160, device="cuda"), tk[:, 32:])
    p2 = ops.pack_conv_up2_h2t(r(128, 256, 3, 3, scale=0.02), r(128, 64, 3, 3, scale=0.03))
    y2 = ops.empty(B, 128, 160, 160)
    total += stress("conv_up2_h2t 256+64->128 @160x160", lambda: ops.conv_up2_h2t(tk, p2, y2))
    p4h = ops.pack_conv_up4_h2t(r(128, 256, 3, 3, scale=0.02))
    y4h = ops.zeros(B, 128, 320, 320)
    total += stress("conv_up4_h2t 256->128 @320x320", lambda: ops.conv_up4_h2t(th, p4h, y4h))
    D = ops.h2_empty(B, 192, 96, 96)
    ops.h2_pack(torch.randn(B, 64, 96, 96, device="cuda") * 0.5, D[:, :8])
    specs = []
    for i in range(4):
        specs.append(dict(x=D[:, :8 + 4 * i], pw=ops.pack_conv_x3(r(32, 64 + 32 * i, 3, 3, scale=0.03), 1, lazy=True), out=D[:, 8 + 4 * i: 12 + 4 * i],
                          epi=ops.pack_epilogue(32, bias=r(32, scale=0.1)), act=ACT_LRELU, slope=0.2))
    yc = ops.h2_empty(B, 64, 96, 96)
    specs.append(dict(x=D, pw=ops.pack_conv_x3(r(64, 192, 3, 3, scale=0.03), 1, lazy=True), out=yc, epi=ops.pack_epilogue(64, bias=r(64, scale=0.1)),
                      res1=D[:, :8], alpha1=1.0))
    ch = ops.conv_chain(specs)
    total += stress("conv_chain (one dense block, 5 convs) @96x96", lambda: (ch.run(), yc)[1])
    # --- x4 taps kernel (8x model) and the fp16 conv of the LINF fp16 path
    t4 = torch.randn(B, 256, 48, 48, device="cuda")
    p4 = ops.pack_conv_up4_x3(r(128, 256, 3, 3, scale=0.02))
    y4 = ops.empty(B, 128, 192, 192)
    total += stress("conv_up4_bf16x3 256->128 @192x192", lambda: ops.conv_up4_x3(t4, p4, y4))
    pf = ops.pack_conv_f16(r(128, 64, 3, 3, scale=0.05), 2)
    yf = ops.empty(B, 128, hw, hw)
    total += stress("conv_f16 64->128 @%dx%d" % (hw, hw), lambda: ops.conv_f16(x, pf, yf))
    # --- LINF conditioning: Fourier features + MLP fused (x3 and fp16)
    import oracle.linf_ref as O                      # input preparation only (coords / cells of a 3x3-patch query grid)
    prep = O.batch_prep(torch.rand(2, 3, 48, 48), (4 * 48, 4 * 48))
    coord, cell = d(prep["coord"]), d(prep["cell"])
    qh, qw = coord.shape[1:3]
    cf = torch.randn(2, 512, 48, 48, device="cuda")
    ws_ = [r(256, 1024, scale=0.03), r(256, 256, scale=0.06), r(256, 256, scale=0.06), r(540, 256, scale=0.06)]
    bs_ = [r(256, scale=0.1), r(256, scale=0.1), r(256, scale=0.1), r(540, scale=0.1)]
    phase = ops.vec(r(128, 2, scale=0.5))
    for x3 in (True, False):
        pk = ops.pack_linf_mlp(ws_, bs_, x3=x3)
        om = ops.empty(2, 540, qh, qw)
        total += stress("linf_mlp %s %dx%d queries" % ("x3" if x3 else "fp16", qh, qw), lambda: ops.linf_mlp(cf, coord, cell, phase, pk, om, 256, x3=x3))
    print("TOTAL differing launches: %d" % total)
    return total


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
