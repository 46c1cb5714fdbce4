"""Which kernels give wrong results when they run on a side stream while the 1x1-only coupling_head loops on the main stream?  The product build has no packed-fp32 instructions any more: to reproduce the fault run it against the packed build,
  bash tools/exp/build_pk.sh && BFSR_HIP_LIB=$PWD/tools/exp/libpk.so python tools/exp/victim_probe.py 32
GPU box: python tools/exp/victim_probe.py [B]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps, MODE_BILINEAR_AC, MODE_NEAREST, ACT_LRELU
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ops = HipOps("cuda:0")
g = np.random.Generator(np.random.PCG64(3))
r = lambda *s, scale=1.0: torch.from_numpy((g.standard_normal(s) * scale).astype(np.float32))
# aggressor: coupling_head without a 3x3 stage (the hoisted fFeatures form / the level-3 chain), 64 x 96^2-class shape
w2 = r(64, 64, 1, 1, scale=0.1)
s0, c0, s2, c2 = r(64, scale=0.1), torch.exp(r(64, scale=0.1)), r(64, scale=0.1), torch.exp(r(64, scale=0.1))
hp1 = ops.pack_coupling_head(None, w2, s0, c0, s2, c2)
raw, h2b = torch.randn(B, 64, 96, 96, device="cuda"), ops.h2_empty(B, 64, 96, 96)
rawq = raw.clone()
aggr = {"coupling_head 1x1-only pre_fmt=0": lambda: ops.coupling_head(None, hp1, raw, h2b, pre_fmt=0),
        "coupling_head 1x1-only pre_fmt=1": lambda: ops.coupling_head(None, hp1, rawq, h2b, pre_fmt=1)}
# victims
vict = {}
bottom, cat = torch.randn(B, 256, 48, 48, device="cuda"), ops.empty(B, 512, 96, 96)
vict["resize bilinear-AC 48->96 (256 ch)"] = lambda: ops.resize(bottom, cat[:, 256:], MODE_BILINEAR_AC, 47.0 / 95.0, 47.0 / 95.0, window=(0, 0, 96, 96))
x96, p48 = torch.randn(B, 128, 96, 96, device="cuda"), ops.empty(B, 128, 48, 48)
vict["maxpool2 96->48"] = lambda: ops.maxpool2(x96, p48)
big_a, big_b = torch.randn(B, 64, 192, 192, device="cuda"), ops.empty(B, 64, 192, 192)
vict["axpb_clamp 64 ch @192^2"] = lambda: ops.axpb_clamp(big_a, big_b, 1.5, 0.25)
pw = ops.pack_conv_x3(r(128, 256, 3, 3, scale=0.02), 2)
y96 = ops.empty(B, 128, 96, 96)
x256 = torch.randn(B, 256, 96, 96, device="cuda")
vict["conv_x3 (register-staged) 256->128 @96^2"] = lambda: ops.conv_x3(x256, pw, y96, act=ACT_LRELU)
xh = ops.h2_pack(torch.randn(B, 64, 192, 192, device="cuda"), ops.h2_empty(B, 64, 192, 192))
pwx = ops.pack_conv_x3(r(64, 64, 3, 3, scale=0.04), 1, lazy=True)
yh = ops.h2_empty(B, 64, 192, 192)
vict["conv_h2x 64->64 @192^2"] = lambda: ops.conv_h2x(xh, pwx, yh, act=ACT_LRELU)
f192 = ops.empty(B, 64, 192, 192)
vict["h2_unpack 64 ch @192^2"] = lambda: ops.h2_unpack(xh, f192)
hp = ops.h2_empty(B, 64, 192, 192)
vict["h2_pack 64 ch @192^2"] = lambda: ops.h2_pack(big_a, hp)
pw1 = ops.pack_conv(r(6, 64, 1, 1, scale=0.1), 1)
o6 = ops.empty(B, 6, 192, 192)
vict["conv 1x1 64->6 (fp32 MFMA) @192^2"] = lambda: ops.conv(big_a, pw1, o6)
main, side = torch.cuda.current_stream(), torch.cuda.Stream()
for an, afn in aggr.items():
    afn(); torch.cuda.synchronize()
    print("== aggressor:", an, flush=True)
    for vn, vfn in vict.items():
        ref = vfn().clone(); torch.cuda.synchronize()
        assert torch.equal(ref, vfn()), vn + " alone is not deterministic"
        bad = worst = nel = 0
        for rep in range(5):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(6):
                    out = vfn()
                ev = side.record_event()
            while not ev.query():
                for _ in range(8):
                    afn()
            torch.cuda.synchronize()
            d = (out.float() - ref.float()).abs()
            d = torch.nan_to_num(d, nan=1e30)
            n = int((d > 0).sum())
            bad += n > 0; nel = max(nel, n); worst = max(worst, float(d.max()))
        print("   %-44s wrong in %d of 5 overlapped runs (up to %d elements, max %.1e)" % (vn, bad, nel, worst), flush=True)
