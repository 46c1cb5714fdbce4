"""Which main-stream kernel corrupts the prior's branch 0 when the two run concurrently (8x model, eps0 = [B,6,384,384])?  The victim runs on a side
stream while ONE candidate kernel loops on the main stream; its output is compared with the result computed alone.  (Against the packed build: bash tools/exp/build_pk.sh && BFSR_HIP_LIB=$PWD/tools/exp/libpk.so python tools/exp/aggressor_probe.py 32 -- the product build is clean.)
GPU box: python tools/exp/aggressor_probe.py [B]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd import synth
from bfsr_amd.ops import HipOps, ACT_LRELU, ACT_RELU
from bfsr_amd.srflow import spec
from bfsr_amd.srflow.models import models as registry
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ops = HipOps("cuda:0")
g = np.random.Generator(np.random.PCG64(3))
r = lambda *s, scale=1.0: torch.from_numpy((g.standard_normal(s) * scale).astype(np.float32))
prior = registry.make({"name": "unet", "args": {"depth": 3, "dim": 64, "bilinear": True, "ops": ops}, "sd": synth.state_dict_from_schema(spec.srflow_prior_schema(), 4321)}, load_sd=True).eval()
pe = prior.engine()
n0 = torch.randn(B, 6, 384, 384, device="cuda")
o0 = ops.empty(B, 6, 384, 384)
pe.forward_branch(0, n0, out=o0); torch.cuda.synchronize()
ref = pe.forward_branch(0, n0, out=o0).clone(); torch.cuda.synchronize()
assert torch.equal(ref, pe.forward_branch(0, n0, out=o0)), "victim alone is not deterministic"
side = torch.cuda.Stream()

# ---- candidate aggressors (shapes of the 8x model at this batch) ---------------------------------------------------------------
cands = {}
C, hh = 24, 192
cn, cc2 = C // 2, 2 * (C - C // 2)
w0, w2 = r(64, cn, 3, 3, scale=0.1), r(64, 64, 1, 1, scale=0.1)
s0, c0, s2, c2 = r(64, scale=0.1), torch.exp(r(64, scale=0.1)), r(64, scale=0.1), torch.exp(r(64, scale=0.1))
w4, b4, ps = r(cc2, 64, 3, 3, scale=0.02), r(cc2, scale=0.2), torch.exp(r(cc2, scale=0.2))
z, pre, hf = torch.randn(B, C, hh, hh, device="cuda"), torch.randn(B, 64, hh, hh, device="cuda") * 0.5, torch.randn(B, 2 * C, hh, hh, device="cuda") * 0.5
hid, zo = ops.h2_empty(B, 64, hh, hh), ops.empty(B, C, hh, hh)
hpk, tpk = ops.pack_coupling_head(w0, w2, s0, c0, s2, c2), ops.pack_coupling_tail(w4, b4, ps)
ops.coupling_head(z, hpk, pre, hid)
cands["coupling_head C=24 @192^2"] = lambda: ops.coupling_head(z, hpk, pre, hid)
cands["coupling_tail C=24 @192^2"] = lambda: ops.coupling_tail(hid, tpk, z, zo, 1, h_ft=hf)
z3 = torch.randn(B, 96, 96, 96, device="cuda")
z1h = ops.h2_pack(z3[:, :48].contiguous(), ops.h2_empty(B, 48, 96, 96))
p0 = ops.pack_conv_x3(r(64, 48, 3, 3, scale=0.05), 2)
raw = ops.empty(B, 64, 96, 96)
preh = ops.h2_pack(torch.randn(B, 64, 96, 96, device="cuda") * 0.5, ops.h2_empty(B, 64, 96, 96))
cands["h2_pack z1 @96^2"] = lambda: ops.h2_pack(z3[:, :48], z1h)
cands["conv_h2x 48->64 @96^2 (+ h2 residual)"] = lambda: ops.conv_h2x(z1h, p0, raw, res1=preh, alpha1=1.0)
hp1 = ops.pack_coupling_head(None, w2, s0, c0, s2, c2)
h2b = ops.h2_empty(B, 64, 96, 96)
cands["coupling_head 1x1-only @96^2"] = lambda: ops.coupling_head(None, hp1, raw, h2b, pre_fmt=0)
p4 = ops.pack_conv_x3(r(96, 64, 3, 3, scale=0.02), 1, lazy=True)
ha, e4 = ops.empty(B, 96, 96, 96), ops.pack_epilogue(96, bias=r(96, scale=0.1))
cands["conv_h2x 64->96 @96^2"] = lambda: ops.conv_h2x(h2b, p4, ha, epi=e4)
Wm = torch.from_numpy(np.linalg.qr(g.standard_normal((96, 96)))[0].astype(np.float32))
hf3, zo3 = torch.randn(B, 192, 96, 96, device="cuda") * 0.5, ops.empty(B, 96, 96, 96)
ab, ae = ops.vec(r(96, scale=0.1)), ops.vec(torch.exp(r(96, scale=0.1)))
wv, wtv = ops.vec(Wm), ops.vec(Wm.t().contiguous())
cands["flow_pointwise_mfma C=96 @96^2"] = lambda: ops.flow_pointwise(z3, zo3, True, h_aff=ha, h_ft=hf3, w=wv, wt=wtv, an_bias=ab, an_escale=ae)
big_a, big_b = torch.randn(B, 64, 192, 192, device="cuda"), torch.empty(B, 64, 192, 192, device="cuda")
cands["axpb_clamp (HBM-bound control)"] = lambda: ops.axpb_clamp(big_a, big_b, 1.0, 0.0)
sq = ops.empty(B, 96, 96, 96)
cands["squeeze2d 24 @192^2"] = lambda: ops.squeeze2d(z, sq)

main = torch.cuda.current_stream()
for name, fn in cands.items():
    fn(); torch.cuda.synchronize()
    bad = worst = 0
    for rep in range(6):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            out = pe.forward_branch(0, n0, out=o0)
            ev = side.record_event()
        while not ev.query():
            for _ in range(8):
                fn()
        torch.cuda.synchronize()
        d = (out - ref).abs()
        n = int((d > 0).sum())
        bad += n > 0
        worst = max(worst, float(d.max()))
    print("%-42s victim differs in %d of 6 overlapped runs (max %.1e)" % (name, bad, worst), flush=True)
