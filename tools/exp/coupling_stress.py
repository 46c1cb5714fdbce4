"""Long determinism stress of the coupling_head / coupling_tail pair in the configurations the ENGINE uses (octet-major hid,
both directions, in-place z, full pointwise chain with the invertible 1x1 and ActNorm vectors): python tools/exp/coupling_stress.py [N]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
g = np.random.Generator(np.random.PCG64(5))
r = lambda *s, scale=1.0: torch.from_numpy((g.standard_normal(s) * scale).astype(np.float32))

def stress(name, fn, clone_in=None):
    ref = fn().clone(); torch.cuda.synchronize()
    bad, worst = 0, 0.0
    for _ in range(N):
        out = fn()
        if not torch.equal(out, ref):
            bad += 1; worst = max(worst, float((out - ref).abs().max()))
    print("%-64s %4d / %d differ%s" % (name, bad, N, "  (max %.2e)" % worst if bad else ""), flush=True)
    return bad

tot = 0
ONLY_PAIR = len(sys.argv) > 2
for B, C, h2 in ((1, 12, 320), (2, 12, 320), (2, 24, 160)):
    cn, cc2 = C // 2, 2 * (C - C // 2)
    w0, w2 = r(64, cn, 3, 3, scale=0.1), r(64, 64, 1, 1, scale=0.1)
    s0, c0, s2, c2 = r(64, scale=0.1), torch.exp(r(64, scale=0.1)), r(64, scale=0.1), torch.exp(r(64, scale=0.1))
    w4, b4, ps = r(cc2, 64, 3, 3, scale=0.02), r(cc2, scale=0.2), torch.exp(r(cc2, scale=0.2))
    Wm = torch.from_numpy(np.linalg.qr(g.standard_normal((C, C)))[0].astype(np.float32))
    ab, ae = ops.vec(r(C, scale=0.1)), ops.vec(torch.exp(r(C, scale=0.1)))
    wv, wtv = ops.vec(Wm), ops.vec(Wm.t().contiguous())
    z0 = torch.randn(B, C, h2, h2, device="cuda")
    pre, hf = torch.randn(B, 64, h2, h2, device="cuda") * 0.5, torch.randn(B, 2 * C, h2, h2, device="cuda") * 0.5
    hid, zo, z = ops.empty(B, 64, h2, h2), ops.empty(B, C, h2, h2), ops.empty(B, C, h2, h2)
    hpk, tpk = ops.pack_coupling_head(w0, w2, s0, c0, s2, c2), ops.pack_coupling_tail(w4, b4, ps)
    for fmt in (1, 0):
        if not ONLY_PAIR:
            tot += stress("head B%d C%d %d^2 hid_fmt=%d" % (B, C, h2, fmt), lambda: ops.coupling_head(z0, hpk, pre, hid, hid_fmt=fmt))
        ops.coupling_head(z0, hpk, pre, hid, hid_fmt=fmt)
        for rev in (0, 1):
            if not ONLY_PAIR:
                tot += stress("tail B%d C%d %d^2 hid_fmt=%d reverse=%d (out of place)" % (B, C, h2, fmt, rev),
                              lambda: ops.coupling_tail(hid, tpk, z0, zo, rev, h_ft=hf, w=wv if rev == 0 else wtv, an_bias=ab, an_escale=ae, hid_fmt=fmt))
            def pair():
                z.copy_(z0)
                hid.normal_()                                            # stale contents differ from launch to launch, as in the engine
                ops.coupling_head(z, hpk, pre, hid, hid_fmt=fmt)
                return ops.coupling_tail(hid, tpk, z, z, rev, h_ft=hf, w=wv if rev == 0 else wtv, an_bias=ab, an_escale=ae, hid_fmt=fmt)
            tot += stress("head+tail pair in place B%d C%d %d^2 hid_fmt=%d reverse=%d" % (B, C, h2, fmt, rev), pair)
print("TOTAL", tot)
