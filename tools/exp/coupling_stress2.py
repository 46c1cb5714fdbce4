"""Very long single-kernel stress (fixed inputs) of coupling_head and coupling_tail: which of the two has the rare fault?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
g = np.random.Generator(np.random.PCG64(5))
r = lambda *s, scale=1.0: torch.from_numpy((g.standard_normal(s) * scale).astype(np.float32))
def stress(name, fn, pre=None):
    ref = fn().clone(); torch.cuda.synchronize()
    bad, worst, cnt = 0, 0.0, []
    for i in range(N):
        if pre is not None: pre()
        out = fn()
        if not torch.equal(out, ref):
            bad += 1; d_ = (out - ref).abs(); worst = max(worst, float(d_.max())); idx = torch.nonzero(d_ > 0)
            cnt.append((int(idx.shape[0]), idx[0].tolist(), idx[-1].tolist()))
    print("%-56s %3d / %d differ%s %s" % (name, bad, N, "  (max %.2e)" % worst if bad else "", cnt[:4]), flush=True)
for B, C, h2 in ((1, 12, 320), (2, 24, 160)):
    cn, cc2 = C // 2, 2 * (C - C // 2)
    w0, w2 = r(64, cn, 3, 3, scale=0.1), r(64, 64, 1, 1, scale=0.1)
    s0, c0, s2, c2 = r(64, scale=0.1), torch.exp(r(64, scale=0.1)), r(64, scale=0.1), torch.exp(r(64, scale=0.1))
    w4, b4, ps = r(cc2, 64, 3, 3, scale=0.02), r(cc2, scale=0.2), torch.exp(r(cc2, scale=0.2))
    Wm = torch.from_numpy(np.linalg.qr(g.standard_normal((C, C)))[0].astype(np.float32))
    ab, ae = ops.vec(r(C, scale=0.1)), ops.vec(torch.exp(r(C, scale=0.1)))
    wv = ops.vec(Wm)
    z0 = torch.randn(B, C, h2, h2, device="cuda")
    pre, hf = torch.randn(B, 64, h2, h2, device="cuda") * 0.5, torch.randn(B, 2 * C, h2, h2, device="cuda") * 0.5
    hid, zo = ops.empty(B, 64, h2, h2), ops.empty(B, C, h2, h2)
    hpk, tpk = ops.pack_coupling_head(w0, w2, s0, c0, s2, c2), ops.pack_coupling_tail(w4, b4, ps)
    for fmt in (1, 0):
        stress("head B%d C%d %d^2 hid_fmt=%d (output randomised before)" % (B, C, h2, fmt), lambda: ops.coupling_head(z0, hpk, pre, hid, hid_fmt=fmt), pre=lambda: hid.normal_())
        ops.coupling_head(z0, hpk, pre, hid, hid_fmt=fmt); torch.cuda.synchronize()
        stress("tail B%d C%d %d^2 hid_fmt=%d reverse=0" % (B, C, h2, fmt),
               lambda: ops.coupling_tail(hid, tpk, z0, zo, 0, h_ft=hf, w=wv, an_bias=ab, an_escale=ae, hid_fmt=fmt), pre=lambda: zo.normal_())
