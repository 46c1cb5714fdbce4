"""Which ingredient of the head -> tail sequence produces the rare mismatch?  Variants of the pair on fixed inputs, stale `hid` randomised."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
g = np.random.Generator(np.random.PCG64(5))
r = lambda *s, scale=1.0: torch.from_numpy((g.standard_normal(s) * scale).astype(np.float32))
def stress(name, fn):
    ref = fn().clone(); torch.cuda.synchronize()
    bad, info = 0, []
    for i in range(N):
        out = fn()
        if not torch.equal(out, ref):
            bad += 1; d_ = (out - ref).abs(); idx = torch.nonzero(d_ > 0)
            info.append("n=%d max=%.1e first=%s last=%s" % (idx.shape[0], float(d_.max()), idx[0].tolist(), idx[-1].tolist()))
    print("%-60s %3d / %d differ  %s" % (name, bad, N, " | ".join(info[:3])), flush=True)
B, C, h2 = 1, 12, 320
cn, cc2 = C // 2, 2 * (C - C // 2)
w0, w2 = r(64, cn, 3, 3, scale=0.1), r(64, 64, 1, 1, scale=0.1)
s0, c0, s2, c2 = r(64, scale=0.1), torch.exp(r(64, scale=0.1)), r(64, scale=0.1), torch.exp(r(64, scale=0.1))
w4, b4, ps = r(cc2, 64, 3, 3, scale=0.02), r(cc2, scale=0.2), torch.exp(r(cc2, scale=0.2))
Wm = torch.from_numpy(np.linalg.qr(g.standard_normal((C, C)))[0].astype(np.float32))
ab, ae = ops.vec(r(C, scale=0.1)), ops.vec(torch.exp(r(C, scale=0.1)))
wv = ops.vec(Wm)
z0 = torch.randn(B, C, h2, h2, device="cuda")
pre, hf = torch.randn(B, 64, h2, h2, device="cuda") * 0.5, torch.randn(B, 2 * C, h2, h2, device="cuda") * 0.5
hid, zo, z = ops.empty(B, 64, h2, h2), ops.empty(B, C, h2, h2), ops.empty(B, C, h2, h2)
hpk, tpk = ops.pack_coupling_head(w0, w2, s0, c0, s2, c2), ops.pack_coupling_tail(w4, b4, ps)
tail = lambda zi, zo_, fmt: ops.coupling_tail(hid, tpk, zi, zo_, 0, h_ft=hf, w=wv, an_bias=ab, an_escale=ae, hid_fmt=fmt)
for fmt in (1, 0):
    def base():
        z.copy_(z0); hid.normal_(); ops.coupling_head(z, hpk, pre, hid, hid_fmt=fmt); return tail(z, z, fmt)
    def outplace():
        hid.normal_(); ops.coupling_head(z0, hpk, pre, hid, hid_fmt=fmt); return tail(z0, zo, fmt)
    def synced():
        z.copy_(z0); hid.normal_(); ops.coupling_head(z, hpk, pre, hid, hid_fmt=fmt); torch.cuda.synchronize(); return tail(z, z, fmt)
    def hid_only():
        hid.normal_(); return ops.coupling_head(z0, hpk, pre, hid, hid_fmt=fmt)
    def no_rand():
        z.copy_(z0); ops.coupling_head(z, hpk, pre, hid, hid_fmt=fmt); return tail(z, z, fmt)
    stress("fmt=%d  in place, stale hid randomised (engine-like)" % fmt, base)
    stress("fmt=%d  out of place" % fmt, outplace)
    stress("fmt=%d  device sync between head and tail" % fmt, synced)
    stress("fmt=%d  head only, hid compared" % fmt, hid_only)
    stress("fmt=%d  in place, hid NOT randomised" % fmt, no_rand)
