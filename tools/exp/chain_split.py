"""Would two half-batch lanes on two streams speed up the sequential coupled-step chain?  16 x (coupling_head -> coupling_tail) at the level-1 / level-2
shapes of config 2 and the level-3 chain (split 3x3 raw -> 1x1-only head -> conv_h2x 64->96 -> pointwise), one stream with B = 8 against two
alternately enqueued lanes of B = 4.  GPU box: python tools/exp/chain_split.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
g = np.random.Generator(np.random.PCG64(1))
r = lambda *s, scale=1.0: torch.from_numpy((g.standard_normal(s) * scale).astype(np.float32))
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
def timed(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
def lanes(gen):
    side.wait_stream(main)
    ga, gb = gen(0, B // 2), gen(B // 2, B)
    live = True
    while live:
        live = next(ga, False) is not False
        with torch.cuda.stream(side):
            live = (next(gb, False) is not False) or live
    main.wait_stream(side)
B = 8
for C, hw in ((12, 320), (24, 160)):
    cn, cc2 = C // 2, 2 * (C - C // 2)
    w0, w2 = r(64, cn, 3, 3, scale=0.1), r(64, 64, 1, 1, scale=0.1)
    s0, c0, s2, c2 = r(64, scale=0.1), torch.exp(r(64, scale=0.1)), r(64, scale=0.1), torch.exp(r(64, scale=0.1))
    w4, b4, ps = r(cc2, 64, 3, 3, scale=0.02), r(cc2, scale=0.2), torch.exp(r(cc2, scale=0.2))
    Wm = torch.from_numpy(np.linalg.qr(g.standard_normal((C, C)))[0].astype(np.float32))
    bias, es, wv = ops.vec(r(C, scale=0.1)), ops.vec(torch.exp(r(C, scale=0.1))), ops.vec(Wm)
    z = torch.randn(B, C, hw, hw, device="cuda")
    pre = torch.randn(B, 16 * 64, hw, hw, device="cuda") * 0.5
    hf = torch.randn(B, 16 * 2 * C, hw, hw, device="cuda") * 0.5
    hid = ops.h2_empty(B, 64, hw, hw)
    hpk, tpk = ops.pack_coupling_head(w0, w2, s0, c0, s2, c2), ops.pack_coupling_tail(w4, b4, ps)
    def chain(b0, b1):
        zz, hh = z[b0:b1], hid[b0:b1]
        for k in range(16):
            ops.coupling_head(zz, hpk, pre[b0:b1, 64 * k: 64 * (k + 1)], hh, pre_fmt=1)
            yield
            ops.coupling_tail(hh, tpk, zz, zz, 1, h_ft=hf[b0:b1, 2 * C * k: 2 * C * (k + 1)], w=wv, an_bias=bias, an_escale=es, h_ft_fmt=1)
            yield
    t1 = timed(lambda: [None for _ in chain(0, B)])
    t2 = timed(lambda: lanes(chain))
    print("C=%d %dx%d: 16 steps one stream %.1f us/step, two lanes %.1f us/step" % (C, hw, hw, t1 / 16, t2 / 16), flush=True)
# level 3
C, hw = 96, 80
w0, w2, w4 = r(64, 48, 3, 3, scale=0.1), r(64, 64, 1, 1, scale=0.1), r(96, 64, 3, 3, scale=0.02)
s0, c0, s2, c2 = r(64, scale=0.1), torch.exp(r(64, scale=0.1)), r(64, scale=0.1), torch.exp(r(64, scale=0.1))
b4, ps = r(96, scale=0.2), torch.exp(r(96, scale=0.2))
Wm = torch.from_numpy(np.linalg.qr(g.standard_normal((C, C)))[0].astype(np.float32))
wv, wt = ops.vec(Wm), ops.vec(Wm.t().contiguous())
ab, ae = ops.vec(r(C, scale=0.1)), ops.vec(torch.exp(r(C, scale=0.1)))
z = torch.randn(B, C, hw, hw, device="cuda")
pre = torch.randn(B, 16 * 64, hw, hw, device="cuda") * 0.5
hf = torch.randn(B, 16 * 2 * C, hw, hw, device="cuda") * 0.5
raw, haff, h2 = ops.empty(B, 64, hw, hw), ops.empty(B, 96, hw, hw), ops.h2_empty(B, 64, hw, hw)
p0, hp = ops.pack_conv_x3(w0, 2), ops.pack_coupling_head(None, w2, s0, c0, s2, c2)
p4, e4 = ops.pack_conv_x3(w4, 1, lazy=True), ops.pack_epilogue(96, bias=b4, post_scale=ps)
def chain3(b0, b1):
    zz = z[b0:b1]
    for k in range(16):
        ops.conv_x3(zz[:, :48], p0, raw[b0:b1], pre_add=pre[b0:b1, 64 * k: 64 * (k + 1)])
        yield
        ops.coupling_head(None, hp, raw[b0:b1], h2[b0:b1], pre_fmt=0)
        yield
        ops.conv_h2x(h2[b0:b1], p4, haff[b0:b1], epi=e4)
        yield
        ops.flow_pointwise(zz, zz, True, h_aff=haff[b0:b1], h_ft=hf[b0:b1, 2 * C * k: 2 * C * (k + 1)], w=wv, wt=wt, an_bias=ab, an_escale=ae)
        yield
t1 = timed(lambda: [None for _ in chain3(0, B)])
t2 = timed(lambda: lanes(chain3))
print("C=96 80x80: 16 steps one stream %.1f us/step, two lanes %.1f us/step" % (t1 / 16, t2 / 16), flush=True)
