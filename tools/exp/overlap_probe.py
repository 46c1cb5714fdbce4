"""How well do a few long full-chip launches on one stream overlap with many short under-filled launches on another?  Side stream: 2 x conv_up2_h2t (config-2
level-1 hoists, ~6.3 ms each); main stream: 16 level-3 coupled steps (4 short launches each).  Enqueue orders: side first / main first / side launches spread
between the main stream's steps.  GPU box: python tools/exp/overlap_probe.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
g = np.random.Generator(np.random.PCG64(1))
r = lambda *s, scale=1.0: torch.from_numpy((g.standard_normal(s) * scale).astype(np.float32))
B = 8
# side work
wt = r(1024, 320, 3, 3, scale=0.02)
xh = ops.h2_empty(B, 256 + 256, 160, 160)
ops.h2_pack(torch.randn(B, 256, 160, 160, device="cuda"), xh[:, :32]); ops.h2_pack_s2d(torch.randn(B, 64, 320, 320, device="cuda"), xh[:, 32:])
pk = ops.pack_conv_up2_h2t(wt[:, 64:].contiguous(), wt[:, :64].contiguous())
big = [ops.empty(B, 1024, 320, 320) for _ in range(2)]
def side_work():
    for o in big:
        ops.conv_up2_h2t(xh, pk, o)
        yield
# main work: level-3 chain
C, hw = 96, 80
w0, w2, w4 = r(64, 48, 3, 3, scale=0.1), r(64, 64, 1, 1, scale=0.1), r(96, 64, 3, 3, scale=0.02)
s0, c0, s2, c2 = r(64, scale=0.1), torch.exp(r(64, scale=0.1)), r(64, scale=0.1), torch.exp(r(64, scale=0.1))
Wm = torch.from_numpy(np.linalg.qr(g.standard_normal((C, C)))[0].astype(np.float32))
wv, wtt = ops.vec(Wm), ops.vec(Wm.t().contiguous())
ab, ae = ops.vec(r(C, scale=0.1)), ops.vec(torch.exp(r(C, scale=0.1)))
z = torch.randn(B, C, hw, hw, device="cuda")
pre = torch.randn(B, 16 * 64, hw, hw, device="cuda") * 0.5
hf = torch.randn(B, 16 * 2 * C, hw, hw, device="cuda") * 0.5
raw, haff, h2 = ops.empty(B, 64, hw, hw), ops.empty(B, 96, hw, hw), ops.h2_empty(B, 64, hw, hw)
p0, hp = ops.pack_conv_x3(w0, 2), ops.pack_coupling_head(None, w2, s0, c0, s2, c2)
p4, e4 = ops.pack_conv_x3(w4, 1, lazy=True), ops.pack_epilogue(96, bias=r(96, scale=0.2), post_scale=torch.exp(r(96, scale=0.2)))
def main_work():
    for k in range(16):
        ops.conv_x3(z[:, :48], p0, raw, pre_add=pre[:, 64 * k: 64 * (k + 1)])
        ops.coupling_head(None, hp, raw, h2, pre_fmt=0)
        ops.conv_h2x(h2, p4, haff, epi=e4)
        ops.flow_pointwise(z, z, True, h_aff=haff, h_ft=hf[:, 2 * C * k: 2 * C * (k + 1)], w=wv, wt=wtt, an_bias=ab, an_escale=ae)
        yield
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
def timed(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
def serial():
    for _ in side_work(): pass
    for _ in main_work(): pass
def side_first():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for _ in side_work(): pass
    for _ in main_work(): pass
    main.wait_stream(side)
def main_first():
    side.wait_stream(main)
    for _ in main_work(): pass
    with torch.cuda.stream(side):
        for _ in side_work(): pass
    main.wait_stream(side)
def spread():
    side.wait_stream(main)
    sw = side_work()
    for i, _ in enumerate(main_work()):
        if i % 8 == 0:
            with torch.cuda.stream(side):
                next(sw, None)
    main.wait_stream(side)
t_side = timed(lambda: [None for _ in side_work()])
t_main = timed(lambda: [None for _ in main_work()])
print("alone: side work %.2f ms, main work %.2f ms, sum %.2f" % (t_side, t_main, t_side + t_main))
for name, f in (("one stream", serial), ("side enqueued first", side_first), ("main enqueued first", main_first), ("side launches spread over the main steps", spread)):
    print("%-45s %.2f ms" % (name, timed(f)), flush=True)
