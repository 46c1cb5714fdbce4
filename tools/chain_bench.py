"""Fused dense-block chain (conv_chain.hip) against the per-launch path (conv_h2x, one or two streams) on the dense blocks of the RRDB trunk.
GPU box: python tools/chain_bench.py [B H NB]   (defaults 8 160 21 = config 2 shape, 7 RRDBs)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator().manual_seed(0)
B, H, NB = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 160, 21)
ring = [ops.h2_pack(torch.randn(B, 192, H, H, device="cuda") * 0.5 if _ == 0 else torch.zeros(B, 192, H, H, device="cuda"), ops.h2_empty(B, 192, H, H)) for _ in range(4)]
shapes = ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64))
DISTINCT = os.environ.get("DISTINCT", "0") == "1"           # own weights per dense block (the real trunk) instead of five shared tensors
def mk():
    return [ops.pack_conv_x3(torch.randn(co, ci, 3, 3, generator=g) * (0.05 / (ci * 9) ** 0.5), 1, lazy=True) for ci, co in shapes]
allw = [mk() for _ in range(NB if DISTINCT else 1)]
epis = [ops.pack_epilogue(co, bias=torch.zeros(co)) for ci, co in shapes]
def specs(b0, b1, nb):
    out, cur = [], 0
    for r in range(nb):
        pws = allw[r % len(allw)]
        D, Dn = ring[cur][b0:b1], ring[(cur + 1) % 4][b0:b1]
        for i, (ci, co) in enumerate(shapes[:4]):
            out.append(dict(x=D[:, :ci // 8], pw=pws[i], out=D[:, ci // 8: ci // 8 + 4], epi=epis[i], act=2, slope=0.2))
        out.append(dict(x=D, pw=pws[4], out=Dn[:, :8], epi=epis[4], res1=D[:, :8], alpha1=0.2))
        cur = (cur + 1) % 4
    return out
def unfused(sp):
    for s in sp:
        kw = {k: v for k, v in s.items() if k not in ("x", "pw", "out")}
        ops.conv_h2x(s["x"], s["pw"], s["out"], **kw)
def timed(f, n=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
flops = 2 * 9 * sum(ci * co for ci, co in shapes) * B * H * H * NB
sp = specs(0, B, NB)
t = timed(lambda: unfused(sp))
print("B=%d %dx%d, %d dense blocks (%.2f TFLOP fp32-equivalent)" % (B, H, H, NB, flops / 1e12), flush=True)
print("  conv_h2x, one launch per conv, one stream : %8.3f ms  %6.1f us per block  %5.0f TFLOP/s-eq" % (t, t / NB * 1e3, flops / t / 1e9), flush=True)
if B % 2 == 0:
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    spa, spb = specs(0, B // 2, NB), specs(B // 2, B, NB)
    def two():
        side.wait_stream(main)
        for a, b in zip(spa, spb):
            unfused([a])
            with torch.cuda.stream(side):
                unfused([b])
        main.wait_stream(side)
    t = timed(two)
    print("  conv_h2x, two half-batch streams          : %8.3f ms  %6.1f us per block  %5.0f TFLOP/s-eq" % (t, t / NB * 1e3, flops / t / 1e9), flush=True)
for rows, per in ((2, 1), (2, 3), (2, NB)):
    chains = [ops.conv_chain(specs(0, B, NB)[5 * i: 5 * (i + per)]) for i in range(0, NB, per)]
    def run():
        for c in chains: c.run()
    t = timed(run)
    ops.check_range()
    print("  conv_chain %d rows/wave, %2d block(s) per launch: %8.3f ms  %6.1f us per block  %5.0f TFLOP/s-eq  (%.2f of 833)" % (rows, per, t, t / NB * 1e3, flops / t / 1e9, flops / t / 1e9 / 833), flush=True)
for ci, co in ((64, 32), (192, 64)):
    i = [s[0] for s in shapes].index(ci)
    pws = allw[0]
    D = ring[0]
    out = ring[1][:, :co // 8]
    one = dict(x=D[:, :ci // 8], pw=pws[i], out=out, epi=epis[i], act=2, slope=0.2)
    ch = ops.conv_chain([one])
    t3 = t2 = 1.0
    t1 = timed(lambda: ops.conv_h2x(one["x"], one["pw"], one["out"], epi=one["epi"], act=2, slope=0.2), 10)
    t2 = timed(lambda: ch.run(), 10)
    f = 2 * 9 * ci * co * B * H * H
    print("  single conv %3d->%2d: conv_h2x %7.1f us (%4.0f TFLOP/s-eq)   chain-of-one %7.1f us (%4.0f TFLOP/s-eq) 4 rows/wave, %7.1f us (%4.0f) 2 rows/wave" % (ci, co, t1 * 1e3, f / t1 / 1e9, t2 * 1e3, f / t2 / 1e9, t3 * 1e3, f / t3 / 1e9), flush=True)
