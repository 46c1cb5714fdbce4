"""Interleaved A/B timing of the dense-block variants under SUSTAINED load (the chip is power-limited: bursts run 10-15 % faster than steady state and
box-to-box spread is +-3 %, so variants are compared round-robin inside one process, medians over rounds).
GPU box: python tools/chain_ab.py B H NB [rounds]"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator().manual_seed(0)
B, H, NB = (int(v) for v in sys.argv[1:4])
ROUNDS = int(sys.argv[4]) if len(sys.argv) > 4 else 5
shapes = ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64))
ring = [ops.h2_pack(torch.randn(B, 192, H, H, device="cuda") * 0.5 if i == 0 else torch.zeros(B, 192, H, H, device="cuda"), ops.h2_empty(B, 192, H, H)) for i in range(4)]
allw = [[ops.pack_conv_x3(torch.randn(co, ci, 3, 3, generator=g) * (0.05 / (ci * 9) ** 0.5), 1, lazy=True) for ci, co in shapes] for _ in range(NB)]
epis = [ops.pack_epilogue(co, bias=torch.zeros(co)) for ci, co in shapes]
def specs(b0, b1):
    out, cur = [], 0
    for r in range(NB):
        pws = allw[r]
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
sp = specs(0, B)
variants = {"conv_h2x launches": lambda: unfused(sp)}
if B % 2 == 0:
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    spa, spb = specs(0, B // 2), specs(B // 2, B)
    def two():
        side.wait_stream(main)
        for a, b in zip(spa, spb):
            unfused([a])
            with torch.cuda.stream(side):
                unfused([b])
        main.wait_stream(side)
    variants["conv_h2x two streams"] = two
for rows in (2,):
    for per, tune, tag in ((NB, 0, "all blocks/launch"), (NB, 0x10000, "all blocks/launch, eager publish"), (3, 0, "3 blocks/launch"), (0, 0, "one conv/launch")):
        if per == 0:
            chains = [ops.conv_chain([s]) for s in sp]
        else:
            chains = [ops.conv_chain(sp[5 * i: 5 * (i + per)]) for i in range(0, NB, per)]
        variants["chain %s" % tag] = (lambda cs=chains, t=tune: [c.run(tune=t) for c in cs])
times = {k: [] for k in variants}
for f in variants.values():
    f()
torch.cuda.synchronize()
for rnd in range(ROUNDS):
    for k, f in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1))
ops.check_range()
flops = 2 * 9 * sum(ci * co for ci, co in shapes) * B * H * H * NB
print("B=%d %dx%d, %d dense blocks, %d interleaved rounds; median (min..max) per dense block" % (B, H, H, NB, ROUNDS))
for k, v in times.items():
    m = statistics.median(v)
    print("  %-48s %8.1f us (%7.1f .. %7.1f)  %4.0f TFLOP/s-eq" % (k, m / NB * 1e3, min(v) / NB * 1e3, max(v) / NB * 1e3, flops / m / 1e9))
