"""Error map of coupling_step vs the torch semantics (debug aid)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bfsr_amd import _lib
if os.environ.get("STEPLIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ["STEPLIB"])
from bfsr_amd.ops import HipOps
from cpu_ops import CpuOps
hip, CPU = HipOps("cuda:0"), CpuOps()
def rnd(seed, *shape, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))
C = int(sys.argv[1]) if len(sys.argv) > 1 else 24
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16, 40)
rev = int(sys.argv[4]) if len(sys.argv) > 4 else 1
B, cn, cc2 = 1, C // 2, 2 * (C - C // 2)
z, pre = rnd(161, B, C, H, W), rnd(162, B, 64, H, W, scale=0.5)
w0, w2 = rnd(163, 64, cn, 3, 3, scale=0.1), rnd(164, 64, 64, 1, 1, scale=0.1)
s0, c0, s2, c2 = rnd(165, 64, scale=0.1), torch.exp(rnd(166, 64, scale=0.1)), rnd(167, 64, scale=0.1), torch.exp(rnd(168, 64, scale=0.1))
w4, b4, ps = rnd(173, cc2, 64, 3, 3, scale=0.02), rnd(174, cc2, scale=0.2), torch.exp(rnd(175, cc2, scale=0.2))
cpk = CPU.pack_coupling_step(w0, w2, s0, c0, s2, c2, w4, b4, ps)
hpk = hip.pack_coupling_step(w0, w2, s0, c0, s2, c2, w4, b4, ps)
ref = CPU.coupling_step(z, torch.empty_like(z), cpk, pre, rev)
zd, pd = hip.to_device(z), hip.to_device(pre)
bad = 0
for rep in range(int(os.environ.get("REPS", "1"))):
    out = hip.coupling_step(zd, hip.empty(B, C, H, W), hpk, pd, rev).cpu()
    err = (out - ref).abs()[0]
    bad += int(err.max().item() > 1e-4)
print("runs with errors: %d of %s" % (bad, os.environ.get("REPS", "1")))
print("max err %.3e; per channel:" % err.max().item(), ["%.1e" % v for v in err.amax((1, 2)).tolist()])
m = err.amax(0)
for y in range(H):
    print("".join("#" if v > 1e-3 else ("+" if v > 2e-5 else ".") for v in m[y].tolist()))
