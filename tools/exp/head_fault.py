"""Round-4 study of the intermittent wrong-half-tile fault of round 3's 8-wave coupling_head.

    BFSR_HIP_LIB=tools/exp/libhf_<variant>.so BFSR_HEAD_WAVES=8 python tools/exp/head_fault.py MODE N

N rounds of encode(B=2) -> decode -> encode(sample 1) at 160x160 LR on fixed inputs (tools/exp/shard_repro.py's loop); a round
"differs" if any eps tensor is not bit-identical to round 0.  MODE:
  plain    nothing else
  poison   hid is filled with NaN before every head launch: a differing round WITH NaN in its eps = a store that never landed,
           WITHOUT = a wrongly computed value
  check    the product engine's BFSR_PAIR_DBG=checkn: every head output compared (no host synchronisation) with the generic fp32-MFMA kernel's;
           prints one HEAD MISMATCH line per faulty launch (row y -> wave y % 8, columns)
  trace    (libhf_trace.so) per-thread stage checksums of every head launch of the B=2 encode; the first differing round prints
           which stage differs first for which (launch, tile, wave, lanes):
           0 z1 global loads | 1 B fragments read from LDS | 3 pre_aff loads | 4 accumulators of the 3x3 | 5 B operand of the 1x1 |
           6 accumulators of the 1x1 | 7 values handed to the stores
Prints one summary line: "HF <variant> <mode>: D / N rounds differ ...".
"""
import ctypes
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from bfsr_amd import synth                                  # noqa: E402
from bfsr_amd.ops import HipOps, MODE_BILINEAR              # noqa: E402
from test_srflow_gpu import build                           # noqa: E402

MODE = sys.argv[1] if len(sys.argv) > 1 else "plain"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
LR = 160
variant = os.path.basename(os.environ.get("BFSR_HIP_LIB", "product")).replace("libhf_", "").replace(".so", "")
waves = int(os.environ.get("BFSR_HEAD_WAVES", "4"))

if MODE == "check":
    os.environ["BFSR_PAIR_DBG"] = "checkn"
hip = HipOps("cuda:0")
m, prior, opt, sd, psd = build(hip, 4)
eng = m.netG.module.engine()
lr = hip.to_device(synth.smooth_lr_batch(21, 2, LR, LR))
lr_up = hip.resize(lr, hip.empty(2, 3, LR * 4, LR * 4), MODE_BILINEAR, 0.25, 0.25)
lr1, lr_up1 = lr[1:2].clone(), lr_up[1:2].clone()

launches = []                                               # (B, H, W, Cz) of every head launch while recording
recording = False
orig_head = hip.coupling_head


def head(z, packed, pre_aff, hid, hid_fmt=0):
    if MODE == "poison":
        hid.fill_(float("nan"))
    if recording:
        launches.append((z.shape[0], z.shape[2], z.shape[3], packed[3]))
    return orig_head(z, packed, pre_aff, hid, hid_fmt=hid_fmt)


hip.coupling_head = head

trace = trace0 = None
if MODE == "trace":
    fn = hip.lib.bfsr_hf_trace
    fn.restype = ctypes.c_longlong
    fn.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int]
    CAP = 96 * 1024 * 1024                                  # dwords (384 MB)
    trace = torch.zeros(CAP, dtype=torch.int32, device="cuda:0")
    trace0 = torch.zeros(CAP, dtype=torch.int32, device="cuda:0")

STAGES = {0: "z1 loads", 1: "B frags (LDS)", 2: "A frags", 3: "pre_aff loads", 4: "acc 3x3", 5: "B operand 1x1", 6: "acc 1x1", 7: "stored values"}


def analyse(tr, tr0, used):
    NT = waves * 64
    off = 0
    for li, (B, H, W, Cz) in enumerate(launches):
        ntiles = ((W + 31) // 32) * ((H + waves - 1) // waves) * B
        n = ntiles * NT * 8
        if off + n > used:
            break
        a, b = tr[off:off + n].view(ntiles, waves, 64, 8), tr0[off:off + n].view(ntiles, waves, 64, 8)
        if not torch.equal(a, b):
            d = (a != b)
            tw = torch.nonzero(d.any(dim=3).any(dim=2))
            print("  first differing head launch: #%d (B=%d %dx%d Cz=%d), %d (tile, wave) pairs differ" % (li, B, H, W, Cz, tw.shape[0]))
            tiles_x = (W + 31) // 32
            tiles_xy = tiles_x * ((H + waves - 1) // waves)
            for t, w in tw[:12].tolist():
                dd = d[t, w]                                # [64 lanes][8 stages]
                st = [s for s in range(8) if bool(dd[:, s].any())]
                tile = t % tiles_xy
                desc = []
                for s in st:
                    lanes = torch.nonzero(dd[:, s]).flatten().tolist()
                    desc.append("%s: lanes %s" % (STAGES[s], _ranges(lanes)))
                print("    tile %d (b=%d x0=%d y0=%d) wave %d -> %s" % (t, t // tiles_xy, (tile % tiles_x) * 32, (tile // tiles_x) * waves, w, "; ".join(desc)))
            return
        off += n
    print("  traces identical for all %d recorded launches (the difference entered elsewhere)" % len(launches))


def _ranges(v):
    out, i = [], 0
    while i < len(v):
        j = i
        while j + 1 < len(v) and v[j + 1] == v[j] + 1:
            j += 1
        out.append("%d-%d" % (v[i], v[j]) if j > i else "%d" % v[i])
        i = j + 1
    return ",".join(out)


ref2 = ref1 = None
differ = nan_rounds = reported = 0
for it in range(N):
    if MODE == "trace":
        hip.lib.bfsr_hf_trace(trace.data_ptr(), CAP, 64)
        recording = (it == 0)
    ep = [e.clone() for e in eng.encode(lr_up, lr)]
    recording = False
    used = 0
    if MODE == "trace":
        used = hip.lib.bfsr_hf_trace(None, 0, 0)
    rt = eng.decode(lr, epses=[e.clone() for e in ep])
    ep1 = [e.clone() for e in eng.encode(lr_up1, lr1)]
    torch.cuda.synchronize()
    if ref2 is None:
        ref2, ref1 = ep, ep1
        if MODE == "trace":
            trace0.copy_(trace)
        continue
    bad = any(not torch.equal(a, b) for a, b in zip(ep, ref2)) or any(not torch.equal(a, b) for a, b in zip(ep1, ref1))
    if bad:
        differ += 1
        has_nan = any(bool(torch.isnan(a).any()) for a in ep + ep1)
        nan_rounds += int(has_nan)
        if reported < 3:
            reported += 1
            for lvl, (a, b) in enumerate(zip(ep, ref2)):
                if not torch.equal(a, b):
                    df = (a - b).abs()
                    idx = torch.nonzero(~(df == 0))
                    print("round %d: B=2 eps%d differs: max %.3e, %d elements, first %s, nan=%s" % (it, lvl, float(torch.nan_to_num(df, nan=9e9).max()), idx.shape[0], idx[0].tolist(), has_nan), flush=True)
            if MODE == "trace" and any(not torch.equal(a, b) for a, b in zip(ep, ref2)):
                analyse(trace, trace0, used)
print("HF %s waves=%d %s: %d / %d rounds differ%s" % (variant, waves, MODE, differ, N - 1, (", %d of them with NaN" % nan_rounds) if MODE == "poison" else ""), flush=True)
