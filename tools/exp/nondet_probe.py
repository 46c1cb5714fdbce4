"""Run-to-run determinism of the 8x LP pass at a given batch under different knob settings (bisecting a nondeterminism seen at B >= 16).
(The nondeterminism needs the packed build: BFSR_HIP_LIB=$PWD/tools/exp/libpk.so, see tools/exp/build_pk.sh.)
GPU box: python tools/exp/nondet_probe.py B [scale lr]"""
import os, sys, subprocess
B = sys.argv[1] if len(sys.argv) > 1 else "64"
scale = sys.argv[2] if len(sys.argv) > 2 else "8"
size = sys.argv[3] if len(sys.argv) > 3 else "96"
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from bfsr_amd import synth
from bfsr_amd.ops import HipOps
from bfsr_amd.srflow import options, spec
from bfsr_amd.srflow.models import create_model, models as registry
from bfsr_amd.srflow.test import lp_infer
B, scale, size = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ops = HipOps("cuda:0")
opt = options.load(options.DEFAULT_CONF)
if scale != 4:
    opt = options.derive_scale(opt, scale)
m = create_model(opt, ops=ops)
m.load_network(synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234))
prior = registry.make({"name": "unet", "args": {"depth": 3, "dim": 64, "bilinear": True, "ops": ops}, "sd": synth.state_dict_from_schema(spec.srflow_prior_schema(), 4321)}, load_sd=True).eval()
x = ops.to_device(synth.lr_batch(1, B, size, size))
ref = None
worst = {}
for it in range(4):
    x.add_(0.0)
    out = lp_infer(m, prior, x, return_all=True)
    cur = {"sr": out["sr"].clone()}
    for i, e in enumerate(out["epses"]):
        cur["eps%%d" %% i] = e.clone()
    for i, e in enumerate(out["epses_learned"]):
        cur["epsl%%d" %% i] = e.clone()
    cur["sr_raw"] = out["sr_raw"].clone()
    if ref is None:
        ref = cur
        continue
    for k in cur:
        d = (cur[k] - ref[k]).abs()
        n = int((d > 0).sum())
        if n:
            bad = sorted(set((d.flatten(1).max(1).values > 0).nonzero().flatten().tolist()))
            worst[k] = max(worst.get(k, (0, 0, []))[:2], (float(d.max()), n)) + (bad[:8],)
print("  ".join("%%s: %%.1e (%%d el, samples %%s)" %% (k, v[0], v[1], v[2]) for k, v in sorted(worst.items())) or "deterministic (4 runs)")
''' % os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
cases = [("default", {}), ("RRDB=launches", {"BFSR_RRDB": "launches"}), ("UP4=reg", {"BFSR_UP4": "reg"}), ("UP4C=0", {"BFSR_UP4C": "0"}), ("L3=reg", {"BFSR_L3": "reg"}),
         ("PRIOR=reg", {"BFSR_PRIOR": "reg"}), ("PRIOR_OVERLAP=0", {"BFSR_PRIOR_OVERLAP": "0"}), ("OVERLAP=0", {"BFSR_OVERLAP": "0"}), ("HOIST=staged", {"BFSR_HOIST": "staged"}),
         ("COUPLING=unfused", {"BFSR_COUPLING": "unfused"})]
only = os.environ.get("CASES")
for name, env in cases:
    if only and name not in only.split(","):
        continue
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD, B, scale, size], capture_output=True, text=True, env=e, timeout=600)
    tail = [l for l in r.stdout.splitlines() if l and not l.startswith("UNet")]
    print("%-18s %s" % (name, tail[-1] if tail else ("ERR " + r.stderr[-300:])), flush=True)
