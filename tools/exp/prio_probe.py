"""Does running the LP pass on a high-priority stream (the engine's side stream for hoists and the second RRDB half stays at normal priority) change the
step time?  GPU box: python tools/exp/prio_probe.py"""
import contextlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bfsr_amd import synth
from bfsr_amd.ops import HipOps
from bfsr_amd.srflow import options, spec
from bfsr_amd.srflow.models import create_model, models as registry
from bfsr_amd.srflow.test import lp_infer
ops = HipOps("cuda:0")
opt = options.load(options.DEFAULT_CONF)
model = create_model(opt, ops=ops)
model.load_network(synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234))
with contextlib.redirect_stdout(sys.stderr):
    prior = registry.make({"name": "unet", "args": {"depth": 3, "dim": 64, "bilinear": True, "ops": ops},
                           "sd": synth.state_dict_from_schema(spec.srflow_prior_schema(), 4321)}, load_sd=True).eval()
x = ops.to_device(synth.lr_batch(0, 8, 160, 160))
def run(stream, n=10):
    ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
    with ctx:
        for _ in range(3):
            x.add_(0.0); lp_infer(model, prior, x, check_range=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            x.add_(0.0); lp_infer(model, prior, x, check_range=False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    print("default stream: %.2f ms" % run(None), flush=True)
    print("high-priority stream for the pass: %.2f ms" % run(torch.cuda.Stream(priority=-1)), flush=True)
