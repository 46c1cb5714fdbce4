#!/usr/bin/env python
"""bench.py -- HR MPix/s of the BFSR hot path on MI355X, one JSON line per run (contract in the task statement).

A "step" = one pass of the hot path over one synthetic LR batch already resident in HBM.  `--config` names the BASELINE.json
configuration (default 2 = the headline, the one the metric is quoted on and that fits one GPU):

  2  SRFlow-LP 4x DF2K config, batch 8 of 160x160 LR per GPU -> 640x640 (weak scaling: 8 crops per GPU at every N)
  3  LINF-LP rrdb-linf-LP, batch 16 of 256x256 LR per GPU, arbitrary scale x2/x3/x4 (`--linf-scale`, default 4)
  4  SRFlow-LP 8x config (scale 8, L=3), the NAMED batch of 64 crops of 96x96 LR sharded over the N GPUs (strong scaling)
  5  LINF-LP rrdb-linf-LP, out-of-distribution x6, the NAMED batch of 128 crops of 128x128 LR sharded over the N GPUs,
     fp16 MFMA path (strong scaling)

SRFlow step: RRDB conditioning encoder -> flow encode -> eps standardise -> prior UNet -> flow decode -> clamp.
LINF step:   device-side input prep -> encoder -> query_log_p -> prior UNet -> query_rgb -> fold + skip -> clamp.
For N > 1 (one process per GPU, `torch.distributed` / RCCL) every step's outputs are all-gathered; the gather of step i is
double-buffered and runs while step i+1 computes (bfsr_amd.dist.AsyncGatherer), the timed region ends when the last gather has.

  python bench.py [--config 2] --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W [--config C]

JSON: `roofline` = the launch shape with the largest total time in the timed region (HIP events on the launch stream around
every launch of the top shapes), `roofline_by_symbol` = the same ranking by kernel family, `roofline_coupling_inverse` = the
HBM-bound fused FlowStep-inverse kernel (SRFlow configs), `cpu_baseline` = the oracle (reference-faithful torch-CPU port) on a
bounded sample: 1 warm-up + median of 3, thread count and host core count stated, `parity` = HIP vs oracle on that sample."""
import argparse
import contextlib
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md (dense peaks, no sparsity)
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_16BIT_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0
# family -> (kernel symbol, hardware peak TFLOP/s, MFMA instructions per fp32-equivalent product, arithmetic)
FAMILIES = {
    "conv": ("conv_mfma_kernel", PEAK_F32_MFMA_TFLOPS, 1, "native fp32 MFMA (v_mfma_f32_32x32x2_f32)"),
    "conv+1x1": ("conv_mfma_kernel<FUSE2>", PEAK_F32_MFMA_TFLOPS, 1, "native fp32 MFMA, fused 3x3 + 1x1"),
    "conv_up2": ("conv_up2_kernel", PEAK_F32_MFMA_TFLOPS, 1, "native fp32 MFMA"),
    "conv_bf16x3": ("conv_bf16x3_kernel", PEAK_16BIT_MFMA_TFLOPS, 6, "3xBF16 split on v_mfma_f32_32x32x16_bf16"),
    "conv_x3s": ("conv3x3_x3s_kernel", PEAK_16BIT_MFMA_TFLOPS, 6, "3xBF16 split on v_mfma_f32_32x32x16_bf16, x3-tensor input by LDS-DMA"),
    "conv_up2_x3": ("conv_up2_bf16x3_kernel", PEAK_16BIT_MFMA_TFLOPS, 6, "3xBF16 split, parity-decomposed conv over nearest-x2 input"),
    "conv_up4_x3": ("conv_up4_bf16x3_kernel", PEAK_16BIT_MFMA_TFLOPS, 6, "3xBF16 split, phase-decomposed conv over nearest-x4 input"),
    "conv_f16x2": ("conv_bf16x3_kernel<PL=2>", PEAK_16BIT_MFMA_TFLOPS, 3, "two-term fp16 split (22-bit operands, power-of-two weight scale), 3 products on v_mfma_f32_32x32x16_f16"),
    "conv_h2x": ("conv3x3_h2x_kernel", PEAK_16BIT_MFMA_TFLOPS, 3, "two-term fp16 split, 3 products on v_mfma_f32_32x32x16_f16, h2-tensor input by LDS-DMA"),
    "conv_chain": ("conv_chain_kernel", PEAK_16BIT_MFMA_TFLOPS, 3, "the RRDB trunk (all dense-block convs + trunk_conv) as ONE persistent launch with per-tile dependency counters; conv3x3_h2x_kernel's arithmetic: two-term fp16 split, 3 products on v_mfma_f32_32x32x16_f16, h2-tensor input by LDS-DMA"),
    "conv_up2_f2": ("conv_up2_bf16x3_kernel<PL=2>", PEAK_16BIT_MFMA_TFLOPS, 3, "two-term fp16 split, parity-decomposed conv over nearest-x2 input"),
    "conv_up2_h2t": ("conv_up2_h2t_kernel", PEAK_16BIT_MFMA_TFLOPS, 3, "two-term fp16 split, conv over cat[key, nearest-x2 taps] at source resolution (parity-decomposed taps, space-to-depth key chunks), h2 input by LDS-DMA"),
    "conv_h2r": ("coupling_tail_kernel<plain conv>", PEAK_16BIT_MFMA_TFLOPS, 3, "two-term fp16 split, 3x3 conv 64 -> <=32 channels over an h2 tensor on the coupling tail's ring kernel"),
    "conv_up4_h2t": ("conv_up4_h2t_kernel", PEAK_16BIT_MFMA_TFLOPS, 3, "two-term fp16 split, conv over nearest-x4 taps at source resolution (25 pre-summed phase blocks), h2 input by LDS-DMA, quad-major output"),
    "conv_up4_f2": ("conv_up4_bf16x3_kernel<PL=2>", PEAK_16BIT_MFMA_TFLOPS, 3, "two-term fp16 split, phase-decomposed conv over nearest-x4 input"),
    "conv_f16": ("conv_f16_kernel", PEAK_16BIT_MFMA_TFLOPS, 1, "fp16 MFMA (v_mfma_f32_32x32x16_f16), fp32 accumulate"),
    "conv_h2s": ("conv3x3_h2s_kernel", PEAK_16BIT_MFMA_TFLOPS, 1, "fp16 MFMA, fp16-stored (h2) input by LDS-DMA, fp32 accumulate"),
    "linf_mlp_x3": ("linf_mlp_kernel<x3>", PEAK_16BIT_MFMA_TFLOPS, 6, "3xBF16 split; Fourier features + 4-layer MLP fused"),
    "linf_mlp_f2": ("linf_mlp_kernel<f16x2>", PEAK_16BIT_MFMA_TFLOPS, 3, "two-term fp16 split, 3 products; Fourier features + 4-layer MLP fused"),
    "linf_mlp_f16": ("linf_mlp_kernel<fp16>", PEAK_16BIT_MFMA_TFLOPS, 1, "fp16 MFMA; Fourier features + 4-layer MLP fused"),
    "conv1x1_f16": ("conv1x1_kernel<fp16>", PEAK_16BIT_MFMA_TFLOPS, 1, "fp16 MFMA GEMM over pixels"),
    "conv1x1_x3": ("conv1x1_kernel<x3>", PEAK_16BIT_MFMA_TFLOPS, 6, "3xBF16 split GEMM over pixels"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json configuration (see module docstring)")
    ap.add_argument("--linf-scale", type=float, default=4.0, help="config 3: the arbitrary scale (2, 3 or 4)")
    ap.add_argument("--batch", type=int, default=None, help="override: LR crops per GPU")
    ap.add_argument("--lr", type=int, default=None, help="override: LR crop side")
    ap.add_argument("--scale", type=int, default=None, choices=[4, 8], help="override: SRFlow scale (8 = config 4's model)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-line", action="store_true", help="config 2: skip the secondary all-native-fp32-MFMA timing")
    ap.add_argument("--cpu-lr", type=int, default=None, help="LR side of the CPU-baseline sample (B=1)")
    ap.add_argument("--mode", default="lp", choices=["lp", "tau"],
                    help="SRFlow configs: lp = the learned-prior pipeline (headline); tau = sampling path: decode eps ~ 0.9*N(0,1) "
                         "without encode/prior (SURVEY 8d secondary workload)")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------------------------
def launch_flop(k):
    """algorithmic flops of one launch (result-preserving schedule) from its ops.py profile key, or None"""
    f = k[0]
    if f in ("conv", "conv_bf16x3", "conv_f16", "conv_f16x2"):
        _, KS, _, Cin, Cout, b_, hh, ww = k
        return 2.0 * Cin * KS * KS * Cout * b_ * hh * ww
    if f == "conv+1x1":
        _, Cin, Cout, b_, hh, ww = k
        return 2.0 * (Cin * 9 + 64) * Cout * b_ * hh * ww
    if f in ("conv_x3s", "conv_h2s", "conv_h2x", "conv_h2r"):
        _, Cin, Cout, b_, hh, ww, _fmt = k[:7]                             # (conv_h2s keys carry the workgroup's M-tile count behind the format)
        return 2.0 * Cin * 9 * Cout * b_ * hh * ww
    if f == "conv_chain":                                               # key: number of convs, sum over them of Cin x Cout, batch, H, W
        _, _n, cc, b_, hh, ww = k
        return 2.0 * 9 * cc * b_ * hh * ww
    if f in ("linf_mlp_x3", "linf_mlp_f16", "linf_mlp_f2"):           # layer 1 (4 neighbours x 256 features) + two hidden layers + output layer
        _, hid, Cout, b_, qh, qw = k
        return 2.0 * (4 * hid * hid + 2 * hid * hid + hid * Cout) * b_ * qh * qw
    if f in ("conv1x1_f16", "conv1x1_x3"):
        _, Cin, Cout, b_, hh, ww = k
        return 2.0 * Cin * Cout * b_ * hh * ww
    if f in ("conv_up2", "conv_up2_x3", "conv_up2_f2"):               # 2x2 source taps per output pixel (parity pre-summed weights)
        _, _, Cin, Cout, b_, hh, ww, cin2 = k          # + cin2 key channels at output resolution (9 taps)
        return 2.0 * (Cin * 4 + cin2 * 9) * Cout * b_ * hh * ww
    if f == "conv_up2_h2t":                                              # Ct taps channels: 2x2 source taps per output pixel; Ck key channels: 9 taps
        _, ct, ck, Cout, b_, hh, ww = k
        return 2.0 * (ct * 4 + ck * 9) * Cout * b_ * hh * ww
    if f == "conv_up4_h2t":                                              # 25 pre-summed matrices per 16 output pixels (key[2] = 9: compact output)
        _, ct, _, Cout, b_, hh, ww = k
        return 2.0 * ct * 25.0 / 16.0 * Cout * b_ * hh * ww
    if f in ("conv_up4_x3", "conv_up4_f2"):                             # 25 pre-summed matrices per 16 output pixels
        _, _, Cin, Cout, b_, hh, ww, _ = k
        return 2.0 * Cin * 25.0 / 16.0 * Cout * b_ * hh * ww
    return None


def _evidence_order(path):
    """profiles/ files in the order they were produced: round, then tag (r06a ... r06z, r06aa ... -- the two-letter tags of a round follow its one-letter ones)."""
    tag = os.path.basename(path).split("_")[0]
    return (tag[:3], len(tag), tag, os.path.basename(path))


def rocprof_avg_ms(cfg, sym):
    """Average duration of a kernel symbol in the committed rocprofv3 --kernel-trace --stats summary of this configuration (the latest
    profiles/rNN*_cfg<cfg>_kernel_stats.csv), or (None, None).  Only meaningful for a launch shape that is the symbol's only one."""
    import csv
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_cfg%d_kernel_stats.csv" % cfg)), key=_evidence_order)
    if not fs:
        return None, None
    base = sym.split("<")[0].split(" ")[0]
    try:
        rows = [r for r in csv.DictReader(open(fs[-1])) if base in (r.get("Name") or "")]
        if len(rows) != 1:
            return None, None
        return float(rows[0]["AverageNs"]) * 1e-6, os.path.basename(fs[-1])
    except (ValueError, OSError, KeyError, TypeError) as e:            # evidence files are optional input
        print("bench: ignoring %s (%s)" % (fs[-1], e), file=sys.stderr)
        return None, None


def roof_entry(t, k, f, n, steps, step_ms, traffic_db, cfg=None):
    sym, hw_peak, per_prod, arith = FAMILIES[k[0]]
    a = f / (t / n * 1e-3) / 1e12
    peak = hw_peak / per_prod
    e = {"bound": "mfma", "kernel": "%s %s" % (sym, list(k)), "achieved": round(a, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
         "frac": round(a / peak, 4), "traffic": None,
         "peak_equiv_fp32": round(peak, 1), "peak_hw": hw_peak, "achieved_hw": round(a * per_prod, 1),
         "peak_basis": ("%s: hardware peak %.0f TFLOP/s dense, %d MFMA(s) per fp32-equivalent product -> `peak` = %.1f is a DERIVED "
                        "fp32-equivalent figure, not a hardware peak; frac = achieved/peak = achieved_hw/peak_hw"
                        % (arith, hw_peak, per_prod, peak)),
         "algorithmic_flop_per_launch": f, "avg_launch_ms": round(t / n, 4), "launches": n,
         "share_of_step": round(t / steps / step_ms, 4)}
    ent = traffic_db.get(json.dumps(list(k)))
    if ent:
        e["traffic"] = ent["hbm_bytes_per_launch"]
        e["traffic_unit"] = ent.get("unit", "HBM-side bytes per launch (rocprofv3 FETCH_SIZE raw + WRITE_SIZE, profiles/)")
    # the same fraction from the rocprofv3 average of the committed kernel-stats file (its launches are spaced by the profiler and run a few
    # per cent slower than back to back under HIP events: both are printed)
    if cfg is not None and k[0] in ("conv_chain",):
        ms, src = rocprof_avg_ms(cfg, sym)
        if ms:
            e["avg_launch_ms_rocprof"] = round(ms, 4)
            e["frac_rocprof"] = round(f / (ms * 1e-3) / 1e12 / peak, 4)
            e["frac_rocprof_source"] = "profiles/%s (committed; rocprofv3 --kernel-trace --stats of this command)" % src
    return e


def load_traffic():
    """per-launch HBM-side bytes from the committed PMC passes of this round (separate --pmc runs, profiles/), keyed by launch shape"""
    db = {}
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_pmc_traffic*.json")), key=_evidence_order):   # in order of production: later passes override
        try:
            db.update(json.load(open(p)).get("kernels", {}))
        except (ValueError, OSError, AttributeError) as e:          # an unreadable evidence file must not take the benchmark down with it
            print("bench: ignoring %s (%s)" % (p, e), file=sys.stderr)
    return db


def median_time(fn, warm=1, reps=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t)
    return statistics.median(ts), ts, r


# ----------------------------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    from bfsr_amd import dist as bdist, synth
    from bfsr_amd.ops import HipOps

    # the container's CPU quota (16 of the host's 256 hardware threads on the pool): torch's default intra-op pool (128 threads) gets the whole
    # process throttled by the cgroup the moment any CPU tensor op runs (bfsr_amd/hostenv.py; measured: LINF passes 21 -> 21 / 57 / 96 ms at random)
    from bfsr_amd import hostenv
    host_threads = hostenv.cap_torch_threads()
    rank, world, local = bdist.init()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, world))
    local = local % torch.cuda.device_count()      # (ranks may share a GPU in a 1-GPU smoke test of the N>1 path)
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    ops = HipOps(dev)
    cfg = args.config
    srflow = cfg in (2, 4)

    # ---- workload ----------------------------------------------------------------------------------------------------------
    if srflow:
        scale = args.scale or (4 if cfg == 2 else 8)
        h = args.lr or (160 if cfg == 2 else 96)
        if cfg == 2:
            B, total, scaling = args.batch or 8, None, "weak"
        else:
            total = 64
            if total % world:
                raise SystemExit("config 4 shards a batch of 64 crops: --gpus must divide 64")
            B, scaling = args.batch or total // world, "strong"
    else:
        scale = float(args.linf_scale) if cfg == 3 else 6.0
        h = args.lr or (256 if cfg == 3 else 128)
        precision = "fp32" if cfg == 3 else "fp16"
        if cfg == 3:
            B, total, scaling = args.batch or 16, None, "weak"
        else:
            total = 128
            if total % world:
                raise SystemExit("config 5 shards a batch of 128 crops: --gpus must divide 128")
            B, scaling = args.batch or total // world, "strong"
    H = int(round(h * scale))
    global_B = B * world

    # ---- models ------------------------------------------------------------------------------------------------------------
    if srflow:
        from bfsr_amd.srflow import options, spec
        from bfsr_amd.srflow.models import create_model, models as registry
        from bfsr_amd.srflow.test import lp_infer
        opt = options.load(options.DEFAULT_CONF)
        if scale != 4:
            opt = options.derive_scale(opt, scale)
        sd = synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234)
        psd = synth.state_dict_from_schema(spec.srflow_prior_schema(), 4321)
        model = create_model(opt, ops=ops)
        model.load_network(sd)
        with contextlib.redirect_stdout(sys.stderr):      # make_unet prints its arguments like the reference does; keep
            prior = registry.make({"name": "unet", "args": {"depth": 3, "dim": 64, "bilinear": True, "ops": ops}, "sd": psd},
                                  load_sd=True).eval()    # stdout for the single JSON line
    else:
        from bfsr_amd.linf import spec as lspec
        from bfsr_amd.linf.models import make
        from bfsr_amd.linf.test import infer_from_lr
        mspec = {"name": "linf-patch", "args": {"encoder_spec": {"name": "rrdb", "args": {"no_upsampling": True}},
                                                 "imnet_spec": {"name": "flow", "args": {"name": "flow"}},
                                                 "flow_layers": 10, "num_layer": 3, "hidden_dim": 256}}
        sd = synth.state_dict_from_schema(lspec.linf_schema(mspec["args"]["encoder_spec"]), 2024)
        psd = synth.state_dict_from_schema(lspec.linf_prior_schema(27), 777)
        with contextlib.redirect_stdout(sys.stderr):
            model = make(mspec, args={"ops": ops, "precision": precision}).eval()
            model.load_state_dict(sd)
            prior = make({"name": "unet", "args": {"in_chans": 27, "depth": 3, "dim": 64, "bilinear": True}},
                         args={"ops": ops, "precision": precision}).eval()
            prior.load_state_dict(psd)

    # distinct seeded batches per rank and step (global sample index = seed), resident in HBM before timing
    n_batches = max(2, min(args.steps + args.warmup, 4))
    batches = [ops.to_device(synth.lr_batch(1000 * rank + i, B, h, h)) for i in range(n_batches)]

    tau_eps = None
    if srflow and args.mode == "tau":     # eps resident before timing: 0.9*N(0,1) from the same PCG64 stream family
        import numpy as np
        tau_eps = []
        for i in range(n_batches):
            g = np.random.Generator(np.random.PCG64(5000 + 1000 * rank + i))
            tau_eps.append([ops.to_device(torch.from_numpy((0.9 * g.standard_normal((B, 6, H // 2, H // 2))).astype(np.float32))),
                            ops.to_device(torch.from_numpy((0.9 * g.standard_normal((B, 96, H // 8, H // 8))).astype(np.float32)))])

    gatherer = bdist.AsyncGatherer(global_B)

    def infer(x, i):
        if not srflow:
            return infer_from_lr(model, prior, x, scale)
        if tau_eps is not None:
            sr = model.netG.module.engine().decode(x, epses=tau_eps[i % n_batches])
            return ops.axpb_clamp(sr, ops.empty(*sr.shape), 1.0, 0.0, 0.0, 1.0)
        return lp_infer(model, prior, x)

    def step(i):
        x = batches[i % n_batches]
        x.add_(0.0)                       # bump the version so the conditioning cache never hits across steps
        sr = infer(x, i)
        if torch.distributed.is_initialized():      # world > 1, or BFSR_DIST_FORCE=1: the real collective even for one rank
            gatherer.submit(sr)           # gather of step i overlaps the compute of step i+1
        return sr

    # warm-up, then ONE untimed ranking step that brackets EVERY launch with HIP events to rank the launch shapes; the timed region
    # then only brackets the launches of the top shapes (+ the inverse tail), keeping the host overhead negligible
    C1 = 12
    # the HBM-bound "coupling inverse" kernel of level 1: the fused tail (Conv2dZeros 64->12 on 16-wide MFMA tiles + the whole
    # pointwise chain of the step) by default, the pointwise-only kernel with BFSR_COUPLING=unfused
    key_tail_fused = ("coupling_tail", 1, C1, B, H // 2, H // 2) if srflow else None
    key_tail = ("flow", 1, C1, B, H // 2, H // 2, True, True, True) if srflow else None
    for i in range(args.warmup):
        step(i)
    gatherer.finish()
    torch.cuda.synchronize()
    # one extra UNTIMED ranking step after the warm-up (allocator and caches are warm, so no first-touch stalls leak into the event
    # times): every launch bracketed by HIP events
    ops.profile_keys, ops.profile = "ALL", {}
    step(args.warmup)
    gatherer.finish()
    torch.cuda.synchronize()
    ranked = sorted(((sum(s_.elapsed_time(e_) for s_, e_ in ev), k) for k, ev in ops.profile.items()
                     if k[0] in FAMILIES), reverse=True)
    warm_by_family = {}
    warm_total_ms = 0.0
    for k, ev in ops.profile.items():
        t = sum(s_.elapsed_time(e_) for s_, e_ in ev)
        warm_total_ms += t
        fam = FAMILIES[k[0]][0] if k[0] in FAMILIES else k[0]
        warm_by_family[fam] = warm_by_family.get(fam, 0.0) + t
    key_head = ("coupling_head", C1 // 2, B, H // 2, H // 2) if srflow else None
    ops.profile_keys = set([k for _, k in ranked[:4]] + ([key_tail, key_tail_fused, key_head] if key_tail else []))
    ops.profile = {}
    bdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + 1 + i)
    gathered = gatherer.finish()
    torch.cuda.synchronize()
    bdist.barrier()
    dt = time.perf_counter() - t0
    if torch.distributed.is_initialized() and (gathered is None or gathered.shape[0] != global_B):
        raise SystemExit("bench: the gathered output of the last step has %s rows, expected the global batch %d"
                         % (None if gathered is None else gathered.shape[0], global_B))
    ops.profile_keys = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    mpix = global_B * H * H / 1e6
    value = mpix * args.steps / dt
    step_ms = dt / args.steps * 1e3

    # ---- roofline ----------------------------------------------------------------------------------------------------------
    traffic_db = load_traffic()
    totals = []
    for k, ev in ops.profile.items():
        f = launch_flop(k)
        if f is not None:
            totals.append((sum(s_.elapsed_time(e_) for s_, e_ in ev), k, f, len(ev)))
    totals.sort(reverse=True)
    roofline = roof_entry(*totals[0], args.steps, step_ms, traffic_db, cfg) if totals else None
    roofline_next = [roof_entry(*x, args.steps, step_ms, traffic_db) for x in totals[1:4]]
    by_symbol = [{"kernel": fam, "ms_per_step": round(t, 3), "share_of_event_time": round(t / warm_total_ms, 4)}
                 for fam, t in sorted(warm_by_family.items(), key=lambda kv: -kv[1])[:8]] if warm_total_ms else None
    roof_tail = None
    if key_tail and (ops.profile.get(key_tail_fused) or ops.profile.get(key_tail)):
        fused = bool(ops.profile.get(key_tail_fused))
        ev = ops.profile[key_tail_fused if fused else key_tail]
        tail_ms = sum(s.elapsed_time(e) for s, e in ev) / len(ev)
        px = B * (H // 2) * (H // 2)
        # Two numerators, both reported.  (1) SURVEY 8d's definition of the coupling-inverse tail: read z, h_aff, h_ft + write z =
        # 20*C B/px.  (2) what the fused tail must move: h_aff never exists in HBM, the kernel reads the 64 hidden channels of the
        # coupling net instead (an h2 tensor, 4 B per element), + h_ft (2C) + z (C) and writes z (C) = 4*(64 + 4C) B/px.
        survey_bytes = 20.0 * C1 * px
        tail_bytes = (4.0 * (64 + 4 * C1) if fused else 20.0 * C1) * px
        a = tail_bytes / (tail_ms * 1e-3) / 1e9
        a_s = survey_bytes / (tail_ms * 1e-3) / 1e9
        roof_tail = {"bound": "hbm",
                     "kernel": ("coupling_tail_kernel<0, 12, reverse> (coupling_tail.hip): Conv2dZeros 64->12 over the h2 hidden tensor (LDS-DMA, "
                                "two-term fp16 split) + level-1 FlowStep inverse tail as its epilogue" if fused
                                else "flow_pointwise_kernel<12,4> reverse (level-1 FlowStep inverse tail)"),
                     "achieved": round(a, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(a / PEAK_HBM_GBS, 4),
                     "achieved_on_survey_bytes": round(a_s, 1), "frac_on_survey_bytes": round(a_s / PEAK_HBM_GBS, 4),
                     "traffic": (traffic_db.get(json.dumps(list(key_tail_fused if fused else key_tail))) or {}).get("hbm_bytes_per_launch"),
                     "algorithmic_bytes_per_launch": tail_bytes, "survey_bytes_per_launch": survey_bytes,
                     "avg_launch_ms": round(tail_ms, 4), "launches": len(ev)}
        if fused and ops.profile.get(key_head):
            # the whole sequential remainder of a level-1 coupled FlowStep = coupling_head + coupling_tail, on the bytes the PAIR must move
            # (head: z1 + pre_aff in, hid out; tail as above) and on SURVEY's 20*C B/px
            evh = ops.profile[key_head]
            head_ms = sum(s.elapsed_time(e) for s, e in evh) / len(evh)
            head_bytes = 4.0 * (C1 // 2 + 64 + 64) * px
            pair = (head_bytes + tail_bytes) / ((head_ms + tail_ms) * 1e-3) / 1e9
            sa = survey_bytes / ((head_ms + tail_ms) * 1e-3) / 1e9
            roof_tail["step"] = {"kernels": "coupling_head_kernel<1,4> + coupling tail (one coupled level-1 FlowStep, inverse)",
                                 "head_avg_launch_ms": round(head_ms, 4), "tail_avg_launch_ms": round(tail_ms, 4),
                                 "head_bytes_per_launch": head_bytes, "head_achieved": round(head_bytes / (head_ms * 1e-3) / 1e9, 1),
                                 "head_frac": round(head_bytes / (head_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                 "pair_bytes_per_step": head_bytes + tail_bytes, "achieved": round(pair, 1), "unit": "GB/s",
                                 "frac": round(pair / PEAK_HBM_GBS, 4),
                                 "achieved_on_survey_bytes": round(sa, 1), "frac_on_survey_bytes": round(sa / PEAK_HBM_GBS, 4),
                                 "note": "event times under the side-stream overlap are inflated by contention; tools/step_bench.py times the two "
                                         "kernels alone (profiles/r04_*_step_bench.txt)"}

    # ---- config 2: the same workload with every contraction on the native fp32 MFMA, reported beside `value` --------------
    fp32_only = None
    if cfg == 2 and rank == 0 and world == 1 and ops.conv_mode != "f32" and not args.no_fp32_line and args.mode == "lp":
        ops32 = HipOps(dev)
        ops32.conv_mode = "f32"
        m32 = create_model(opt, ops=ops32)
        m32.load_network(sd)
        with contextlib.redirect_stdout(sys.stderr):
            p32 = registry.make({"name": "unet", "args": {"depth": 3, "dim": 64, "bilinear": True, "ops": ops32}, "sd": psd},
                                load_sd=True).eval()
        for i in range(max(1, min(args.warmup, 2))):
            batches[i % n_batches].add_(0.0)
            lp_infer(m32, p32, batches[i % n_batches])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            batches[i % n_batches].add_(0.0)
            lp_infer(m32, p32, batches[i % n_batches])
        torch.cuda.synchronize()
        d32 = time.perf_counter() - t1
        fp32_only = {"value": round(B * H * H / 1e6 * args.steps / d32, 4), "unit": "MPix/s", "ms_per_step": round(d32 / args.steps * 1e3, 3),
                     "note": "same workload and steps, all convs on v_mfma_f32_32x32x2_f32 (BFSR_CONV=f32)"}
        del m32, p32, ops32
        torch.cuda.empty_cache()

    # ---- CPU baseline (rank 0, N = 1 only) + parity of the HIP path on the same sample -------------------------------------
    cpu_baseline, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ncpu = os.cpu_count() or 1
        # the oracle is a torch-CPU (MKL-DNN) program: measured on the GPU box's 256-core host it is fastest at 16 threads
        # (8/16/32/64/128 threads: 7.4 / 5.7 / 6.8 / 16.1 / 30.5 s per 160x160 crop, profiles/r03_cpu_thread_sweep.json)
        nt = min(16, ncpu, hostenv.effective_cpus())
        torch.set_num_threads(nt)
        if srflow:
            import numpy as np
            import oracle.srflow_ref as O          # checker / baseline only
            cl = args.cpu_lr or (160 if cfg == 2 else 96)
            x = synth.lr_batch(99, 1, cl, cl)
            nb = opt["network_G"]["nb"]
            if args.mode == "tau":
                g = np.random.Generator(np.random.PCG64(4999))
                ep = [torch.from_numpy((0.9 * g.standard_normal((1, 6, cl * scale // 2, cl * scale // 2))).astype(np.float32)),
                      torch.from_numpy((0.9 * g.standard_normal((1, 96, cl * scale // 8, cl * scale // 8))).astype(np.float32))]
                med, ts, ref = median_time(lambda: O.srflow_decode(x, ep, sd, opt, nb))
                what = "oracle srflow_decode (RRDB + reverse flow)"
                out = model.netG.module.engine().decode(ops.to_device(x), epses=[ops.to_device(e) for e in ep])
                torch.cuda.synchronize()
                parity = {"max_abs_sr_raw": float((out.cpu() - ref).abs().max()), "ref_absmax_sr_raw": float(ref.abs().max()),
                          "sample": "same %dx%d image and eps vs oracle" % (cl, cl)}
            else:
                med, ts, ref = median_time(lambda: O.lp_pipeline(x, sd, psd, opt, nb, return_all=True))
                what = "oracle lp_pipeline (reference op order, RRDB twice)"
                out = lp_infer(model, prior, x, return_all=True)
                torch.cuda.synchronize()
                parity = {"max_abs_sr": float((out["sr"].cpu() - ref["sr"]).abs().max()),
                          "max_abs_sr_raw": float((out["sr_raw"].cpu() - ref["sr_raw"]).abs().max()),
                          "ref_absmax_sr_raw": float(ref["sr_raw"].abs().max()), "sample": "same %dx%d image vs oracle" % (cl, cl)}
            hr = cl * scale
        else:
            import oracle.linf_ref as O            # checker / baseline only
            cl = args.cpu_lr or 96                 # bounded sample (the full 256x256 crop takes > 30 s per run on the host)
            hr = int(round(cl * scale))
            x = synth.lr_batch(99, 1, cl, cl)

            def run():
                return O.lp_pipeline(O.batch_prep(x, (hr, hr)), sd, psd, mspec, (hr, hr), return_all=True)
            med, ts, ref = median_time(run)
            what = "oracle LINF lp_pipeline (input prep + encoder + query_log_p + prior + query_rgb, 256-row chunks)"
            out = infer_from_lr(model, prior, x, scale, return_all=True)
            torch.cuda.synchronize()
            parity = {"max_abs_pred": float((out["pred"].cpu() - ref["pred"]).abs().max()),
                      "max_abs_z_lr": float((out["z_lr"].cpu() - ref["z_lr"]).abs().max()),
                      "ref_absmax_z_lr": float(ref["z_lr"].abs().max()), "precision": "fp32" if cfg == 3 else "fp16 MFMA path vs fp32 oracle",
                      "sample": "same %dx%d image vs oracle" % (cl, cl)}
        cpu_baseline = {"value": round(hr * hr / 1e6 / med, 5), "unit": "MPix/s", "cores": nt, "host_cpu_count": ncpu, "kind": "port",
                        "sample": "1 image %dx%d->%dx%d, %s; 1 warm-up + median of 3 runs (%s s) at torch.set_num_threads(%d)"
                                  % (cl, cl, hr, hr, what, ", ".join("%.1f" % t for t in ts), nt)}

    if rank == 0:
        # parity gate: an fp32-configuration line is only printed when the HIP path agrees with the oracle on the sample (north_star: 1e-4)
        if parity is not None and cfg in (2, 3, 4):
            bad = {k: v for k, v in parity.items() if k.startswith("max_abs") and k in ("max_abs_sr", "max_abs_pred") and not (v <= 1e-4)}
            if bad:
                raise SystemExit("bench.py: parity failure vs the oracle, no line printed: %s" % bad)
        if srflow:
            name = "SRFlow-LP %dx flow-inverse SR (%d->%d)" % (scale, h, H)
            path = ("LP path: RRDB + encode + standardise + prior UNet + decode + clamp" if args.mode == "lp"
                    else "tau=0.9 sampling path (RRDB + decode, no encode/prior)")
            wl = "SRFlow-LP %dx %s (K=16,L=3,nb=23), " % (scale, "DF2K config" if scale == 4 else "config derived from the 4X yml (scale 8, L 3)")
            split_txt = ("two-term fp16 split of both operands (hi + lo = 22 significant bits, weights pre-scaled by a power of two so that "
                         "their lo terms stay normal), 3 cross products lo*hi + hi*lo + hi*hi on the fp16 MFMA, fp32 accumulation: error of "
                         "the whole pipeline against an fp64 evaluation = the CPU fp32 evaluation's "
                         "(tests/test_srflow_gpu.py::test_end_to_end_error_vs_fp64, tests/test_hip_ops.py::test_conv_h2x_is_fp32_accurate); "
                         "RRDB block activations are STORED as that hi + lo pair" if getattr(ops, "split", "bf16x3") == "f16x2" else
                         "exact 3-term bf16 split of both operands (6 cross products on the bf16 MFMA, error vs fp64 = native fp32 MFMA's, "
                         "tests/test_hip_ops.py::test_conv_bf16x3_is_fp32_accurate; RRDB block activations are STORED as that exact split = lossless)")
            arithmetic = ("fp32 tensors and accumulation; 3x3 convs with >=32 input channels contract on the 16-bit matrix pipe with the "
                          + split_txt + "; everything else native fp32" if ops.conv_mode == "x3" else "native fp32 MFMA / fp32 VALU")
            dtype = ("f32" if ops.conv_mode != "x3" else
                     "f32 (fp16x2-split operands: 22-bit hi + lo pairs on the fp16 MFMA, fp32 accumulate)" if getattr(ops, "split", "") == "f16x2" else
                     "f32 (bf16x3-split operands: exact 24-bit triples on the bf16 MFMA, fp32 accumulate)")
        else:
            name = "LINF-LP rrdb-linf-LP x%g arbitrary-scale SR (%d->%d)" % (scale, h, H)
            path = "LP path: input prep + RRDB encoder + query_log_p + prior UNet + query_rgb + fold + skip + clamp"
            wl = "LINF-LP rrdb-linf-LP, "
            arithmetic = ("fp32-accurate split contraction on the 16-bit matrix pipe (BFSR_SPLIT=%s, see config 2)" % getattr(ops, "split", "bf16x3") if cfg == 3 else
                          "fp16 MFMA path (BASELINE config 5): encoder / coef|freq / MLP / prior contractions round their operands to fp16, "
                          "fp32 accumulation, fp32 tensors, the flow itself in fp32; tolerance vs the fp32 reference 1e-3 on the output "
                          "(tests/test_linf_gpu.py::test_fp16_mfma_path_vs_reference_golden)")
            dtype = (("f32 (fp16x2-split operands: 22-bit hi + lo pairs on the fp16 MFMA, fp32 accumulate)" if getattr(ops, "split", "") == "f16x2" and ops.conv_mode == "x3"
                      else "f32") if cfg == 3 else "f16")
        batch_txt = ("batch=%d/GPU" % B) if scaling == "weak" else ("named batch of %d crops sharded over %d GPU(s) = %d/GPU" % (global_B, world, B))
        line = {
            "metric": "HR MPix/s, %s, %s" % (name, "LP pipeline" if args.mode == "lp" or not srflow else path),
            "value": round(value, 4), "unit": "MPix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_ms, 3), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": dtype, "data": "synthetic", "arithmetic": arithmetic,
            "config": {"baseline_config": cfg,
                       "workload": wl + "%s %dx%d LR synthetic -> %dx%d, %s%s" % (
                           batch_txt, h, h, H, H, path, ", + double-buffered RCCL all-gather of outputs" if world > 1 else ""),
                       "parallelism": "dp%d" % world, "global_batch": global_B, "weights": "seeded synthetic (conditioned recipe)"},
            "roofline": roofline, "roofline_next_kernels": roofline_next, "roofline_by_symbol": by_symbol,
            "roofline_by_symbol_note": "HIP-event time per kernel family in the untimed ranking step; with the side stream active (configs "
                                       "whose batch does not fill the chip) events of overlapping kernels are inflated by contention and the "
                                       "families sum to more than ms_per_step -- profiles/r06k_keys_cfg2_no_overlap.txt has the BFSR_OVERLAP=0 table",
            "roofline_coupling_inverse": roof_tail,
            "cpu_baseline": cpu_baseline, "parity": parity,
            # passes that the range guard of the fp16-pair split re-ran under the bf16x3 split (bfsr_amd/guard.py); the guard's 4-byte read-back at
            # the end of every pass is INSIDE the timed region
            "fallbacks": int(getattr(ops, "fallbacks", 0)),
            "host": {"torch_threads": host_threads, "effective_cpus": hostenv.effective_cpus(), "cpu_count": os.cpu_count()},
        }
        if fp32_only is not None or cfg == 2:
            line["value_native_fp32_mfma"] = fp32_only
        print(json.dumps(line))
    if torch.distributed.is_initialized():              # world > 1, or a single rank under BFSR_DIST_FORCE=1
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
