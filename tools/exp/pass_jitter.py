"""Per-pass wall time distribution of eager LINF-LP passes (each pass synchronised), with the cgroup's CPU-throttling counters and the Python GC
on / off -- where do the occasional 30-40 ms host stalls come from?  Usage: python tools/exp/pass_jitter.py [--batch 16] [--passes 40]"""
import argparse, contextlib, gc, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def cpu_stat():
    for p in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        if os.path.exists(p):
            return {l.split()[0]: int(l.split()[1]) for l in open(p) if len(l.split()) == 2}
    return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--passes", type=int, default=40)
    a = ap.parse_args()
    from bfsr_amd import synth
    from bfsr_amd.ops import HipOps
    from bfsr_amd.linf import spec as lspec
    from bfsr_amd.linf.models import make
    from bfsr_amd.linf.test import infer_from_lr
    B, h, scale, precision = a.batch, 128, 6.0, "fp16"
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        if os.path.exists(p):
            print(p, open(p).read().strip())
    print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
    ops = HipOps("cuda:0")
    mspec = {"name": "linf-patch", "args": {"encoder_spec": {"name": "rrdb", "args": {"no_upsampling": True}},
                                             "imnet_spec": {"name": "flow", "args": {"name": "flow"}}, "flow_layers": 10, "num_layer": 3, "hidden_dim": 256}}
    with contextlib.redirect_stdout(sys.stderr):
        model = make(mspec, args={"ops": ops, "precision": precision}).eval()
        model.load_state_dict(synth.state_dict_from_schema(lspec.linf_schema(mspec["args"]["encoder_spec"]), 2024))
        prior = make({"name": "unet", "args": {"in_chans": 27, "depth": 3, "dim": 64, "bilinear": True}}, args={"ops": ops, "precision": precision}).eval()
        prior.load_state_dict(synth.state_dict_from_schema(lspec.linf_prior_schema(27), 777))
    x = ops.to_device(synth.lr_batch(1, B, h, h))
    for _ in range(3):
        x.add_(0.0)
        infer_from_lr(model, prior, x, scale)
    torch.cuda.synchronize()

    def run(tag, n):
        s0 = cpu_stat()
        ts = []
        for _ in range(n):
            x.add_(0.0)
            t0 = time.perf_counter()
            infer_from_lr(model, prior, x, scale)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            ts.append(((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
        s1 = cpu_stat()
        w = sorted(t[1] for t in ts)
        print("%s: wall median %.1f min %.1f max %.1f mean %.1f ms; host-enqueue median %.1f ms; throttled +%d periods (+%.1f ms)" % (
            tag, w[len(w) // 2], w[0], w[-1], sum(w) / len(w), sorted(t[0] for t in ts)[len(ts) // 2],
            s1.get("nr_throttled", 0) - s0.get("nr_throttled", 0), (s1.get("throttled_usec", 0) - s0.get("throttled_usec", 0)) / 1e3))
        print("   walls:", " ".join("%.0f" % t[1] for t in ts))

    run("gc on ", a.passes)
    gc.disable()
    run("gc off", a.passes)
    gc.enable()
    torch.set_num_threads(1)
    run("1 torch thread", a.passes)
    # the same pass as a HIP graph
    try:
        from bfsr_amd.linf.test import _lp_infer
        from bfsr_amd.linf import prep
        H = round(h * scale)
        g = torch.cuda.CUDAGraph()
        x.add_(0.0)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            batch = prep.prepare_batch(ops, x, (H, H), model.patch_size, True)
            _lp_infer(model, prior, batch, (H, H), 0, False)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            batch = prep.prepare_batch(ops, x, (H, H), model.patch_size, True)
            out = _lp_infer(model, prior, batch, (H, H), 0, False)
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.passes):
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        w = sorted(ts)
        print("graph replay: median %.1f min %.1f max %.1f mean %.1f ms" % (w[len(w) // 2], w[0], w[-1], sum(w) / len(w)))
        x.add_(0.0)
        ref = infer_from_lr(model, prior, x, scale)
        g.replay()
        torch.cuda.synchronize()
        print("graph == eager: max abs diff", float((out - ref).abs().max()))
    except Exception as e:  # noqa
        import traceback
        traceback.print_exc()
        print("graph capture failed:", repr(e)[:400])


main()
