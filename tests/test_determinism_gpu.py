"""-m gpu: run-to-run bitwise determinism -- of every MFMA kernel alone (tools/determinism_stress.py, 30 launches each) and of the
engines' whole kernel sequences (100 rounds; the opt-in soak test_soak_* runs 10^5 launches): timing-dependent faults show in a fraction of the launches, which one parity run can miss
("the head fault")."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mfma_kernels_are_run_to_run_deterministic():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "determinism_stress.py"), "30"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "TOTAL differing launches: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_kernel_fuzz_random_shapes():
    """tools/fuzz_kernels.py: ~230 random shapes (1-pixel images, ragged tiles, odd channel counts, channel-slice views, 1-3 samples) over the conv
    families -- the register-staged split / fp16 / x2 / x4 / wide-1x1 kernels against the CPU double, the LDS-DMA family over h2 tensors (conv_h2x, a
    conv_chain of random length bit-identical to its launches, conv_up2_h2t, conv_up4_h2t) against fp64 convs of the same 22-bit inputs."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_kernels.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "mismatches: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("scale,lr_size,overlap,rounds,batch", [(4, 160, None, 100, 2), (4, 160, "0", 40, 2), (8, 96, None, 100, 2), (4, 160, None, 30, 8)])
def test_engine_is_reproducible_run_to_run(scale, lr_size, overlap, rounds, batch, monkeypatch):
    """The fault class that isolated kernel stress does not see (round 3: the 8-wave coupling_head wrote a wrong half row once in
    10^3-10^4 launches, only inside the engine's kernel sequence; study: tools/exp/head_fault.py).  100 rounds of
    encode(B=2) -> decode -> encode(sample 1) on fixed inputs -- ~6 500 head and tail launches per case; the opt-in soak below runs 2 x 10^5 -- must reproduce round 0 bit for bit,
    and the single-sample call must equal the batch call's sample.  Cases: the 4x model with and without the side stream, the 8x model
    (C = 12 / 24 levels at 384^2 / 192^2); and 30 rounds at the bench batch (B = 8: every persistent kernel loops over several items per
    workgroup there -- the regime in which the store-data hazard of DESIGN.md section 3 item 8 showed and B = 2 did not)."""
    import torch
    from bfsr_amd import synth
    from bfsr_amd.ops import HipOps, MODE_BILINEAR
    from test_srflow_gpu import build
    if overlap is not None:
        monkeypatch.setenv("BFSR_OVERLAP", overlap)
    hip = HipOps("cuda:0")
    m, prior, opt, sd, psd = build(hip, scale)
    eng = m.netG.module.engine()
    lr = hip.to_device(synth.smooth_lr_batch(21, batch, lr_size, lr_size))
    lr_up = hip.resize(lr, hip.empty(batch, 3, lr_size * scale, lr_size * scale), MODE_BILINEAR, 1.0 / scale, 1.0 / scale)
    lr1, lr_up1 = lr[1:2].clone(), lr_up[1:2].clone()
    ref2 = ref1 = rt0 = None
    for it in range(rounds):
        if it % 4 == 0:
            lr.add_(0.0)                   # same values, new tensor version: the conditioning (RRDB trunk -- the fused conv_chain launch at B = 8 --,
            lr1.add_(0.0)                  # taps kernels, hoists) is recomputed instead of served from the engine's cache
        ep = [e.clone() for e in eng.encode(lr_up, lr)]
        rt = eng.decode(lr, epses=[e.clone() for e in ep]).clone()
        ep1 = [e.clone() for e in eng.encode(lr_up1, lr1)]
        if ref2 is None:
            ref2, ref1, rt0 = ep, ep1, rt
            continue
        for lvl in range(len(ep)):
            assert torch.equal(ep[lvl], ref2[lvl]), "round %d: encode(B=%d) eps%d differs from round 0" % (it, batch, lvl)
            assert torch.equal(ep1[lvl], ref1[lvl]), "round %d: encode(B=1) eps%d differs from round 0" % (it, lvl)
            assert torch.equal(ep[lvl][1:2], ep1[lvl]), "round %d: eps%d of sample 1 depends on the batch" % (it, lvl)
        assert torch.equal(rt, rt0), "round %d: decode differs from round 0" % it
    hip.check_range()


def test_plain_kernels_next_to_an_mfma_kernel_on_another_stream():
    """The co-residency case behind the packed-fp32 finding, kernel by kernel: the 1x1-only coupling_head (the one MFMA kernel of the library that leaves registers
    and LDS free on its SIMDs) loops on the main stream while a plain kernel runs on a side stream; the plain kernel's result must equal what it computes alone.
    (On a library built WITH packed fp32 the bilinear resize fails this in every run: tools/exp/victim_probe.py, profiles/r05_packed_fp32_hazard.txt.)"""
    import numpy as np
    import torch
    from bfsr_amd.ops import HipOps, MODE_BILINEAR, MODE_BILINEAR_AC
    ops = HipOps("cuda:0")
    g = np.random.Generator(np.random.PCG64(3))
    r = lambda *sh, scale=1.0: torch.from_numpy((g.standard_normal(sh) * scale).astype(np.float32))
    B = 32
    hp1 = ops.pack_coupling_head(None, r(64, 64, 1, 1, scale=0.1), r(64, scale=0.1), torch.exp(r(64, scale=0.1)), r(64, scale=0.1), torch.exp(r(64, scale=0.1)))
    raw, h2b = torch.randn(B, 64, 96, 96, device="cuda"), ops.h2_empty(B, 64, 96, 96)
    aggr = lambda: ops.coupling_head(None, hp1, raw, h2b, pre_fmt=0)
    bottom, cat = torch.randn(B, 256, 48, 48, device="cuda"), ops.empty(B, 512, 96, 96)
    b96, c192 = torch.randn(B, 64, 96, 96, device="cuda"), ops.empty(B, 64, 192, 192)
    z, zo = torch.randn(B, 24, 192, 192, device="cuda"), ops.empty(B, 24, 192, 192)
    haff, hft = torch.randn(B, 24, 192, 192, device="cuda") * 0.5, torch.randn(B, 48, 192, 192, device="cuda") * 0.5
    Wm = torch.from_numpy(np.linalg.qr(g.standard_normal((24, 24)))[0].astype(np.float32))
    wv, wt, ab, ae = ops.vec(Wm), ops.vec(Wm.t().contiguous()), ops.vec(r(24, scale=0.1)), ops.vec(torch.exp(r(24, scale=0.1)))
    e6, n6 = torch.randn(B, 6, 384, 384, device="cuda"), ops.empty(B, 6, 384, 384)
    p48, sq = ops.empty(B, 64, 48, 48), ops.empty(B, 96, 96, 96)
    xh, f96 = ops.h2_pack(b96, ops.h2_empty(B, 64, 96, 96)), ops.empty(B, 64, 96, 96)
    victims = {
        "resize bilinear (align_corners) 48 -> 96": lambda: ops.resize(bottom, cat[:, 256:], MODE_BILINEAR_AC, 47.0 / 95.0, 47.0 / 95.0, window=(0, 0, 96, 96)),
        "resize bilinear 96 -> 192": lambda: ops.resize(b96, c192, MODE_BILINEAR, 0.5, 0.5),
        "flow_pointwise C = 24 (reverse, both conditionals)": lambda: ops.flow_pointwise(z, zo, True, h_aff=haff, h_ft=hft, w=wv, wt=wt, an_bias=ab, an_escale=ae),
        "standardize 6 ch @384^2": lambda: ops.standardize(e6, n6),
        "maxpool2 96 -> 48": lambda: ops.maxpool2(b96, p48),
        "squeeze2d 24 ch @192^2": lambda: ops.squeeze2d(z, sq),
        "axpb_clamp 64 ch @192^2": lambda: ops.axpb_clamp(c192, c192, 1.0, 0.0, -3.0, 3.0),
        "h2_unpack 64 ch @96^2": lambda: ops.h2_unpack(xh, f96),
    }
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    aggr()
    torch.cuda.synchronize()
    for name, fn in victims.items():
        ref = fn().clone()
        torch.cuda.synchronize()
        for rep in range(3):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(4):
                    out = fn()
                ev = side.record_event()
            while not ev.query():
                for _ in range(8):
                    aggr()
            torch.cuda.synchronize()
            bad = int((out != ref).sum())
            assert bad == 0, "%s: %d elements differ from the result computed alone (overlapped run %d)" % (name, bad, rep)


@pytest.mark.parametrize("scale,batch,size", [(8, 16, 96), (4, 8, 160)])
def test_lp_pass_is_reproducible_run_to_run_with_every_overlap(scale, batch, size):
    """The WHOLE LP pass (RRDB, encode, standardise, both prior branches, decode) four times on the same input, conditioning recomputed every time, every
    intermediate latent and the image bit-identical -- at batches where all the stream overlaps are active at once (hoists and the prior's big branch on the
    side stream, level-3 lanes, the fused RRDB launch).  Round 5 found the bilinear resize of the prior's branch 0 wrong in a few hundred elements per
    pass here (8x model, B >= 16): its packed-fp32 instruction chains went wrong when their waves shared SIMDs with the MFMA waves of the 1x1-only
    coupling_head on the other stream; the engine-level reproducibility tests above never run the prior and did not see it."""
    import torch
    from bfsr_amd import synth
    from bfsr_amd.ops import HipOps
    from bfsr_amd.srflow.test import lp_infer
    from test_srflow_gpu import build
    hip = HipOps("cuda:0")
    m, prior, opt, sd, psd = build(hip, scale)
    x = hip.to_device(synth.lr_batch(1, batch, size, size))
    ref = None
    for it in range(int(os.environ.get("BFSR_LP_PASSES", "4"))):          # BFSR_LP_PASSES=<n>: the same check as a soak (profiles/r05_lp_soak.txt)
        x.add_(0.0)
        out = lp_infer(m, prior, x, return_all=True)
        cur = [out["sr"].clone(), out["sr_raw"].clone()] + [e.clone() for e in out["epses"]] + [e.clone() for e in out["epses_learned"]]
        if ref is None:
            ref = cur
            continue
        for i, (a, b) in enumerate(zip(cur, ref)):
            assert torch.equal(a, b), "pass %d: tensor %d of (sr, sr_raw, epses..., epses_learned...) differs from pass 0 in %d elements" % (it, i, int((a != b).sum()))
    assert hip.fallbacks == 0


@pytest.mark.parametrize("precision,rounds", [("fp32", 25), ("fp16", 15)])
def test_linf_pipeline_is_reproducible_run_to_run(precision, rounds):
    """The same guard for the LINF-LP path, in the fp32-accurate mode (conv_h2x encoder, fused MLP on the fp16 pair, flow) and on the fp16 MFMA
    path of BASELINE config 5 (conv_h2s encoder, linf_mlp<fp16>, conv_f16): LP passes over a fixed batch reproduce the first one bit for bit."""
    import torch
    from bfsr_amd import synth
    from bfsr_amd.ops import HipOps
    from bfsr_amd.linf import prep
    from bfsr_amd.linf.test import lp_infer
    from test_linf_gpu import _bench_size_models
    hip = HipOps("cuda:0")
    m, prior, sd, psd = _bench_size_models(hip, precision)
    lr = hip.to_device(synth.smooth_lr_batch(44, 4, 96, 96))
    batch = prep.prepare_batch(hip, lr, (384, 384), 3, True)
    ref = None
    for it in range(rounds):
        out = lp_infer(m, prior, batch, (384, 384), return_all=True)
        cur = {k: out[k].clone() for k in ("z_lr", "z_learned", "pred")}
        if ref is None:
            ref = cur
        for k in cur:
            assert torch.equal(cur[k], ref[k]), "round %d: %s differs from round 0" % (it, k)


@pytest.mark.skipif(not os.environ.get("BFSR_SOAK"), reason="opt-in soak (BFSR_SOAK=<rounds>, minutes of GPU time): see DESIGN.md section 5, round 5")
def test_soak_coupling_pair_inside_the_engine_sequence():
    """coupling_tail_kernel<0,12,*> has the two ingredients of the round-3 head fault's signature (VGPR-returning loads
    landing under a dependent MFMA chain; two barrier-locked waves per SIMD).  300 engine rounds are evidence for ~2 x 10^4 launches; the round-3
    fault needed 10^3-10^4 launches per event.  This soak runs BFSR_SOAK rounds (default use: 6500 = 2 x 10^5 level-1 tail launches and as many
    head launches, the same again at level 2) of encode + decode on fixed inputs INSIDE the engine's kernel sequence and compares every round's
    latents and image with round 0 on the device, bit for bit.  Results per box: profiles/r05_soak_*.txt."""
    import torch
    from bfsr_amd import synth
    from bfsr_amd.ops import HipOps, MODE_BILINEAR
    from test_srflow_gpu import build
    rounds = int(os.environ["BFSR_SOAK"])
    # BFSR_SOAK_CFG="scale,batch,lr" soaks another shape, e.g. "8,8,96": the 8x model (conv_up4_h2t, level-1 pair at 384^2) at a batch where the
    # fused RRDB launch (conv_chain), the level-3 lanes and several items per persistent workgroup are active
    scale, batch, size = (int(v) for v in os.environ.get("BFSR_SOAK_CFG", "4,2,160").split(","))
    hip = HipOps("cuda:0")
    m, prior, opt, sd, psd = build(hip, scale)
    eng = m.netG.module.engine()
    lr = hip.to_device(synth.smooth_lr_batch(21, batch, size, size))
    lr_up = hip.resize(lr, hip.empty(batch, 3, size * scale, size * scale), MODE_BILINEAR, 1.0 / scale, 1.0 / scale)
    ref_ep = ref_rt = None
    bad = 0
    recond = int(os.environ.get("BFSR_SOAK_RECOND", "0"))       # N > 0: recompute the conditioning (RRDB / conv_chain, taps kernels, hoists) every N-th round
    for it in range(rounds):
        if recond and it % recond == 0:
            lr.add_(0.0)
        ep = eng.encode(lr_up, lr)
        rt = eng.decode(lr, epses=ep)
        if ref_ep is None:
            ref_ep, ref_rt = [e.clone() for e in ep], rt.clone()
            continue
        same = torch.equal(rt, ref_rt) and all(torch.equal(a, b) for a, b in zip(ep, ref_ep))
        if not same:
            bad += 1
            print("round %d differs from round 0" % it, flush=True)
    n_pair = sum(1 for ly in eng.layers if ly.type == "step" and ly.coupled and getattr(eng.steps[ly.index], "fused", False))
    print("soak (%dx model, B = %d, %d^2): %d rounds, %d fused coupled steps per direction -> %d head and %d tail launches; %d rounds differ" % (
        scale, batch, size, rounds, n_pair, 2 * n_pair * rounds, 2 * n_pair * rounds, bad), flush=True)
    hip.check_range()
    assert bad == 0
