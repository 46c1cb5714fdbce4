"""-m gpu: SRFlow-LP end-to-end parity through the drop-in API on the HIP kernels: committed golden vectors of
the genuine reference, the oracle on fresh seeded inputs, size-independent properties at larger sizes
(encode->decode round trip, batch-sharding invariance), and per-stage goldens."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from bfsr_amd import synth                      # noqa: E402
from bfsr_amd.srflow import options, spec       # noqa: E402


@pytest.fixture(scope="module")
def hip():
    from bfsr_amd.ops import HipOps
    return HipOps("cuda:0")


def build(hip, scale=4):
    from bfsr_amd.srflow.models import create_model, models as registry
    opt = options.load(options.DEFAULT_CONF)
    if scale != 4:
        opt = options.derive_scale(opt, scale)
    sd = synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234)
    psd = synth.state_dict_from_schema(spec.srflow_prior_schema(), 4321)
    m = create_model(opt, ops=hip)
    m.load_network(sd)
    prior = registry.make({"name": "unet", "args": {"depth": 3, "dim": 64, "bilinear": True, "ops": hip}, "sd": psd},
                          load_sd=True).eval()
    return m, prior, opt, sd, psd


@pytest.fixture(scope="module")
def model4(hip):
    return build(hip, 4)


def _chk(name, got, ref, tol=1e-4):
    got = got.detach().cpu()
    assert not torch.isnan(got).any(), name + " NaN"
    err = (got - ref).abs().max().item()
    bound = tol * max(1.0, ref.abs().max().item())
    assert err <= bound, "%s: max-abs %.3e > %.3e" % (name, err, bound)
    return err


@pytest.mark.parametrize("fx", ["srflow_e2e_4x_a", "srflow_e2e_4x_b"])
def test_golden_e2e_4x(model4, golden_dir, fx):
    from bfsr_amd.srflow.test import lp_infer
    m, prior, opt, sd, psd = model4
    g = np.load(os.path.join(golden_dir, fx + ".npz"))
    out = lp_infer(m, prior, torch.from_numpy(g["lr"]), return_all=True)
    T = lambda k: torch.from_numpy(g[k])
    for i in (0, 1):
        _chk("eps%d" % i, out["epses"][i], T("eps%d" % i))
        _chk("epsn%d" % i, out["epses_norm"][i], T("epsn%d" % i))
        _chk("epsl%d" % i, out["epses_learned"][i], T("epsl%d" % i))
    _chk("sr_raw", out["sr_raw"], T("sr_raw"))
    _chk("sr", out["sr"], T("sr"), 1e-4)       # north_star: <= 1e-4 max-abs on the [0,1] output


def test_reference_caller_sequence_through_srflowmodel(model4, golden_dir):
    """The LP block exactly as SRFlow-LP/code/test.py:139-148 drives it -- `model.get_encode_z(lr, lr_up, epses=[],
    add_gt_noise=False)` -> per-pixel standardisation -> `prior_model(epses)` -> `model.get_sr(lq=lr, epses=...)` -- through the
    SRFlowModel wrapper (not the engine-level `lp_infer`), against the reference golden."""
    import torch.nn.functional as F
    m, prior, opt, sd, psd = model4
    g = np.load(os.path.join(golden_dir, "srflow_e2e_4x_a.npz"))
    T = lambda k: torch.from_numpy(g[k])
    lr = T("lr")
    lr_up = F.interpolate(lr, scale_factor=opt["scale"], mode="bilinear", align_corners=False)      # test.py:137
    epses_lr = m.get_encode_z(lr, lr_up, epses=[], add_gt_noise=False)                                # test.py:139
    assert isinstance(epses_lr, list) and len(epses_lr) == 2
    for i in (0, 1):
        _chk("eps%d" % i, epses_lr[i], T("eps%d" % i))
    for idx, eps in enumerate(epses_lr):                                                              # test.py:141-145
        eps_mean = eps.mean(dim=1, keepdim=True)
        eps_std = eps.std(dim=1, keepdim=True)
        epses_lr[idx] = (eps - eps_mean) / (eps_std + 1e-8)
        _chk("epsn%d" % idx, epses_lr[idx], T("epsn%d" % idx))
    epses_learned = prior(epses_lr)                                                                   # test.py:147
    sr = m.get_sr(lq=lr, epses=epses_learned)                                                         # test.py:148
    assert len(epses_learned) == 2                          # decode copies the list, the caller's is not consumed
    _chk("sr_raw", sr, T("sr_raw"))
    _chk("sr", sr.clamp(0, 1), T("sr"), 1e-4)


def test_golden_e2e_8x(hip, golden_dir):
    from bfsr_amd.srflow.test import lp_infer
    m, prior, opt, sd, psd = build(hip, 8)
    g = np.load(os.path.join(golden_dir, "srflow_e2e_8x.npz"))
    out = lp_infer(m, prior, torch.from_numpy(g["lr"]), return_all=True)
    T = lambda k: torch.from_numpy(g[k])
    for i in (0, 1):
        _chk("eps%d" % i, out["epses"][i], T("eps%d" % i))
    _chk("sr_raw", out["sr_raw"], T("sr_raw"))
    _chk("sr", out["sr"], T("sr"))


def test_golden_rrdb_and_prior(model4, hip, golden_dir):
    m, prior, opt, sd, psd = model4
    eng = m.netG.module.engine()
    g = np.load(os.path.join(golden_dir, "srflow_rrdb.npz"))
    lr = hip.to_device(torch.from_numpy(g["lr"]))
    eng.conditioning(lr)
    for level, key in ((1, "fea_up2"), (2, "fea_up1"), (3, "fea_up0")):
        got = eng.ws.bufs["ft%d" % level]
        # a level consumed through conv_up2 keeps only its 64 key channels (the upsampled taps are never materialised)
        _chk(key, got, torch.from_numpy(g[key])[:, :got.shape[1]], 2e-5)
    p = np.load(os.path.join(golden_dir, "srflow_prior.npz"))
    for a, b, c, d in (("e0", "e1", "z0", "z1"), ("e0b", "e1b", "z0b", "z1b")):
        out = prior([torch.from_numpy(p[a]), torch.from_numpy(p[b])])
        _chk(c, out[0], torch.from_numpy(p[c]), 2e-5)
        _chk(d, out[1], torch.from_numpy(p[d]), 2e-5)


def test_prior_full_resolution_convs_on_h2x_vs_oracle(model4, hip, monkeypatch):
    """The prior's big branch with its nine full-resolution convs on conv_h2x (unet_engine.SRFlowPriorEngine._use_h2: active from 32 tiles of
    16 x 32 per sample) against the pinned oracle and against the register-staged path (BFSR_PRIOR=reg) on the same latents; odd sizes exercise the
    pad / window of the up path next to the h2 buffers."""
    import oracle.srflow_ref as O
    m, prior, opt, sd, psd = model4
    g = torch.Generator().manual_seed(5)
    e0, e1 = torch.randn(2, 6, 250, 290, generator=g), torch.randn(2, 96, 62, 72, generator=g)
    eng = prior.engine()
    assert eng._use_h2(hip.to_device(e0)) and not eng._use_h2(hip.to_device(e1))
    got = [t.cpu() for t in prior([e0, e1])]
    ref = O.srflow_prior([e0, e1], psd)
    for k in range(2):
        _chk("z%d" % k, got[k], ref[k], 2e-5)
    monkeypatch.setenv("BFSR_PRIOR", "reg")
    assert not eng._use_h2(hip.to_device(e0))
    reg = [t.cpu() for t in prior([e0, e1])]
    assert float((reg[0] - got[0]).abs().max()) <= 2e-5 * float(ref[0].abs().max())
    assert torch.equal(reg[1], got[1])
    assert hip.fallbacks == 0


def test_prior_h2_glue_equals_launches(model4, hip, monkeypatch):
    """Both branches of the prior with the h2 pooling / up-sampling kernels of round 6 and with the launches they replace (BFSR_PRIOR_GLUE=launches): the same bits,
    at a size where two levels of the big branch run on the LDS-DMA kernels and with odd sizes (pad windows)."""
    m, prior, opt, sd, psd = model4
    g = torch.Generator().manual_seed(6)
    for shp in (((2, 6, 320, 320), (2, 96, 80, 80)), ((1, 6, 250, 290), (1, 96, 62, 72))):
        e = [torch.randn(*s_, generator=g) for s_ in shp]
        res = {}
        for mode in ("fused", "launches"):
            monkeypatch.setenv("BFSR_PRIOR_GLUE", mode)
            res[mode] = [t.clone() for t in prior(e)]
        for k in range(2):
            assert torch.equal(res["fused"][k], res["launches"][k]), "z%d: max diff %.3e" % (k, float((res["fused"][k] - res["launches"][k]).abs().max()))


def test_vs_oracle_fresh_input_and_roundtrip(model4, hip):
    import oracle.srflow_ref as O
    from bfsr_amd.srflow.test import lp_infer
    m, prior, opt, sd, psd = model4
    lr = synth.smooth_lr_batch(11, 2, 24, 40)
    out = lp_infer(m, prior, lr, return_all=True)
    ref = O.lp_pipeline(lr, sd, psd, opt, 23, return_all=True)
    _chk("sr_raw", out["sr_raw"], ref["sr_raw"])
    _chk("sr", out["sr"], ref["sr"])
    # invertibility: decode(encode(x)) == x
    eng = m.netG.module.engine()
    rt = eng.decode(hip.to_device(lr), epses=out["epses"])
    _chk("roundtrip", rt, ref["lr_up"], 1e-4)


def test_roundtrip_and_batch_invariance_at_bench_size(model4, hip):
    """Size-independent properties at the BASELINE crop size (160x160 LR -> 640x640)."""
    from bfsr_amd.ops import MODE_BILINEAR
    m, prior, opt, sd, psd = model4
    eng = m.netG.module.engine()
    lr = hip.to_device(synth.smooth_lr_batch(21, 2, 160, 160))
    lr_up = hip.resize(lr, hip.empty(2, 3, 640, 640), MODE_BILINEAR, 0.25, 0.25)
    ep = eng.encode(lr_up, lr)
    assert ep[0].shape == (2, 6, 320, 320) and ep[1].shape == (2, 96, 80, 80)
    rt = eng.decode(lr, epses=ep)
    err = (rt - lr_up).abs().max().item()
    assert err <= 1e-4, "round trip %.3e" % err
    # per-sample independence: sample 1 alone gives bit-identical latents (exactness of batch sharding)
    lr1 = lr[1:2].clone()
    lr_up1 = lr_up[1:2].clone()
    ep1 = eng.encode(lr_up1, lr1)
    for lvl, (a, b) in enumerate(zip(ep, ep1)):
        if not torch.equal(a[1:2], b):
            df = (a[1:2] - b).abs()
            idx = torch.nonzero(df > 0)
            raise AssertionError("batch sharding changed the result: eps%d max |diff| %.3e in %d of %d elements, first at %s, channels %s"
                                 % (lvl, float(df.max()), idx.shape[0], df.numel(), idx[0].tolist(), sorted(set(idx[:, 1].tolist()))[:12]))


def test_lp_pass_at_bench_batch_vs_oracle(model4, hip):
    """The parity check bench.py makes, inside pytest: the LP pass over the BASELINE config-2 batch (B = 8 crops of 160 x 160 -> 640 x 640: the
    fused dense-block chain, the side stream and every persistent kernel looping over several items per workgroup are active only at this
    size) against the pinned oracle on ONE of the crops (the CPU oracle needs ~6 s per crop), <= 1e-4 max-abs on sr (north_star)."""
    import oracle.srflow_ref as O
    from bfsr_amd.srflow.test import lp_infer
    m, prior, opt, sd, psd = model4
    lr = synth.smooth_lr_batch(77, 8, 160, 160)
    out = lp_infer(m, prior, lr, return_all=True)
    pick = 5
    ref = O.lp_pipeline(lr[pick:pick + 1], sd, psd, opt, 23, return_all=True)
    for k in ("sr_raw", "sr"):
        err = float((out[k][pick:pick + 1].cpu() - ref[k]).abs().max())
        assert err <= 1e-4, "%s of crop %d in the B = 8 batch: max-abs %.3e vs the oracle" % (k, pick, err)
    assert m.netG.module.engine().ops.fallbacks == 0


def test_lp_pass_8x_at_config4_crop_vs_oracle(hip):
    """The 8x model at BASELINE config 4's crop size (96 x 96 -> 768 x 768, a 4-crop shard): at this size the x4 level runs on conv_up4_h2t
    with the compact hand-over, level 3 on the wide pair and the prior's big branch on the LDS-DMA kernels -- kernel choices the 16 x 16
    reference golden (test_golden_e2e_8x) is too small to reach.  One crop against the pinned oracle, <= 1e-4 max-abs on sr (north_star)."""
    import oracle.srflow_ref as O
    from bfsr_amd.srflow.test import lp_infer
    m, prior, opt, sd, psd = build(hip, 8)
    lr = synth.smooth_lr_batch(78, 4, 96, 96)
    out = lp_infer(m, prior, lr, return_all=True)
    pick = 2
    ref = O.lp_pipeline(lr[pick:pick + 1], sd, psd, opt, 23, return_all=True)
    for k in ("sr_raw", "sr"):
        err = float((out[k][pick:pick + 1].cpu() - ref[k]).abs().max())
        assert err <= 1e-4, "%s of crop %d of the 8x shard: max-abs %.3e vs the oracle" % (k, pick, err)
    assert m.netG.module.engine().ops.fallbacks == 0


def test_roundtrip_and_batch_invariance_at_config4_size(hip):
    """BASELINE config 4's per-GPU batch on the 8x model (B = 8, 96x96 LR -> 768x768): decode(encode(x)) = x within 1e-4, the LP pass is
    finite, and a shard of the batch gives bit-identical latents (what the data-parallel split of the 64-crop batch relies on)."""
    from bfsr_amd.ops import MODE_BILINEAR
    from bfsr_amd.srflow.test import lp_infer
    m, prior, opt, sd, psd = build(hip, 8)
    eng = m.netG.module.engine()
    lr = hip.to_device(synth.smooth_lr_batch(61, 8, 96, 96))
    lr_up = hip.resize(lr, hip.empty(8, 3, 768, 768), MODE_BILINEAR, 0.125, 0.125)
    ep = [e.clone() for e in eng.encode(lr_up, lr)]
    rt = eng.decode(lr, epses=[e.clone() for e in ep])
    err = (rt - lr_up).abs().max().item()
    assert err <= 1e-4, "round trip %.3e" % err
    lo, hi_ = 2, 5                                        # a middle shard
    lrs, lrus = lr[lo:hi_].clone(), lr_up[lo:hi_].clone()
    eps_s = eng.encode(lrus, lrs)
    for lvl, (a, b) in enumerate(zip(ep, eps_s)):
        assert torch.equal(a[lo:hi_], b), "eps%d of a shard differs from the batch call: max %.3e" % (lvl, float((a[lo:hi_] - b).abs().max()))
    sr = lp_infer(m, prior, lr)
    sr_s = lp_infer(m, prior, lrs)
    assert torch.isfinite(sr).all() and sr.shape == (8, 3, 768, 768)
    assert torch.equal(sr[lo:hi_], sr_s), "LP output of a shard differs from the batch call"


def test_x4_level_compact_taps_equals_pre_add(hip, monkeypatch):
    """The x4 level of the 8x model (RRDBNet_arch.py:105-117 feeding FlowUpsamplerNet level 1): the default form (conv_up4_h2t writes nine class
    values per source pixel, the key conv expands and adds them: BFSR_UP4C=1) against the read-modify-write form (=0): the hoisted conditioning
    tensors and the LP output must be the same BITS."""
    from bfsr_amd.srflow.test import lp_infer
    m, prior, opt, sd, psd = build(hip, 8)
    eng = m.netG.module.engine()
    lr = hip.to_device(synth.smooth_lr_batch(21, 3, 96, 96))
    res = {}
    for v in ("1", "0"):
        monkeypatch.setenv("BFSR_UP4C", v)
        lr.add_(0.0)                                      # defeats the conditioning cache
        c1 = eng._await(eng.conditioning(lr)[1])
        torch.cuda.synchronize()
        res[v] = {k: c1[k].clone() for k in ("pre_aff", "h_ft")}
        res[v]["sr"] = lp_infer(m, prior, lr).clone()
    for k in ("pre_aff", "h_ft", "sr"):
        assert torch.equal(res["1"][k], res["0"][k]), "%s differs between the compact and the pre_add form: max %.3e" % (
            k, float((res["1"][k] - res["0"][k]).abs().max()))


def test_tau_path_runs(model4, hip):
    """Non-LP sampling path (get_z + Split2d sampling): finite output of the right shape."""
    m, prior, opt, sd, psd = model4
    lr = synth.lr_batch(31, 1, 16, 16)
    sr, z = m.get_sr_with_z(lr, heat=0.5, seed=3)
    assert sr.shape == (1, 3, 64, 64) and torch.isfinite(sr).all()
    assert z.shape == (1, 96, 8, 8)


def test_full_div2k_sized_image_roundtrip(model4, hip):
    """Maximum-size case: a DIV2K-validation-sized LR image (340x510 -> 1360x2040, batch 1 like the reference's test.py).
    The hoisted level-1 tensors exceed 2 GiB here; decode(encode(x)) must still return x."""
    from bfsr_amd.ops import MODE_BILINEAR
    m, prior, opt, sd, psd = model4
    eng = m.netG.module.engine()
    lr = hip.to_device(synth.smooth_lr_batch(51, 1, 340, 510))
    lr_up = hip.resize(lr, hip.empty(1, 3, 1360, 2040), MODE_BILINEAR, 0.25, 0.25)
    ep = eng.encode(lr_up, lr)
    assert ep[0].shape == (1, 6, 680, 1020) and ep[1].shape == (1, 96, 170, 255)
    rt = eng.decode(lr, epses=ep)
    err = (rt - lr_up).abs().max().item()
    assert torch.isfinite(rt).all() and err <= 1e-4, "round trip %.3e" % err
    eng.ws.bufs.clear()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("tag", ["a", "b"])
def test_nll_and_logdet_golden(model4, golden_dir, tag):
    """forward(reverse=False) -> (epses, nll, logdet), forward(reverse=True) -> (sr, logdet): the reference's values
    (tests/golden/srflow_logdet.npz, generated from the genuine SRFlowNet), relative tolerance 1e-5."""
    import torch.nn.functional as F
    m, prior, opt, sd, psd = model4
    net = m.netG.module
    g = np.load(os.path.join(golden_dir, "srflow_logdet.npz"))
    lr = torch.from_numpy(g["lr_" + tag])
    lr_up = F.interpolate(lr, scale_factor=4, mode="bilinear", align_corners=False)
    epses, nll, logdet = net(gt=lr_up, lr=lr, reverse=False, epses=[], add_gt_noise=False)
    rel = lambda a, b: float(((a.cpu() - b).abs() / b.abs().clamp_min(1.0)).max())
    assert rel(nll, torch.from_numpy(g["nll_" + tag])) <= 1e-5, (nll, g["nll_" + tag])
    assert rel(logdet, torch.from_numpy(g["logdet_" + tag])) <= 1e-5, (logdet, g["logdet_" + tag])
    sr, logdet_rev = net(lr=lr, reverse=True, epses=list(epses))
    assert rel(logdet_rev, torch.from_numpy(g["logdet_rev_" + tag])) <= 1e-5


def test_nll_vs_oracle_fresh_input(model4):
    import oracle.srflow_ref as O
    import torch.nn.functional as F
    m, prior, opt, sd, psd = model4
    lr = synth.smooth_lr_batch(77, 2, 24, 16)
    lr_up = F.interpolate(lr, scale_factor=4, mode="bilinear", align_corners=False)
    z, nll, logdet = m.netG.module(gt=lr_up, lr=lr, reverse=False, add_gt_noise=False)
    oe, onll, old = O.srflow_normal_flow(lr_up, lr, sd, opt, 23)
    _chk("z", z, oe[-1])
    assert ((nll.cpu() - onll).abs() / onll.abs()).max() <= 1e-5
    assert ((logdet.cpu() - old).abs() / old.abs()).max() <= 1e-5


def test_end_to_end_error_vs_fp64_at_config_size(hip):
    """The same evidence AT THE SIZE OF BASELINE CONFIG 2 ("f32 at the config size is argued, not tested"): one 160 x 160
    crop of the bench's own input generator through the default pipeline (two-term fp16 split, whole RRDB trunk as one conv_chain launch when the
    batch allows, the fused coupling pair, conv_up2_h2t) against an fp64 evaluation of the oracle.  K of the contractions is what it is in the
    bench (Cin * 9 up to 2 880 + 576 in the level-1 hoist), the image is 100x the pixels of the 16 x 16 test.  The HIP pipeline's error must stay
    within 2x of the CPU fp32 evaluation's own error against the same truth -- i.e. it IS an fp32-class evaluation, not a reduced-precision one."""
    import oracle.srflow_ref as O
    from bfsr_amd.srflow.models import create_model, models as registry
    from bfsr_amd.srflow.test import lp_infer
    opt = options.load(options.DEFAULT_CONF)
    sd = synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234)
    psd = synth.state_dict_from_schema(spec.srflow_prior_schema(), 4321)
    lr = synth.lr_batch(1000, 1, 160, 160)                 # rank 0 / batch 0 of bench.py
    dbl = lambda m: {k: (v.double() if v.is_floating_point() else v) for k, v in m.items()}
    truth = O.lp_pipeline(lr.double(), dbl(sd), dbl(psd), opt, 23, return_all=True)
    cpu32 = O.lp_pipeline(lr, sd, psd, opt, 23, return_all=True)
    assert hip.conv_mode == "x3" and hip.split == "f16x2", "this test is about the default arithmetic"
    m = create_model(opt, ops=hip)
    m.load_network(sd)
    prior = registry.make({"name": "unet", "args": {"depth": 3, "dim": 64, "bilinear": True, "ops": hip}, "sd": psd}, load_sd=True).eval()
    n0 = hip.fallbacks
    out = lp_infer(m, prior, lr, return_all=True)
    assert hip.fallbacks == n0, "the range guard re-ran the pass under bf16x3: this would not test the fp16 pair"
    err = {k: float((out[k].cpu().double() - truth[k]).abs().max()) for k in ("sr_raw", "sr")}
    err["z"] = float((out["epses"][1].cpu().double() - truth["epses"][1]).abs().max())
    ref = {k: float((cpu32[k].double() - truth[k]).abs().max()) for k in ("sr_raw", "sr")}
    ref["z"] = float((cpu32["epses"][1].double() - truth["epses"][1]).abs().max())
    print("160 x 160, max-abs error vs the fp64 oracle: HIP f16x2 %s | CPU fp32 %s" % (err, ref))
    for k in ("sr_raw", "sr", "z"):
        assert err[k] <= 2.0 * ref[k] + 1e-7, (k, err, ref)
        assert err[k] <= 1e-4


def test_end_to_end_error_vs_fp64(hip):
    """Pipeline-level evidence for the split contractions: against an fp64 run of the oracle (ground truth) the HIP pipeline's error
    with the two-term fp16 split (the default: 3 products, BFSR_SPLIT=f16x2) and with the three-term bf16 split (6 products) is at
    the level of the native-fp32-MFMA pipeline's and of the CPU fp32 oracle's own error."""
    import oracle.srflow_ref as O
    from bfsr_amd.ops import HipOps
    from bfsr_amd.srflow.models import create_model, models as registry
    from bfsr_amd.srflow.test import lp_infer
    opt = options.load(options.DEFAULT_CONF)
    sd = synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234)
    psd = synth.state_dict_from_schema(spec.srflow_prior_schema(), 4321)
    lr = synth.smooth_lr_batch(3, 1, 16, 16)              # the two CPU oracle runs (fp64, fp32) dominate this test's time
    dbl = lambda m: {k: (v.double() if v.is_floating_point() else v) for k, v in m.items()}
    truth = O.lp_pipeline(lr.double(), dbl(sd), dbl(psd), opt, 23, return_all=True)
    cpu32 = O.lp_pipeline(lr, sd, psd, opt, 23, return_all=True)
    errs = {}
    for mode in ("f16x2", "bf16x3", "f32"):
        ops = HipOps("cuda:0")
        ops.conv_mode = "f32" if mode == "f32" else "x3"
        if mode != "f32":
            ops.split = mode
        m = create_model(opt, ops=ops)
        m.load_network(sd)
        prior = registry.make({"name": "unet", "args": {"depth": 3, "dim": 64, "bilinear": True, "ops": ops}, "sd": psd},
                              load_sd=True).eval()
        out = lp_infer(m, prior, lr, return_all=True)
        errs[mode] = {k: float((out[k].cpu().double() - truth[k]).abs().max()) for k in ("sr_raw", "sr")}
        errs[mode]["z"] = float((out["epses"][1].cpu().double() - truth["epses"][1]).abs().max())
    errs["cpu32"] = {k: float((cpu32[k].double() - truth[k]).abs().max()) for k in ("sr_raw", "sr")}
    errs["cpu32"]["z"] = float((cpu32["epses"][1].double() - truth["epses"][1]).abs().max())
    print("max-abs error vs fp64 oracle:", errs)
    for mode in ("f16x2", "bf16x3"):
        for k in ("sr_raw", "sr", "z"):
            assert errs[mode][k] <= 2.0 * max(errs["f32"][k], errs["cpu32"][k]) + 1e-7, (mode, k, errs)
            assert errs[mode][k] <= 1e-4
