"""-m gpu: the range guard of the two-term fp16 split is unconditional (bfsr_amd/guard.py): every public entry point
either returns the right answer or raises -- never inf / NaN from an overflow, never a silently wrong value.

The hostile models are EXACT rescalings of the seeded synthetic ones (powers of two, so the rescaled network is the same function up to
rounding): the RRDB trunk's activations are multiplied by S = 2^17 (beyond fp16's 65504: the fp16 pair cannot hold them) or by 2^-17 (far
below the 2^-3 at which the lo term of the pair goes subnormal) and every consumer of the trunk's outputs is divided by S.  An arithmetic
with fp32's exponent range does not care; the fp16 pair must notice and the pass must be re-run under the bf16x3 split.  Truth = the CPU
oracle on the same rescaled weights."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from bfsr_amd import synth                      # noqa: E402
from bfsr_amd.srflow import options, spec       # noqa: E402


def rescale_trunk(sd, opt, S):
    """RRDB trunk activations x S (conv_first and every bias of the trunk; LeakyReLU and the residual sums are positively homogeneous), the
    consumers / S: upconv1 (-> fea_up2 at its old scale) and the conditioning rows of fFeatures.0 / fAffine.0 that read trunk outputs -- the 256
    stacked tap channels at every level, and the 64 key channels at the levels whose key IS the trunk output (fea_up1, fea_up0)."""
    sd = {k: v.clone() for k, v in sd.items()}
    for k in sd:
        if k in ("RRDB.conv_first.weight", "RRDB.conv_first.bias", "RRDB.trunk_conv.bias") or (k.startswith("RRDB.RRDB_trunk.") and k.endswith(".bias")):
            sd[k] = sd[k] * S
    sd["RRDB.upconv1.weight"] = sd["RRDB.upconv1.weight"] / S
    names = spec.level_to_name(opt["scale"])
    for ly in spec.flow_layers(opt):
        if ly.type != "step" or not ly.coupled:
            continue
        lo = 64 if names[ly.level] in ("fea_up2", "fea_up4", "fea_up8") else 0          # keys produced by the upconv chain keep their scale
        p = "flowUpsamplerNet.layers.%d.affine." % ly.index
        wf, wa = sd[p + "fFeatures.0.weight"], sd[p + "fAffine.0.weight"]
        nz = wa.shape[1] - 320
        wf[:, lo:] /= S
        wa[:, nz + lo:] /= S
    return sd


def rescale_key_channels(sd, opt, chans, s):
    """A few channels of fea_up2 (upconv1's output, the 64 key channels of level 1) sit a factor s below the rest: those rows of upconv1
    (weight and bias; the LeakyReLU behind it is positively homogeneous) x s, the rows of the level-1 conditioning convs that read them / s."""
    sd = {k: v.clone() for k, v in sd.items()}
    idx = torch.tensor(list(chans))
    sd["RRDB.upconv1.weight"][idx] *= s
    sd["RRDB.upconv1.bias"][idx] *= s
    for ly in spec.flow_layers(opt):
        if ly.type == "step" and ly.coupled and spec.level_to_name(opt["scale"])[ly.level] == "fea_up2":
            p = "flowUpsamplerNet.layers.%d.affine." % ly.index
            wf, wa = sd[p + "fFeatures.0.weight"], sd[p + "fAffine.0.weight"]
            nz = wa.shape[1] - 320
            wf[:, idx] /= s
            wa[:, nz + idx] /= s
    return sd


@pytest.fixture(scope="module")
def hip():
    from bfsr_amd.ops import HipOps
    return HipOps("cuda:0")


def _models(ops, opt, sd, psd):
    from bfsr_amd.srflow.models import create_model, models as registry
    m = create_model(opt, ops=ops)
    m.load_network(sd)
    prior = registry.make({"name": "unet", "args": {"depth": 3, "dim": 64, "bilinear": True, "ops": ops}, "sd": psd}, load_sd=True).eval()
    return m, prior


@pytest.mark.parametrize("S", [2.0 ** 17, 2.0 ** -17, "channels"])
def test_srflow_lp_pass_on_rescaled_trunk(S):
    """lp_infer and the SRFlowModel call sequence of test.py:139-148 on the rescaled model against the CPU oracle (fp32) on the same weights:
    <= 1e-4 on sr -- through the bf16x3 re-run for S = 2^17 (ops.fallbacks counts it), for S = 2^-17 and for three key channels 2^-13 below the
    rest (with the weights that read them 2^13 above) either way.  (The unscaled model runs without a fallback: test_lp_pass_at_bench_batch_vs_oracle.)"""
    import oracle.srflow_ref as O
    from bfsr_amd.ops import HipOps
    from bfsr_amd.srflow.test import lp_infer
    opt = options.load(options.DEFAULT_CONF)
    sd0 = synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234)
    if S == "channels":         # three key channels 2^-13 (1.2e-4) below the rest, the weights that read them 2^13 above
        sd, S = rescale_key_channels(sd0, opt, (3, 17, 40), 2.0 ** -13), 0.5
    else:
        sd = rescale_trunk(sd0, opt, S)
    psd = synth.state_dict_from_schema(spec.srflow_prior_schema(), 4321)
    lr = synth.smooth_lr_batch(5, 1, 16, 16)
    truth = O.lp_pipeline(lr, sd, psd, opt, 23, return_all=True)
    if S in (2.0 ** 17, 0.5):        # (one oracle run more: only where a new kind of rescaling is introduced)
        base = O.lp_pipeline(lr, sd0, psd, opt, 23, return_all=True)
        assert float((truth["sr"] - base["sr"]).abs().max()) <= 2e-5, "the rescaling is not function-preserving"
    ops = HipOps("cuda:0")
    m, prior = _models(ops, opt, sd, psd)
    out = lp_infer(m, prior, lr, return_all=True)
    for k in ("sr_raw", "sr"):
        got = out[k].cpu()
        assert torch.isfinite(got).all(), k
        err = float((got - truth[k]).abs().max())
        assert err <= 1e-4, "S = %g: %s max-abs %.3e (fallbacks %d)" % (S, k, err, ops.fallbacks)
    print("S = %g: fallbacks %d" % (S, ops.fallbacks))
    if S > 1:
        assert ops.fallbacks == 1, "an overflow of the fp16 pair must re-run the pass under bf16x3"
    if S <= 1:
        return
    # the wrapper API: get_encode_z -> standardise -> prior -> get_sr, each call guarded on its own
    n0 = ops.fallbacks
    import torch.nn.functional as F
    lr_up = F.interpolate(lr, scale_factor=opt["scale"], mode="bilinear", align_corners=False)
    eps = m.get_encode_z(lr, lr_up, epses=[], add_gt_noise=False)
    assert len(eps) == 2
    eps = [(e - e.mean(dim=1, keepdim=True)) / (e.std(dim=1, keepdim=True) + 1e-8) for e in eps]
    sr = m.get_sr(lq=lr, epses=prior(eps))
    assert torch.isfinite(sr).all()
    assert float((sr.cpu() - truth["sr_raw"]).abs().max()) <= 1e-4
    if S > 1:
        assert ops.fallbacks >= n0 + 2, "get_encode_z and get_sr each overflow and each fall back"


def test_guard_is_off_only_when_asked(hip):
    """check_range=False skips the guard (and its synchronisation): the overflow then stays visible in the flag word."""
    from bfsr_amd.ops import HipOps
    from bfsr_amd.srflow.test import lp_infer
    opt = options.load(options.DEFAULT_CONF)
    sd = rescale_trunk(synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234), opt, 2.0 ** 17)
    psd = synth.state_dict_from_schema(spec.srflow_prior_schema(), 4321)
    ops = HipOps("cuda:0")
    m, prior = _models(ops, opt, sd, psd)
    lp_infer(m, prior, synth.smooth_lr_batch(5, 1, 16, 16), check_range=False)
    assert ops.fallbacks == 0
    with pytest.raises(RuntimeError, match="range of the two-term fp16 split"):
        ops.check_range()


@pytest.mark.parametrize("S", [2.0 ** 17, 2.0 ** -17])
def test_linf_pass_on_rescaled_encoder(S):
    """The LINF side: the RRDB encoder's activations x S (conv_first + every encoder bias), its consumers (the coef | freq convs) / S.  lp_infer and
    infer_from_lr agree with the oracle; S = 2^17 overflows inside the encoder and goes through the fallback."""
    import oracle.linf_ref as OL
    from bfsr_amd.ops import HipOps
    from bfsr_amd.linf.models import make
    from bfsr_amd.linf.test import lp_infer, infer_from_lr
    from test_linf_cpu import mspec, weights
    sd, psd = weights("rrdb", 2024)
    base = {k: v.clone() for k, v in sd.items()}
    for k in sd:
        if k in ("encoder.conv_first.weight", "encoder.conv_first.bias", "encoder.trunk_conv.bias") or (k.startswith("encoder.RRDB_trunk.") and k.endswith(".bias")):
            sd[k] = sd[k] * S
    for n in ("coef.weight", "freq.weight"):
        sd[n] = sd[n] / S
    ops = HipOps("cuda:0")
    m = make(mspec("rrdb"), args={"ops": ops}).eval()
    m.load_state_dict(sd)
    prior = make({"name": "unet", "args": {"in_chans": 27, "depth": 3, "dim": 64, "bilinear": True}}, args={"ops": ops}).eval()
    prior.load_state_dict(psd)
    lr = synth.smooth_lr_batch(9, 1, 24, 24)
    H = W = 96
    batch = OL.batch_prep(lr, (H, W))
    truth = OL.lp_pipeline(batch, sd, psd, mspec("rrdb"), (H, W), return_all=True)
    if S > 1:
        ref0 = OL.lp_pipeline(batch, base, psd, mspec("rrdb"), (H, W), return_all=True)
        assert float((truth["pred"] - ref0["pred"]).abs().max()) <= 2e-5, "the rescaling is not function-preserving"
    out = lp_infer(m, prior, batch, (H, W), return_all=True)
    for k in ("z_lr", "pred"):
        got = out[k].cpu()
        assert torch.isfinite(got).all(), k
        err = float((got - truth[k]).abs().max())
        assert err <= 1e-4 * max(1.0, float(truth[k].abs().max())), "S = %g: %s max-abs %.3e (fallbacks %d)" % (S, k, err, ops.fallbacks)
    if S > 1:
        assert ops.fallbacks == 1, "an overflow inside the encoder must re-run the pass under bf16x3"
    again = infer_from_lr(m, prior, lr, 4)
    assert float((again.cpu() - truth["pred"]).abs().max()) <= 1e-4


def test_chain_timeout_degrades_to_launches(hip):
    """A dependency time-out of the fused RRDB launch (status bit 2: its workgroups were not all resident, e.g. a co-tenant holds compute units)
    must not fail the pass: the guard disables the fused launch for this HipOps, repeats the pass on per-conv launches and returns the SAME bits.
    The time-out itself is injected (the first flag read-back reports bit 2); the kernel-side bound is ~2 s of spinning."""
    from bfsr_amd.ops import HipOps
    from bfsr_amd.srflow.test import lp_infer
    opt = options.load(options.DEFAULT_CONF)
    sd = synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234)
    psd = synth.state_dict_from_schema(spec.srflow_prior_schema(), 4321)
    ops = HipOps("cuda:0")
    m, prior = _models(ops, opt, sd, psd)
    lr = synth.smooth_lr_batch(31, 6, 160, 160)               # 300 tiles of 16 x 32: the fused launch is active
    eng = m.netG.module.engine()
    assert eng.rrdb._use_chain(ops.to_device(lr))
    ref = lp_infer(m, prior, lr).clone()
    real, state = ops.read_range_flag, {"n": 0}

    def injected():
        v = real()
        state["n"] += 1
        return v | 4 if state["n"] == 1 else v
    ops.read_range_flag = injected
    try:
        out = lp_infer(m, prior, lr)
    finally:
        ops.read_range_flag = real
    assert getattr(ops, "chain_timeouts", 0) == 1 and ops.chain_disabled
    assert not eng.rrdb._use_chain(ops.to_device(lr))
    assert torch.equal(out, ref), "the per-conv launches must reproduce the fused launch bit for bit"


@pytest.mark.parametrize("shape", [(3, 20, 16, 24), (2, 7, 9, 13)])          # 16-byte vector path (H*W % 4 == 0) | scalar path
def test_channel_range_check_bits(hip, shape):
    """bfsr_channel_range_check on a channel-slice view (round 6 rule, per SAMPLE): bit 0 iff a channel reaches 65504 or is not finite; bit 3 iff
    a channel of some sample is tiny everywhere (0 < max |x| < 2^-7) AND its absolute split error matters: max_c(m_c g_c) < g_c / 32 -- the whole
    sample is tiny, or the weights that read the tiny channel are far above the rest (gain).  An all-zero channel, a few tiny elements among
    normal ones and a tiny channel read by ordinary weights next to normal channels raise nothing."""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(7)
    base = torch.randn(B, C + 3, H, W, generator=g)
    base[:, 1] = 0.0                                       # exact zeros are exact in any format
    base[0, 2, 3, 4] = 1.0e-6                              # one tiny element among normal ones
    x = hip.to_device(base)
    view = x[:, 1:C + 1]                                   # a channel slice of a wider buffer (batch stride != C*H*W)
    hip.read_range_flag()
    hip.check_channels(view)
    assert hip.read_range_flag() == 0
    x[:, 3] *= 1.0e-4                                      # channel 2 of the view: max ~ 4e-4 < 2^-7, next to normal channels read by the same weights
    hip.check_channels(view)
    assert hip.read_range_flag() == 0                      # (round 5 flagged this)
    gain = torch.full((C,), 2.0 ** -13, device=x.device)
    gain[2] = 1.0                                          # ... but read by weights 2^13 above the rest: its error is what the consumer sees
    hip.check_channels(view, gain)
    assert hip.read_range_flag() == 8
    gain[2] = 2.0 ** -13
    gain[5] = 1.0                                          # the heavy weights read a normal channel instead
    hip.check_channels(view, gain)
    assert hip.read_range_flag() == 0
    x[:, 3] *= 1.0e4
    keep = x[B - 1].clone()
    x[B - 1] *= 2.0 ** -12                                 # ONE sample tiny as a whole: its trigger does not depend on its batch-mates
    hip.check_channels(view)
    assert hip.read_range_flag() == 8
    hip.check_channels(view, tiny=0.0)                     # overflow side only
    assert hip.read_range_flag() == 0
    x[B - 1] = keep
    hip.check_channels(view[:B - 1])
    assert hip.read_range_flag() == 0
    x[B - 1, C, H - 1, W - 1] = 7.0e4                      # last channel of the view
    hip.check_channels(view)
    assert hip.read_range_flag() == 1
    x[B - 1, C, H - 1, W - 1] = float("nan")
    hip.check_channels(view)
    assert hip.read_range_flag() == 1
    x[B - 1, C, H - 1, W - 1] = 1.0
    x[:, C + 1] = 1.0e5                                    # outside the view: not looked at
    hip.check_channels(view)
    assert hip.read_range_flag() == 0


def test_channel_gain_vector(hip):
    """ops.channel_gain: per input channel sum |w| over the taps, max over the couts, normalised to the conv's largest input channel (over ALL of its
    channels, also when only a slice is asked for), max over the convs."""
    g = torch.Generator().manual_seed(3)
    w1, w2 = torch.randn(8, 12, 3, 3, generator=g), torch.randn(5, 20, 3, 3, generator=g)
    w1[:, 4] *= 64.0
    def ref(w):
        m = w.abs().double().sum(dim=(2, 3)).max(dim=0)[0]
        return (m / m.max()).float()
    got = hip.channel_gain((w1, 2, 10), (w2, 6, 14)).cpu()
    want = torch.maximum(ref(w1)[2:10], ref(w2)[6:14])
    assert torch.allclose(got, want, rtol=1e-6, atol=0) and float(got[2]) == 1.0
    alone = hip.channel_gain((w1, 2, 10)).cpu()
    assert float(alone.max()) == 1.0 and float(alone[0]) < 0.1           # normalised to the heavy channel 4
