"""Architecture spec + parameter schema of SRFlowNet / the SRFlow-LP prior, derived from the
YAML `opt` exactly the way the reference constructs them (config-driven: K, L, scale,
additionalFlowNoAffine, split.enable, stackRRDB.blocks, fea_up0).

Reference anchors (under /root/reference/SRFlow-LP/code/):
  models/modules/FlowUpsamplerNet.py:30-115  layer list, levelToName, unused `f` conv
  models/modules/FlowStep.py:31-86           actnorm / invconv / affine members
  models/modules/FlowAffineCouplingsAblation.py:25-55, 127-135   fAffine / fFeatures widths
  models/modules/Split.py:26-37              Split2d.conv
  models/modules/RRDBNet_arch.py:25-87       RRDB parameters
  models/unet.py:109-152                     prior UNet parameters
The state_dict key names are the checkpoint contract.
"""
from collections import OrderedDict

from .options import opt_get

IN_CHANNELS_RRDB = 320      # hard-coded, FlowAffineCouplingsAblation.py:30
HIDDEN = 64                 # FlowAffineCouplingsAblation.py:35-36 default
BASE = 160                  # FlowUpsamplerNet built for a (160,160,3) image, SRFlowNet_arch.py:48


def level_to_name(scale):
    """FlowUpsamplerNet.py:49-74."""
    if scale == 16:
        return {0: "fea_up16", 1: "fea_up8", 2: "fea_up4", 3: "fea_up2", 4: "fea_up1"}
    if scale == 8:
        return {0: "fea_up8", 1: "fea_up4", 2: "fea_up2", 3: "fea_up1", 4: "fea_up0"}
    if scale == 4:
        return {0: "fea_up4", 1: "fea_up2", 2: "fea_up1", 3: "fea_up0", 4: "fea_up-1"}
    raise NotImplementedError("scale %r" % (scale,))


class LayerSpec(object):
    __slots__ = ("index", "type", "C", "level", "coupled", "C_pass", "C_consume")

    def __init__(self, index, type, C, level, coupled=False, C_pass=0, C_consume=0):
        self.index, self.type, self.C, self.level = index, type, C, level
        self.coupled, self.C_pass, self.C_consume = coupled, C_pass, C_consume

    def __repr__(self):
        return "LayerSpec(%d,%s,C=%d,level=%d,coupled=%s)" % (
            self.index, self.type, self.C, self.level, self.coupled)


POINTWISE_CHANNELS = (3, 6, 12, 24, 48, 96)     # channel counts bfsr_flow_pointwise is instantiated for (flow_ops.hip)


def check_supported(opt):
    """Fail at construction time on every reference option that changes the maths and that this engine does not implement
    (the reference would build a different network; silently ignoring the key would produce wrong images)."""
    g = lambda *k, default=None: opt_get(opt, ["network_G", "flow"] + list(k), default)
    if g("coupling") != "CondAffineSeparatedAndCond":
        raise NotImplementedError("only flow.coupling=CondAffineSeparatedAndCond is on the hot path (FlowStep.py:77-83)")
    bad = []
    if g("split", "conditional"):
        bad.append("split.conditional (Split2d conv conditioned on ft, FlowUpsamplerNet.py:156)")
    if g("split", "cond_channels"):
        bad.append("split.cond_channels (FlowUpsamplerNet.py:157)")
    if g("split", "logs_eps"):
        bad.append("split.logs_eps (exp(logs)+logs_eps, Split.py:46)")
    if g("split", "type", default="Split2d") != "Split2d":
        bad.append("split.type != Split2d (FlowUpsamplerNet.py:160)")
    if g("levelConditional", "conditional") is True or g("levelConditional", "n_channels"):
        bad.append("levelConditional (FlowUpsamplerNet.py:83,274)")
    if g("condAff") or g("condFtAffine"):
        bad.append("condAff / condFtAffine (FlowUpsamplerNet.py:144-147)")
    if g("norm"):
        bad.append("norm (FlowUpsamplerNet.py:79)")
    if g("CondAffineSeparatedAndCond", "hidden_channels") not in (None, HIDDEN):
        bad.append("CondAffineSeparatedAndCond.hidden_channels != 64 (FlowAffineCouplingsAblation.py:34-35)")
    if g("CondAffineSeparatedAndCond", "eps", default=1e-4) != 1e-4:
        bad.append("CondAffineSeparatedAndCond.eps != 1e-4 (FlowAffineCouplingsAblation.py:37)")
    if g("fea_up-1"):
        bad.append("fea_up-1 (RRDBNet_arch.py:139)")
    if bad:
        raise NotImplementedError("reference options outside the accelerated path: " + "; ".join(bad))


def flow_layers(opt):
    """The `FlowUpsamplerNet.layers` list as LayerSpec objects; `level` is the level the
    reference derives from the construction-time size (log2(160/size), :230,:280)."""
    check_supported(opt)
    flow = opt["network_G"]["flow"]
    L = flow["L"]
    K = flow["K"]
    Ks = [K] * (L + 1) if isinstance(K, int) else list(K)
    n_add = int(flow.get("additionalFlowNoAffine", 0) or 0)
    split_on = bool(opt_get(opt, ["network_G", "flow", "split", "enable"]))
    correction = 0 if opt_get(opt, ["network_G", "flow", "split", "correct_splits"], False) else 1
    ratio = opt_get(opt, ["network_G", "flow", "split", "consume_ratio"]) or 0.5
    C = 3
    out = []
    for level in range(1, L + 1):
        C *= 4
        out.append(LayerSpec(len(out), "squeeze", C, level))
        for _ in range(n_add):
            out.append(LayerSpec(len(out), "step", C, level, coupled=False))
        for _ in range(Ks[level]):
            out.append(LayerSpec(len(out), "step", C, level, coupled=True))
        if split_on and level < L - correction:
            consume = int(round(C * ratio))
            out.append(LayerSpec(len(out), "split", C, level, C_pass=C - consume, C_consume=consume))
            C -= consume
    for ly in out:
        if ly.type == "step" and ly.C not in POINTWISE_CHANNELS:
            raise NotImplementedError("FlowStep with C=%d channels (L=%d): bfsr_flow_pointwise supports C in %s" %
                                      (ly.C, L, POINTWISE_CHANNELS))
    return out


def final_channels(opt):
    ls = flow_layers(opt)
    last = ls[-1]
    return last.C_pass if last.type == "split" else last.C


def n_rrdb_channels(opt):
    blocks = opt_get(opt, ["network_G", "flow", "stackRRDB", "blocks"])
    return 64 if blocks is None else (len(blocks) + 1) * 64


# --------------------------------------------------------------------------------------------
# parameter schemas: OrderedDict name -> (shape, kind)
# `kind` drives the synthetic-weight recipe (bfsr_amd/synth.py); it is not part of the contract.
# --------------------------------------------------------------------------------------------
def rrdb_schema(opt, nb, nf=64, gc=32, in_nc=3, out_nc=3, prefix="RRDB."):
    s = OrderedDict()

    def conv(name, co, ci):
        s[prefix + name + ".weight"] = ((co, ci, 3, 3), "kaiming0.1")
        s[prefix + name + ".bias"] = ((co,), "bias_small")

    conv("conv_first", nf, in_nc)
    for b in range(nb):
        for r in (1, 2, 3):
            p = "RRDB_trunk.%d.RDB%d." % (b, r)
            for i in range(1, 5):
                conv(p + "conv%d" % i, gc, nf + (i - 1) * gc)
            conv(p + "conv5", nf, nf + 4 * gc)
    conv("trunk_conv", nf, nf)
    conv("upconv1", nf, nf)
    conv("upconv2", nf, nf)
    scale = opt["scale"]
    if scale >= 8:
        conv("upconv3", nf, nf)
    if scale >= 16:
        conv("upconv4", nf, nf)
    if scale >= 32:
        conv("upconv5", nf, nf)
    conv("HRconv", nf, nf)
    conv("conv_last", out_nc, nf)
    return s


def _coupling_net_schema(s, p, cin, cout):
    s[p + "0.weight"] = ((HIDDEN, cin, 3, 3), "flowconv")
    s[p + "0.actnorm.bias"] = ((1, HIDDEN, 1, 1), "an_bias")
    s[p + "0.actnorm.logs"] = ((1, HIDDEN, 1, 1), "an_logs")
    s[p + "2.weight"] = ((HIDDEN, HIDDEN, 1, 1), "flowconv")
    s[p + "2.actnorm.bias"] = ((1, HIDDEN, 1, 1), "an_bias")
    s[p + "2.actnorm.logs"] = ((1, HIDDEN, 1, 1), "an_logs")
    s[p + "4.weight"] = ((cout, HIDDEN, 3, 3), "zeros_w")
    s[p + "4.bias"] = ((cout,), "zeros_b_affine")
    s[p + "4.logs"] = ((cout, 1, 1), "zeros_logs")


def flow_schema(opt, prefix="flowUpsamplerNet."):
    s = OrderedDict()
    for ly in flow_layers(opt):
        p = prefix + "layers.%d." % ly.index
        if ly.type == "step":
            s[p + "actnorm.bias"] = ((1, ly.C, 1, 1), "an_bias")
            s[p + "actnorm.logs"] = ((1, ly.C, 1, 1), "an_logs")
            s[p + "invconv.weight"] = ((ly.C, ly.C), "orthogonal")
            if ly.coupled:
                cn = ly.C // 2
                _coupling_net_schema(s, p + "affine.fAffine.", cn + IN_CHANNELS_RRDB, (ly.C - cn) * 2)
                _coupling_net_schema(s, p + "affine.fFeatures.", IN_CHANNELS_RRDB, ly.C * 2)
        elif ly.type == "split":
            s[p + "conv.weight"] = ((ly.C_consume * 2, ly.C_pass, 3, 3), "zeros_w")
            s[p + "conv.bias"] = ((ly.C_consume * 2,), "zeros_b_split")
            s[p + "conv.logs"] = ((ly.C_consume * 2, 1, 1), "zeros_logs")
    # unused 3x3 conv `f` (FlowUpsamplerNet.py:107-110): constructed, never called, in the state_dict
    aff_in = n_rrdb_channels(opt)
    f_out = 2 * 3 * 64 // 2 // 2 if opt_get(opt, ["network_G", "flow", "split", "enable"]) else 2 * 3 * 64
    s[prefix + "f.0.weight"] = ((f_out, aff_in, 3, 3), "kaiming0.1")
    s[prefix + "f.0.bias"] = ((f_out,), "bias_small")
    return s


def srflownet_schema(opt, nb=None):
    g = opt["network_G"]
    s = rrdb_schema(opt, nb if nb is not None else g["nb"], nf=g["nf"], in_nc=g["in_nc"], out_nc=g["out_nc"])
    s.update(flow_schema(opt))
    return s


def _dense_schema(s, p, nf, gc, out_dim):
    for i in range(1, 5):
        s[p + "conv%d.weight" % i] = ((gc, nf + (i - 1) * gc, 3, 3), "kaiming0.1")
        s[p + "conv%d.bias" % i] = ((gc,), "bias_small")
    s[p + "conv5.weight"] = ((out_dim, nf + 4 * gc, 3, 3), "kaiming0.1")
    s[p + "conv5.bias"] = ((out_dim,), "bias_small")


def _double_conv_schema(s, p, cin, cout, mid=None):
    mid = mid or cout
    s[p + "double_conv.0.weight"] = ((mid, cin, 3, 3), "kaiming")
    _bn_schema(s, p + "double_conv.1.", mid)
    s[p + "double_conv.3.weight"] = ((cout, mid, 3, 3), "kaiming")
    _bn_schema(s, p + "double_conv.4.", cout)


def _bn_schema(s, p, c):
    s[p + "weight"] = ((c,), "bn_weight")
    s[p + "bias"] = ((c,), "bn_bias")
    s[p + "running_mean"] = ((c,), "bn_mean")
    s[p + "running_var"] = ((c,), "bn_var")
    s[p + "num_batches_tracked"] = ((), "bn_count")


def unet_body_schema(s, tag, depth, dim, bilinear, out_ch):
    """Shared UNet body naming (models/unet.py:120-152) with a per-branch suffix `tag`."""
    factor = 2 if bilinear else 1
    if not bilinear:
        raise NotImplementedError("only bilinear=True is shipped (confs/SRFlow-LP_DF2K_4X.yml:58-62)")
    for i in range(depth):
        cin = dim * (2 ** i)
        cout = dim * (2 ** (i + 1)) // (factor if i == depth - 1 else 1)
        _double_conv_schema(s, "down_layers%s.%d.maxpool_conv.1." % (tag, i), cin, cout)
    for i in range(depth):
        cin = dim * (2 ** (depth - i))
        cout = dim * (2 ** (depth - i - 1)) // (factor if i < depth - 1 else 1)
        _double_conv_schema(s, "up_layers%s.%d.conv." % (tag, i), cin, cout, cin // 2)
    _double_conv_schema(s, "inc%s." % tag, dim, dim)
    s["outc%s.conv.weight" % tag] = ((out_ch, dim, 1, 1), "kaiming")
    s["outc%s.conv.bias" % tag] = ((out_ch,), "bias_small")


def srflow_prior_schema(depth=3, dim=64, bilinear=True):
    """models/unet.py:109-152 (member creation order: input_proj0/1, down0, up0, down1, up1, inc0,
    inc1, outc0, outc1)."""
    s = OrderedDict()
    _dense_schema(s, "input_proj0.", 6, dim, dim)
    _dense_schema(s, "input_proj1.", 96, dim, dim)
    parts = {}
    for tag, out_ch in (("0", 6), ("1", 96)):
        t = OrderedDict()
        unet_body_schema(t, tag, depth, dim, bilinear, out_ch)
        parts[tag] = t
    # registration order in the reference: down_layers0, up_layers0, down_layers1, up_layers1, inc0, inc1, outc0, outc1
    for tag in ("0", "1"):
        for k, v in parts[tag].items():
            if k.startswith("down_layers") or k.startswith("up_layers"):
                s[k] = v
    for pre in ("inc", "outc"):
        for tag in ("0", "1"):
            for k, v in parts[tag].items():
                if k.startswith(pre):
                    s[k] = v
    return s
