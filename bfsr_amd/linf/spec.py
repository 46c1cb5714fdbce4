"""Parameter schemas of the LINF-LP models (checkpoint contract):
`{'model': {'name','args','sd'}}` / `{'prior_model': {...}}` dicts written by LINF-LP/train.py:234-244 and
read by LINF-LP/test.py:276-281.  Reference anchors (under /root/reference/LINF-LP/):
  models/linf.py:218-242   LINFPatch members (encoder, coef, freq, phase, layers, imnet)
  models/flow.py:11-26,73-94  Flow / NaiveLinear (bias registered before _weight)
  models/rrdb.py:77-103    RRDBNet (upconv1/2, HRconv, conv_last exist even with no_upsampling)
  models/edsr.py:92-146    EDSR (sub_mean / add_mean MeanShift convs are in the state_dict)
  models/unet.py:105-142   prior UNet(in_chans)"""
from collections import OrderedDict

from ..srflow.spec import _bn_schema, _dense_schema, _double_conv_schema  # shared UNet block naming


def rrdb_schema(prefix="encoder.", in_nc=3, out_nc=3, nf=64, nb=23, gc=32):
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
    for n in ("upconv1", "upconv2", "HRconv"):
        conv(n, nf, nf)
    conv("conv_last", out_nc, nf)
    return s


def edsr_schema(prefix="encoder.", n_resblocks=16, n_feats=64, n_colors=3):
    s = OrderedDict()
    for n in ("sub_mean", "add_mean"):
        s[prefix + n + ".weight"] = ((3, 3, 1, 1), "meanshift_w")
        s[prefix + n + ".bias"] = ((3,), "bias_small")
    s[prefix + "head.0.weight"] = ((n_feats, n_colors, 3, 3), "kaiming")
    s[prefix + "head.0.bias"] = ((n_feats,), "bias_small")
    for i in range(n_resblocks):
        for j in (0, 2):
            s[prefix + "body.%d.body.%d.weight" % (i, j)] = ((n_feats, n_feats, 3, 3), "kaiming0.3")
            s[prefix + "body.%d.body.%d.bias" % (i, j)] = ((n_feats,), "bias_small")
    s[prefix + "body.%d.weight" % n_resblocks] = ((n_feats, n_feats, 3, 3), "kaiming0.3")
    s[prefix + "body.%d.bias" % n_resblocks] = ((n_feats,), "bias_small")
    return s


def encoder_schema(encoder_spec, prefix="encoder."):
    name, args = encoder_spec["name"], dict(encoder_spec.get("args") or {})
    if name == "rrdb":
        return rrdb_schema(prefix, args.get("in_nc", 3), args.get("out_nc", 3), args.get("nf", 64), args.get("nb", 23),
                           args.get("gc", 32)), args.get("nf", 64)
    if name == "edsr-baseline":
        return edsr_schema(prefix, args.get("n_resblocks", 16), args.get("n_feats", 64)), args.get("n_feats", 64)
    raise NotImplementedError("encoder '%s' is outside the hot-path scope (SURVEY.md section 2a)" % name)


def flow_schema(prefix="imnet.", flow_layers=10, patch_size=3):
    D = 3 * patch_size * patch_size
    s = OrderedDict()
    for i in range(flow_layers):
        s[prefix + "linears.%d.bias" % i] = ((D,), "linf_bias")
        s[prefix + "linears.%d._weight" % i] = ((D, D), "linf_linear")
    s[prefix + "last.bias"] = ((D,), "linf_bias")
    s[prefix + "last._weight"] = ((D, D), "linf_linear")
    return s


def linf_schema(encoder_spec, flow_layers=10, num_layer=3, hidden_dim=256, patch_size=3):
    s, out_dim = encoder_schema(encoder_spec)
    for n in ("coef", "freq"):
        s[n + ".weight"] = ((hidden_dim, out_dim, 3, 3), "kaiming")
        s[n + ".bias"] = ((hidden_dim,), "bias_small")
    s["phase.weight"] = ((hidden_dim // 2, 2), "default_conv")
    cin = hidden_dim * 4
    for j in range(num_layer):
        s["layers.%d.weight" % (2 * j)] = ((hidden_dim, cin, 1, 1), "kaiming")
        s["layers.%d.bias" % (2 * j)] = ((hidden_dim,), "bias_small")
        cin = hidden_dim
    s["layers.%d.weight" % (2 * num_layer)] = ((flow_layers * patch_size * patch_size * 3 * 2, hidden_dim, 1, 1), "linf_last")
    s["layers.%d.bias" % (2 * num_layer)] = ((flow_layers * patch_size * patch_size * 3 * 2,), "linf_last_bias")
    s.update(flow_schema("imnet.", flow_layers, patch_size))
    return s


def linf_prior_schema(in_chans, depth=3, dim=64, bilinear=True):
    """models/unet.py:105-142: input_proj, lr_proj.{0,2}, down_layers, up_layers, inc, outc."""
    s = OrderedDict()
    half = dim // 2
    _dense_schema(s, "input_proj.", in_chans, half, half)
    s["lr_proj.0.weight"] = ((in_chans, 3, 3, 3), "kaiming")
    s["lr_proj.0.bias"] = ((in_chans,), "bias_small")
    _dense_schema(s, "lr_proj.2.", in_chans, half, half)
    factor = 2 if bilinear else 1
    if not bilinear:
        raise NotImplementedError("only bilinear=True is shipped")
    for i in range(depth):
        cin = dim * (2 ** i)
        cout = dim * (2 ** (i + 1)) // (factor if i == depth - 1 else 1)
        _double_conv_schema(s, "down_layers.%d.maxpool_conv.1." % i, cin, cout)
    for i in range(depth):
        cin = dim * (2 ** (depth - i))
        cout = dim * (2 ** (depth - i - 1)) // (factor if i < depth - 1 else 1)
        _double_conv_schema(s, "up_layers.%d.conv." % i, cin, cout, cin // 2)
    _double_conv_schema(s, "inc.", dim, dim)
    s["outc.conv.weight"] = ((in_chans, dim, 1, 1), "kaiming")
    s["outc.conv.bias"] = ((in_chans,), "bias_small")
    return s
