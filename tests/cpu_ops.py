"""TEST DOUBLE (tests only): a torch-CPU implementation of the `bfsr_amd.ops.HipOps` interface.

Two uses: (1) `-m "not gpu"` tests run the product's host-side schedule (hoisting, buffer slicing, weight
packing order, split/squeeze plumbing) on CPU against the oracle and the golden vectors; (2) `-m gpu` tests
compare every HIP op against the same semantics on seeded inputs.  Never imported by the product.
"""
import torch
import torch.nn.functional as F

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2


class PackedConv(object):
    def __init__(self, w, mtile):
        self.w = w
        self.Cout, self.Cin, self.KS = w.shape[0], w.shape[1], w.shape[2]
        self.mtile = mtile


def _cv(v):
    return None if v is None else v.view(1, -1, 1, 1)


class CpuOps(object):
    device = torch.device("cpu")

    def empty(self, *shape):
        return torch.full(shape, float("nan"), dtype=torch.float32)     # poison: catches reads of unwritten scratch

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32)

    def to_device(self, t):
        return t.detach().to(torch.float32).contiguous()

    def vec(self, t):
        return t.detach().reshape(-1).to(torch.float32).contiguous().clone()

    def pack_conv(self, w, mtile=None, out_perm=None):
        w = w.detach().to(torch.float32).contiguous().clone()
        if out_perm is not None:
            w = w[torch.as_tensor(out_perm)].contiguous()
        return PackedConv(w, mtile)

    def conv(self, x, pw, out, in_shift=0, bias=None, pre_add=None, aff_shift=None, aff_scale=None, aff_post=None,
             act=ACT_NONE, slope=0.2, post_scale=None, res1=None, alpha1=1.0, res2=None, alpha2=1.0):
        xin = x
        if in_shift:
            xin = F.interpolate(x, scale_factor=1 << in_shift, mode="nearest")
        assert not torch.isnan(xin).any(), "conv input contains unwritten (NaN) scratch"
        v = F.conv2d(xin, pw.w, None, 1, (pw.KS - 1) // 2)
        if bias is not None:
            v = v + _cv(bias)
        if pre_add is not None:
            v = v + pre_add
        if aff_shift is not None:
            v = v + _cv(aff_shift)
        if aff_scale is not None:
            v = v * _cv(aff_scale)
        if aff_post is not None:
            v = v + _cv(aff_post)
        if act == ACT_RELU:
            v = F.relu(v)
        elif act == ACT_LRELU:
            v = F.leaky_relu(v, slope)
        if post_scale is not None:
            v = v * _cv(post_scale)
        if res1 is not None:
            v = alpha1 * v + res1
        if res2 is not None:
            v = alpha2 * v + res2
        out.copy_(v)
        return out

    def flow_pointwise(self, z_in, z_out, reverse, h_aff=None, h_ft=None, w=None, an_bias=None, an_escale=None, eps=1e-4):
        x = z_in.clone()
        C = x.shape[1]
        cn = C // 2
        assert not torch.isnan(x).any()

        def ss(h):
            return h[:, 0::2], torch.sigmoid(h[:, 1::2] + 2.0) + eps

        W = None if w is None else w.view(C, C, 1, 1)
        if reverse:
            if h_aff is not None:
                sh, sc = ss(h_aff)
                x = torch.cat([x[:, :cn], x[:, cn:] / sc - sh], 1)
            if h_ft is not None:
                sh, sc = ss(h_ft)
                x = x / sc - sh
            if W is not None:
                x = F.conv2d(x, W)
            if an_bias is not None:
                x = x * _cv(an_escale) - _cv(an_bias)
        else:
            if h_aff is not None:
                sh, sc = ss(h_aff)
                x = torch.cat([x[:, :cn], (x[:, cn:] + sh) * sc], 1)
            if an_bias is not None:
                x = (x + _cv(an_bias)) * _cv(an_escale)
            if W is not None:
                x = F.conv2d(x, W)
            if h_ft is not None:
                sh, sc = ss(h_ft)
                x = (x + sh) * sc
        z_out.copy_(x)
        return z_out

    def squeeze2d(self, x, y):
        y.copy_(F.pixel_unshuffle(x, 2))
        return y

    def unsqueeze2d(self, x, y):
        y.copy_(F.pixel_shuffle(x, 2))
        return y

    def split2d(self, h, src, dst, reverse):
        mean, logs = h[:, 0::2], h[:, 1::2]
        dst.copy_(mean + torch.exp(logs) * src if reverse else (src - mean) / torch.exp(logs))
        return dst

    def standardize(self, x, y):
        m = torch.mean(x, dim=[1], keepdim=True)
        s = torch.std(x, dim=[1], keepdim=True)
        y.copy_((x - m) / (s + 1e-8))
        return y

    def resize(self, x, y, mode, r_h, r_w, window=None):
        OH, OW = y.shape[2], y.shape[3]
        oy0, ox0, RH, RW = window if window is not None else (0, 0, OH, OW)
        IH, IW = x.shape[2], x.shape[3]
        ry = torch.arange(RH, dtype=torch.float32)
        rx = torch.arange(RW, dtype=torch.float32)
        r_h = torch.tensor(r_h, dtype=torch.float32)
        r_w = torch.tensor(r_w, dtype=torch.float32)
        if mode == 0:
            sy = torch.clamp(torch.floor(ry * r_h).long(), max=IH - 1)
            sx = torch.clamp(torch.floor(rx * r_w).long(), max=IW - 1)
            r = x[:, :, sy][:, :, :, sx]
        else:
            if mode == 1:
                fy = torch.clamp((ry + 0.5) * r_h - 0.5, min=0)
                fx = torch.clamp((rx + 0.5) * r_w - 0.5, min=0)
            else:
                fy, fx = ry * r_h, rx * r_w
            y0 = torch.clamp(fy.long(), max=IH - 1)
            x0 = torch.clamp(fx.long(), max=IW - 1)
            y1 = torch.clamp(y0 + 1, max=IH - 1)
            x1 = torch.clamp(x0 + 1, max=IW - 1)
            hl1 = (fy - y0.float()).view(1, 1, -1, 1)
            hl0 = 1.0 - hl1
            wl1 = (fx - x0.float()).view(1, 1, 1, -1)
            wl0 = 1.0 - wl1
            g = lambda yy, xx: x[:, :, yy][:, :, :, xx]
            r = hl0 * (wl0 * g(y0, x0) + wl1 * g(y0, x1)) + hl1 * (wl0 * g(y1, x0) + wl1 * g(y1, x1))
        y.zero_()
        y[:, :, oy0:oy0 + RH, ox0:ox0 + RW] = r
        return y

    def maxpool2(self, x, y):
        y.copy_(F.max_pool2d(x, 2))
        return y

    def axpb_clamp(self, x, y, a=1.0, b=0.0, lo=-3.4e38, hi=3.4e38, r=None):
        v = a * x + b
        if r is not None:
            v = v + r
        y.copy_(torch.clamp(v, lo, hi))
        return y
