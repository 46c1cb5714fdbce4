"""TEST DOUBLE (tests only): a torch-CPU implementation of the `bfsr_amd.ops.HipOps` interface.

Two uses: (1) `-m "not gpu"` tests run the product's host-side schedule (hoisting, buffer slicing, weight
packing order, split/squeeze plumbing) on CPU against the oracle and the golden vectors; (2) `-m gpu` tests
compare every HIP op against the same semantics on seeded inputs.  Never imported by the product.
"""
import numpy as np
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
        if t.dtype == torch.float32 and t.is_contiguous() and not t.requires_grad:
            return t
        return t.detach().to(torch.float32).contiguous()

    def vec(self, t):
        return t.detach().reshape(-1).to(torch.float32).contiguous().clone()

    def pack_conv(self, w, mtile=None, out_perm=None):
        w = w.detach().to(torch.float32).contiguous().clone()
        if out_perm is not None:
            w = w[torch.as_tensor(out_perm)].contiguous()
        return PackedConv(w, mtile)

    def pack_conv_f16(self, w, mtile=None):
        return PackedConv(w.detach().to(torch.float32).half().float().contiguous().clone(), mtile)

    def conv_f16(self, x, pw, out, **kw):
        """Semantics: inputs and weights rounded to fp16 (RNE), fp32 accumulation and epilogue."""
        return self.conv(x.half().float(), pw, out, **kw)

    conv_mode = "f32"

    def pack_conv_x3(self, w, mtile=None, lazy=False):
        return self.pack_conv(w, mtile)

    def conv_x3(self, x, pw, out, **kw):
        return self.conv(x, pw, out, **kw)

    # ---- x3 tensors: [B, C/8, 3, H, W, 8] bf16, x = h + m + l exactly (HipOps.x3_empty)
    def x3_empty(self, B, Cc, H, W):
        assert Cc % 8 == 0
        return torch.full((B, Cc // 8, 3, H, W, 8), float("nan"), dtype=torch.bfloat16)

    @staticmethod
    def _x3_to_f32(t):
        B, C8, _, H, W, _ = t.shape
        v = (t[:, :, 0].float() + t[:, :, 1].float()) + t[:, :, 2].float()          # [B, C8, H, W, 8]
        return v.permute(0, 1, 4, 2, 3).reshape(B, C8 * 8, H, W)

    def x3_pack(self, x, out):
        B, Cc, H, W = x.shape
        v = x.reshape(B, Cc // 8, 8, H, W).permute(0, 1, 3, 4, 2)
        h = v.bfloat16()
        r1 = v - h.float()
        m = r1.bfloat16()
        l = (r1 - m.float()).bfloat16()
        out[:, :, 0], out[:, :, 1], out[:, :, 2] = h, m, l
        assert torch.equal(self._x3_to_f32(out), x), "x3 split is not lossless"
        return out

    def x3_unpack(self, x, out):
        out.copy_(self._x3_to_f32(x))
        return out

    def conv_x3s(self, x, pw, out, epi=None, act=0, slope=0.2, res1=None, alpha1=1.0, res2=None, alpha2=1.0, tune=0, y_fmt=0, up4=None):
        f = self._x3_to_f32
        tmp = torch.empty(out.shape[0], pw.Cout, x.shape[3], x.shape[4]) if out.dtype == torch.bfloat16 else out
        self.conv(f(x), pw, tmp, epi=epi, act=act, slope=slope, res1=None if res1 is None else f(res1), alpha1=alpha1,
                  res2=None if res2 is None else f(res2), alpha2=alpha2, y_fmt=y_fmt)
        if up4 is not None:                      # the compact x4 taps result (conv_up4_h2t(compact=True)), expanded and added to the quad-major result
            assert y_fmt and out.dtype != torch.bfloat16
            B, Cout, H4, W4 = tmp.shape
            c5 = up4.view(B, Cout, 9, H4 // 4, W4 // 4)
            cls = (0, 1, 1, 2)
            full = torch.empty(B, Cout, H4, W4)
            for py in range(4):
                for px in range(4):
                    full[:, :, py::4, px::4] = c5[:, :, cls[py] * 3 + cls[px]]
            tmp.copy_(self.quads(self.quads(tmp, inverse=True) + full))
        if out.dtype == torch.bfloat16:
            self.x3_pack(tmp, out)
        return out

    # ---- h2 tensors: [B, C/8, 2, H, W, 8] fp16, hi = fp16(x), lo = fp16(x - hi) (HipOps.h2_empty); convs read hi only
    def h2_empty(self, B, Cc, H, W):
        assert Cc % 8 == 0
        return torch.full((B, Cc // 8, 2, H, W, 8), float("nan"), dtype=torch.float16)

    @staticmethod
    def _h2_planes(t):
        B, C8, _, H, W, _ = t.shape
        f = lambda p: p.float().permute(0, 1, 4, 2, 3).reshape(B, C8 * 8, H, W)
        return f(t[:, :, 0]), f(t[:, :, 1])

    def h2_pack(self, x, out, hi_only=False):
        B, Cc, H, W = x.shape
        v = x.reshape(B, Cc // 8, 8, H, W).permute(0, 1, 3, 4, 2)
        h = v.half()
        out[:, :, 0] = h
        if not hi_only:                          # hi-only outputs keep their NaN lo plane: a residual read of one would show
            out[:, :, 1] = (v - h.float()).half()
        return out

    def h2_unpack(self, x, out):
        hi, lo = self._h2_planes(x)
        out.copy_(hi + lo)
        return out

    def pack_conv_h2s(self, w):
        return PackedConv(w.detach().to(torch.float32).half().float().contiguous().clone(), 1)

    def conv_h2s(self, x, pw, out, epi=None, act=0, slope=0.2, res1=None, alpha1=1.0, res2=None, alpha2=1.0, hi_only=False, tune=0):
        full = lambda t: None if t is None else sum(self._h2_planes(t))
        tmp = torch.empty(out.shape[0], pw.Cout, x.shape[3], x.shape[4]) if out.dtype == torch.float16 else out
        self.conv(self._h2_planes(x)[0], pw, tmp, epi=epi, act=act, slope=slope, res1=full(res1), alpha1=alpha1, res2=full(res2), alpha2=alpha2)
        if out.dtype == torch.float16:
            self.h2_pack(tmp, out, hi_only=hi_only)
        return out

    def pack_conv1x1(self, w, x3=True):
        w = w.detach().to(torch.float32).reshape(w.shape[0], w.shape[1], 1, 1)
        return PackedConv((w if x3 else w.half().float()).contiguous().clone(), 8)

    def conv1x1(self, x, pw, out, x3=True, **kw):
        return self.conv(x if x3 else x.half().float(), pw, out, **kw)

    def pack_conv_up4_x3(self, w):
        return self.pack_conv_up2(w, 1)

    def conv_up4_x3(self, x, pw, out, epi=None, pre_add=None, act=0, slope=0.2):
        xin = F.interpolate(x, scale_factor=4, mode="nearest")
        return self.conv(xin, pw, out, epi=epi, pre_add=pre_add.clone() if pre_add is not None else None, act=act, slope=slope)

    def pack_conv_up2_x3(self, w):
        return self.pack_conv_up2(w, 1)

    def conv_up2_x3(self, x, pw, out, epi=None, pre_add=None, act=0, slope=0.2, tune=0, y_fmt=0):
        return self.conv_up2(x, pw, out, epi=epi, pre_add=pre_add.clone() if pre_add is not None else None, act=act, slope=slope, y_fmt=y_fmt)

    def pack_conv_up2(self, w, mtile=2):
        return PackedConv(w.detach().to(torch.float32).contiguous().clone(), mtile)

    def conv_up2(self, x, pw, out, epi=None, pre_add=None, act=ACT_NONE, slope=0.2, key=None, y_fmt=0):
        """Semantics: the plain 3x3 conv over cat[key channels, materialised nearest-x2 upsample] (original weights)."""
        xin, w = F.interpolate(x, scale_factor=2, mode="nearest"), pw
        if key is not None:
            x2, pk = key
            xin = torch.cat([x2, xin], 1)
            w = PackedConv(torch.cat([pk.w, pw.w], 1), pw.mtile)
        return self.conv(xin, w, out, epi=epi, pre_add=pre_add, act=act, slope=slope, y_fmt=y_fmt)

    def pack_epilogue(self, Cout, bias=None, aff_shift=None, aff_scale=None, aff_post=None, post_scale=None):
        vs = dict(bias=bias, aff_shift=aff_shift, aff_scale=aff_scale, aff_post=aff_post, post_scale=post_scale)
        if all(v is None for v in vs.values()):
            return None
        return {k: (None if v is None else v.detach().reshape(-1).to(torch.float32).clone()) for k, v in vs.items()}

    def conv(self, x, pw, out, in_shift=0, epi=None, pre_add=None, act=ACT_NONE, slope=0.2, res1=None, alpha1=1.0,
             res2=None, alpha2=1.0, tune=0, stage2=None, bias=None, aff_shift=None, aff_scale=None, aff_post=None,
             post_scale=None, y_fmt=0):
        """y_fmt=1: `out` and `pre_add` are quad-major ([B][Cout/4][H][W][4] in the same buffers; HipOps.conv_x3 / conv_up2_x3 / conv_h2x)."""
        if y_fmt and pre_add is not None:
            pre_add = self.quads(pre_add, inverse=True)
        if epi is not None:
            bias, aff_shift, aff_scale = epi["bias"], epi["aff_shift"], epi["aff_scale"]
            aff_post, post_scale = epi["aff_post"], epi["post_scale"]
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
        if stage2 is not None:
            pw2, epi2, act2 = stage2
            v = F.conv2d(v, pw2.w)
            if epi2 is not None:
                for k, fn in (("bias", torch.add), ("aff_shift", torch.add), ("aff_scale", torch.mul), ("aff_post", torch.add)):
                    if epi2[k] is not None:
                        v = fn(v, _cv(epi2[k]))
            v = F.relu(v) if act2 == ACT_RELU else (F.leaky_relu(v, slope) if act2 == ACT_LRELU else v)
            if epi2 is not None and epi2["post_scale"] is not None:
                v = v * _cv(epi2["post_scale"])
        out.copy_(self.quads(v) if y_fmt else v)
        return out

    def flow_pointwise(self, z_in, z_out, reverse, h_aff=None, h_ft=None, w=None, an_bias=None, an_escale=None, eps=1e-4,
                       wt=None):
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

    # ---- LINF-LP ------------------------------------------------------------------------------------
    def linf_features(self, cf, coord, cell, phase, out, hidden):
        import numpy as np
        coef, freq = cf[:, :hidden], cf[:, hidden:]
        h, w = cf.shape[-2:]
        rx, ry, e = 2 / h / 2, 2 / w / 2, 1e-6
        seq = lambda n: (-1 + 1.0 / n) + (2 * (1.0 / n)) * torch.arange(n).float()
        fc = torch.stack(torch.meshgrid(seq(h), seq(w), indexing="ij"), 0).unsqueeze(0).expand(cf.shape[0], 2, h, w)
        freqs, coefs, areas = [], [], []
        for vx in (-1, 1):
            for vy in (-1, 1):
                c_ = coord.clone()
                c_[..., 0] += vx * rx + e
                c_[..., 1] += vy * ry + e
                c_.clamp_(-1 + 1e-6, 1 - 1e-6)
                q = F.grid_sample(fc, c_.flip(-1), mode="nearest", align_corners=False)
                rel = coord.permute(0, 3, 1, 2) - q
                rel[:, 0] *= h
                rel[:, 1] *= w
                rc = cell.clone()
                rc[:, 0] *= h
                rc[:, 1] *= w
                co = F.grid_sample(coef, c_.flip(-1), mode="nearest", align_corners=False)
                fr = F.grid_sample(freq, c_.flip(-1), mode="nearest", align_corners=False)
                fr = torch.stack(torch.split(fr, hidden // 2, dim=1), dim=2)
                fr = torch.sum(fr * rel.unsqueeze(1), dim=2) + F.linear(rc, phase.view(hidden // 2, 2))[..., None, None]
                freqs.append(torch.cat((torch.cos(np.pi * fr), torch.sin(np.pi * fr)), 1))
                coefs.append(co)
                areas.append(torch.abs(rel[:, 0] * rel[:, 1]) + 1e-9)
        tot = torch.stack(areas).sum(0)
        areas = [areas[3], areas[2], areas[1], areas[0]]
        out.copy_(torch.cat([((areas[i] / tot).unsqueeze(1) * coefs[i]) * freqs[i] for i in range(4)], 1))
        return out

    def pack_coupling_head(self, w0_z1, w2, shift0, scale0, shift2, scale2):
        f = lambda t: t.detach().to(torch.float32).clone()
        return (None if w0_z1 is None else f(w0_z1), f(w2).reshape(64, 64, 1, 1), f(shift0).reshape(-1), f(scale0).reshape(-1), f(shift2).reshape(-1),
                f(scale2).reshape(-1))

    @staticmethod
    def quads(t, inverse=False):
        """[B,C,H,W] values <-> the same buffer quad-major [B][C/4][H][W][4] (pre_fmt / h_ft_fmt = 1 of the coupling kernels)."""
        B, Cc, H, W = t.shape
        if inverse:
            return t.reshape(B, Cc // 4, H, W, 4).permute(0, 1, 4, 2, 3).reshape(B, Cc, H, W)
        return t.reshape(B, Cc // 4, 4, H, W).permute(0, 1, 3, 4, 2).reshape(B, Cc, H, W)

    def coupling_head(self, z, packed, pre_aff, hid, pre_fmt=0):
        """hid: an h2 tensor (h2_empty(B, 64, H, W)) or, for the per-op tests, an fp32 [B,64,H,W] tensor."""
        w0, w2, s0, c0, s2, c2 = packed
        if pre_fmt:
            pre_aff = self.quads(pre_aff, inverse=True)
        if w0 is None:                           # the 1x1-only form (hoisted fFeatures nets)
            t = F.relu((pre_aff + _cv(s0)) * _cv(c0))
        else:
            t = F.relu((F.conv2d(z[:, :w0.shape[1]], w0, None, 1, 1) + pre_aff + _cv(s0)) * _cv(c0))
        v = F.relu((F.conv2d(t, w2) + _cv(s2)) * _cv(c2))
        if hid.dtype == torch.float16:
            self.h2_pack(v, hid)
        else:
            hid.copy_(v)
        return hid

    def pack_coupling_tail(self, w4, bias, post_scale):
        f = lambda t: t.detach().to(torch.float32).clone()
        return f(w4), f(bias).reshape(-1), f(post_scale).reshape(-1), w4.shape[0]

    def coupling_tail(self, hid, packed, z_in, z_out, reverse, h_ft=None, w=None, an_bias=None, an_escale=None, eps=1e-4, h_ft_fmt=0):
        w4, b4, ps, _ = packed
        if hid.dtype == torch.float16:
            hid = sum(self._h2_planes(hid))
        if h_ft is not None and h_ft_fmt:
            h_ft = self.quads(h_ft, inverse=True)
        h_aff = (F.conv2d(hid, w4, None, 1, 1) + _cv(b4)) * _cv(ps)
        return self.flow_pointwise(z_in, z_out, reverse, h_aff=h_aff, h_ft=h_ft, w=w, an_bias=an_bias, an_escale=an_escale, eps=eps)

    def conv_h2r(self, x, packed, out, epi=None, act=0, slope=0.2, y_fmt=0):
        w4 = packed[0]
        return self.conv(sum(self._h2_planes(x)), PackedConv(w4, 1), out, epi=epi, act=act, slope=slope, y_fmt=y_fmt)

    def h2_pack_s2d(self, x, out):
        return self.h2_pack(F.pixel_unshuffle(x, 2).reshape(x.shape[0], x.shape[1], 4, x.shape[2] // 2, x.shape[3] // 2).transpose(1, 2)
                            .reshape(x.shape[0], 4 * x.shape[1], x.shape[2] // 2, x.shape[3] // 2), out)

    def pack_conv_up2_h2t(self, w_taps, w_key=None):
        w = w_taps.detach().to(torch.float32).clone()
        wk = None if w_key is None else w_key.detach().to(torch.float32).clone()
        return (w, wk), 1.0, w.shape[0], w.shape[1], 0 if wk is None else wk.shape[1]

    def conv_up2_h2t(self, x, packed, out, pre_add=None):
        """conv3x3(cat([key, nearest_up2(taps)])): x = h2 tensor of the taps followed by the space-to-depth planes of the key channels;
        out / pre_add hold the quad-major layout."""
        (w, wk), _, Cout, Ct, Ck = packed
        xf = sum(self._h2_planes(x))
        y = F.conv2d(F.interpolate(xf[:, :Ct], scale_factor=2, mode="nearest"), w, None, 1, 1)
        if Ck:
            B, _, h, ww = xf.shape
            key = F.pixel_shuffle(xf[:, Ct:].reshape(B, 4, Ck, h, ww).transpose(1, 2).reshape(B, 4 * Ck, h, ww), 2)
            y = y + F.conv2d(key, wk, None, 1, 1)
        if pre_add is not None:
            y = y + self.quads(pre_add, inverse=True)
        out.copy_(self.quads(y))
        return out

    def pack_conv_up4_h2t(self, w_taps):
        w = w_taps.detach().to(torch.float32).clone()
        return w, 1.0, w.shape[0], w.shape[1]

    def conv_up4_h2t(self, x, packed, out, pre_add=None, compact=False):
        """conv3x3(nearest_up4(taps)) + pre_add: x = h2 tensor of the taps; out / pre_add hold the quad-major layout.
        compact=True: out [B, 9*Cout, h, w] receives the nine phase-class values per source pixel (the double keeps them as [B][Cout][9][h][w]: the layout is
        private to the producer / consumer pair) -- what conv_x3s(up4=) expands and adds."""
        w, _, Cout, Ct = packed
        y = F.conv2d(F.interpolate(sum(self._h2_planes(x)), scale_factor=4, mode="nearest"), w, None, 1, 1)
        if compact:
            B, _, H4, W4 = y.shape
            c5 = out.view(B, Cout, 9, H4 // 4, W4 // 4)
            rep = (0, 1, 3)                      # an output phase of each class {0}, {1, 2}, {3}
            for cy in range(3):
                for cx in range(3):
                    c5[:, :, cy * 3 + cx] = y[:, :, rep[cy]::4, rep[cx]::4]
            return out
        if pre_add is not None:
            y = y + self.quads(pre_add, inverse=True)
        out.copy_(self.quads(y))
        return out

    def check_range(self):
        pass

    @staticmethod
    def _ai_quads(ai, layers, D, inverse=False):
        """[B, 2*D*layers, qh, qw] values <-> the quad-major padded layout of linf_mlp(out_fmt=1) / linf_flow(ai_fmt=1), held in a
        [B, blk*layers, qh, qw] buffer (blk = 2*S, S = D rounded up to a multiple of four): [B][layers][blk/4][qh*qw][4]."""
        B, _, qh, qw = ai.shape
        S = (D + 3) // 4 * 4                     # per layer: S raw scales (D used), then S shifts (D used)
        blk = 2 * S
        if inverse:
            v = ai.reshape(B, layers, blk // 4, qh * qw, 4).permute(0, 1, 2, 4, 3).reshape(B, layers, blk, qh, qw)
            return torch.cat([v[:, :, :D], v[:, :, S:S + D]], 2).reshape(B, 2 * D * layers, qh, qw)
        v = torch.zeros(B, layers, blk, qh, qw)
        a5 = ai.reshape(B, layers, 2 * D, qh, qw)
        v[:, :, :D], v[:, :, S:S + D] = a5[:, :, :D], a5[:, :, D:]
        return v.reshape(B, layers, blk // 4, 4, qh * qw).permute(0, 1, 2, 4, 3).reshape(B, layers * blk, qh, qw)

    def pack_linf_mlp(self, ws, bs, x3=True, quad_layers=None):
        rnd = (lambda t: t) if x3 else (lambda t: t.half().float())
        cout = ws[3].shape[0] if quad_layers is None else (2 * ((quad_layers[1] + 3) // 4 * 4)) * quad_layers[0]
        return ([rnd(t.detach().to(torch.float32).reshape(t.shape[0], t.shape[1], 1, 1)).clone() for t in ws],
                [b.detach().to(torch.float32).reshape(-1).clone() for b in bs], cout, quad_layers)

    def linf_mlp(self, cf, coord, cell, phase, packed, out, hidden, x3=True):
        """Semantics of the fused kernel: Fourier features -> 1x1 MLP (ReLU between layers); fp16 mode rounds every layer's
        operands to fp16, accumulation fp32."""
        ws, bs, _, quad = packed
        rnd = (lambda t: t) if x3 else (lambda t: t.half().float())
        x = self.linf_features(cf, coord, cell, phase, torch.empty(cf.shape[0], 4 * hidden, coord.shape[1], coord.shape[2]), hidden)
        for j, (w, b) in enumerate(zip(ws, bs)):
            x = F.conv2d(rnd(x), w, b)
            if j < len(ws) - 1:
                x = F.relu(x)
        out.copy_(x if quad is None else self._ai_quads(x, quad[0], quad[1]))
        return out

    def zeros_f64(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def resample_taps(self, x, y, idx, w, dim):
        g = x[:, :, idx.long(), :] if dim == 0 else x[:, :, :, idx.long()]          # [B,C,O,P,W] / [B,C,H,O,P]
        y.copy_((g * w.view(1, 1, *w.shape, 1)).sum(3) if dim == 0 else (g * w.view(1, 1, 1, *w.shape)).sum(4))
        return y

    def sqdiff_sum(self, a, b, shave=0, luma=False, rgb_range=1.0):
        d = (a - b) / rgb_range
        if luma:
            d = (d * torch.tensor([65.738, 129.057, 25.064]).view(1, 3, 1, 1) / 256).sum(1, keepdim=True)
        if shave:
            d = d[..., shave:-shave, shave:-shave]
        return d.double().pow(2).sum(dim=(1, 2, 3))

    def ssim_sum(self, a, b, window121, scale=255.0):
        k = window121.view(1, 1, 11, 11)
        B, Cc, H, W = a.shape
        p, q = (a.double() * scale).reshape(B * Cc, 1, H, W), (b.double() * scale).reshape(B * Cc, 1, H, W)
        f = lambda t: F.conv2d(t, k)
        m1, m2 = f(p), f(q)
        v1, v2, cv = f(p * p) - m1 * m1, f(q * q) - m2 * m2, f(p * q) - m1 * m2
        C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
        smap = ((2 * m1 * m2 + C1) * (2 * cv + C2)) / ((m1 * m1 + m2 * m2 + C1) * (v1 + v2 + C2))
        return smap.sum(dim=(1, 2, 3)).view(B, Cc)

    def ssim_sum_w(self, a, b, window, cov_norm=1.0, scale=1.0):
        ws = int(round(window.numel() ** 0.5))
        k = window.view(1, 1, ws, ws)
        B, Cc, H, W = a.shape
        p, q = (a.double() * scale).reshape(B * Cc, 1, H, W), (b.double() * scale).reshape(B * Cc, 1, H, W)
        f = lambda t: F.conv2d(t, k)
        m1, m2 = f(p), f(q)
        v1, v2, cv = cov_norm * (f(p * p) - m1 * m1), cov_norm * (f(q * q) - m2 * m2), cov_norm * (f(p * q) - m1 * m2)
        C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
        smap = ((2 * m1 * m2 + C1) * (2 * cv + C2)) / ((m1 * m1 + m2 * m2 + C1) * (v1 + v2 + C2))
        return smap.sum(dim=(1, 2, 3)).view(B, Cc)

    def to_uint8(self, x):
        return torch.round(torch.clamp(x, 0, 1) * 255.0).to(torch.uint8)

    def logscale_sum(self, h, acc, coef=1.0, eps=1e-4):
        acc += coef * torch.log(torch.sigmoid(h[:, 1::2] + 2.0) + eps).double().sum(dim=(1, 2, 3))
        return acc

    def gaussian_logp(self, x, acc, h=None, coef=1.0):
        l2pi = float(np.log(2 * np.pi))
        if h is None:
            acc += coef * (-0.5 * (x ** 2 + l2pi)).double().sum(dim=(1, 2, 3))
        else:
            mean, logs = h[:, 0::2], h[:, 1::2]
            acc += coef * (-0.5 * (logs * 2.0 + (x - mean) ** 2 / torch.exp(logs * 2.0) + l2pi)).double().sum(dim=(1, 2, 3))
        return acc

    def linf_flow(self, x, ai, y, lin_w, lin_b, layers, reverse, eps=1e-4, log_p=None, logdet_const=0.0, ai_fmt=0):
        B, D, qh, qw = x.shape
        if ai_fmt:
            ai = self._ai_quads(ai, layers, D, inverse=True)
        v = x.permute(0, 2, 3, 1).reshape(-1, D)
        a = ai.permute(0, 2, 3, 1).reshape(-1, 2 * D * layers)
        Wm, bb = lin_w.view(layers + 1, D, D), lin_b.view(layers + 1, D)
        sc = lambda i: torch.sigmoid(a[:, 2 * D * i: 2 * D * i + D] + 2.0) + eps
        sh = lambda i: a[:, 2 * D * i + D: 2 * D * (i + 1)]
        if not reverse:
            ld = torch.full((v.shape[0],), float(logdet_const))
            for i in range(layers):
                v = F.linear(v, Wm[i], bb[i])
                v = v * sc(i) + sh(i)
                ld = ld + torch.log(sc(i)).sum(-1)
            v = F.linear(v, Wm[layers], bb[layers])
            if log_p is not None:
                log_p.copy_((ld + (-0.5 * (v ** 2 + float(np.log(2 * np.pi)))).sum(-1)).reshape(log_p.shape))
        elif int(reverse) == 2:        # VJP of the inverse w.r.t. its input: Wm[i] = inv(W_i)^T, no biases / shifts
            for i in range(layers):
                v = F.linear(v, Wm[i]) / sc(i)
            v = F.linear(v, Wm[layers])
        else:
            v = F.linear(v - bb[layers], Wm[layers])
            for i in reversed(range(layers)):
                v = (v - sh(i)) / sc(i)
                v = F.linear(v - bb[i], Wm[i])
        y.copy_(v.reshape(B, qh, qw, D).permute(0, 3, 1, 2))
        return y

    def patch_fold(self, p, img, ps):
        B, cp, qh, qw = p.shape
        full = F.fold(p.reshape(B, cp, -1), output_size=(qh * ps, qw * ps), kernel_size=(ps, ps), stride=ps)
        img.copy_(full[..., :img.shape[2], :img.shape[3]])
        return img

    def patch_unfold(self, img, p, ps):
        B, Cc, H, W = img.shape
        qh, qw = p.shape[2], p.shape[3]
        x = F.pad(img, (0, qw * ps - W, 0, qh * ps - H))
        u = x.unfold(2, ps, ps).unfold(3, ps, ps)               # B C qh qw ps ps
        p.copy_(u.contiguous().view(B, Cc, qh, qw, ps * ps).permute(0, 1, 4, 2, 3).reshape(B, Cc * ps * ps, qh, qw))
        return p

    def conv_direct(self, x, w, bias, y, stride, pad, act=ACT_NONE, slope=0.2):
        v = F.conv2d(x, w, bias, stride, pad)
        if act == ACT_RELU:
            v = F.relu(v)
        elif act == ACT_LRELU:
            v = F.leaky_relu(v, slope)
        y.copy_(v)
        return y

    def grid_sample_add(self, x, coord, acc, out):
        out.copy_(acc + F.grid_sample(x, coord.flip(-1), mode="bilinear", padding_mode="border", align_corners=False))
        return out


class CpuOpsX3(CpuOps):
    """the test double in the product's default contraction mode: the engines then take the x3 paths (x3-tensor RRDB blocks on
    conv_x3s, parity-decomposed hoists, the fused LINF conditioning kernel) -- on the double these are exact fp32 restatements,
    so the reference goldens still apply"""
    conv_mode = "x3"
    split = "f16x2"     # the engines then schedule the coupling_head / coupling_tail pair and the quad-major hand-over (x3 tensors stay lossless here)
