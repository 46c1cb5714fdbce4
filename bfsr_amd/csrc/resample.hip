// resample.hip -- resize (nearest / bilinear), 2x2 max-pool and affine-clamp kernels (gfx950).
// Index conventions follow the torch ops the reference calls:
//   nearest                 : src = min(floor(dst * r), in-1)                     (F.interpolate default)
//   bilinear, align False   : src = max(0, (dst+0.5)*r - 0.5), i1 = min(i0+1, in-1)
//   bilinear, align True    : src = dst * r with r = (in-1)/(out-1)
// and the bilinear blend is  hl0*(wl0*p00 + wl1*p01) + hl1*(wl0*p10 + wl1*p11).
#include <hip/hip_runtime.h>
#include "../../include/bfsr_hip.h"
#include "launch_util.h"

typedef _Float16 rs_half8 __attribute__((ext_vector_type(8)));

namespace {

// one output element (identical float operations on every path: the vector kernel below must give the same bits)
__device__ __forceinline__ float resize_one(const float* __restrict__ xc, int IH, int IW, int ry, int rx, int RH, int RW, int mode, float r_h, float r_w)
{
    if (!(ry >= 0 && ry < RH && rx >= 0 && rx < RW)) return 0.f;
    if (mode == 0) {
        int sy = (int)floorf((float)ry * r_h); sy = sy < IH - 1 ? sy : IH - 1;
        int sx = (int)floorf((float)rx * r_w); sx = sx < IW - 1 ? sx : IW - 1;
        return xc[(long long)sy * IW + sx];
    }
    float fy, fx;
    if (mode == 1) {
        fy = ((float)ry + 0.5f) * r_h - 0.5f; fy = fy < 0.f ? 0.f : fy;
        fx = ((float)rx + 0.5f) * r_w - 0.5f; fx = fx < 0.f ? 0.f : fx;
    } else {
        fy = (float)ry * r_h;
        fx = (float)rx * r_w;
    }
    int y0 = (int)fy; y0 = y0 < IH - 1 ? y0 : IH - 1;
    int x0 = (int)fx; x0 = x0 < IW - 1 ? x0 : IW - 1;
    const int y1 = y0 + (y0 < IH - 1 ? 1 : 0);
    const int x1 = x0 + (x0 < IW - 1 ? 1 : 0);
    const float hl1 = fy - (float)y0, hl0 = 1.f - hl1;
    const float wl1 = fx - (float)x0, wl0 = 1.f - wl1;
    const float p00 = xc[(long long)y0 * IW + x0], p01 = xc[(long long)y0 * IW + x1];
    const float p10 = xc[(long long)y1 * IW + x0], p11 = xc[(long long)y1 * IW + x1];
    return hl0 * (wl0 * p00 + wl1 * p01) + hl1 * (wl0 * p10 + wl1 * p11);
}

__global__ void resize_kernel(const float* __restrict__ x, long long x_bs, int IH, int IW, float* __restrict__ y,
                              long long y_bs, int OH, int OW, int RH, int RW, int oy0, int ox0, int C, int mode,
                              float r_h, float r_w)
{
    const long long n = (long long)C * OH * OW;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const int ox = (int)(i % OW);
    const int oy = (int)((i / OW) % OH);
    const int c = (int)(i / ((long long)OW * OH));
    const float* xc = x + (long long)b * x_bs + (long long)c * IH * IW;
    y[(long long)b * y_bs + i] = resize_one(xc, IH, IW, oy - oy0, ox - ox0, RH, RW, mode, r_h, r_w);
}

// the scalar kernel with 32-bit index arithmetic (C*OH*OW < 2^31: the launcher checks) -- for widths that are not multiples of four (the 257 x 257 query grids of
// LINF-LP: the 64-bit divisions of resize_kernel held the prior's bilinear up-sampling at 1.1 TB/s, 1.96 ms per config-5 pass)
__global__ void resize32_kernel(const float* __restrict__ x, long long x_bs, int IH, int IW, float* __restrict__ y,
                                long long y_bs, int OH, int OW, int RH, int RW, int oy0, int ox0, unsigned n, int mode,
                                float r_h, float r_w)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const unsigned row = i / (unsigned)OW;
    const int ox = (int)(i - row * (unsigned)OW);
    const unsigned c = row / (unsigned)OH;
    const int oy = (int)(row - c * (unsigned)OH);
    const float* xc = x + (long long)b * x_bs + (long long)c * IH * IW;
    y[(long long)b * y_bs + i] = resize_one(xc, IH, IW, oy - oy0, ox - ox0, RH, RW, mode, r_h, r_w);
}

// four consecutive outputs of one row per thread, one 16-byte store, 32-bit index arithmetic (OW % 4 == 0, 16-byte aligned rows, C*OH*OW < 2^31:
// the launcher checks).  The scalar kernel spends its time in two 64-bit divisions per element: 0.73 ms for the 906 MB LR-skip image of
// BASELINE config 5 (1.2 TB/s).
__global__ void resize4_kernel(const float* __restrict__ x, long long x_bs, int IH, int IW, float* __restrict__ y,
                               long long y_bs, int OH, int OW4, int RH, int RW, int oy0, int ox0, unsigned n4, int mode,
                               float r_h, float r_w)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int b = blockIdx.y;
    const unsigned row = i / (unsigned)OW4;
    const int ox = (int)(i - row * (unsigned)OW4) * 4;
    const unsigned c = row / (unsigned)OH;
    const int oy = (int)(row - c * (unsigned)OH);
    const float* xc = x + (long long)b * x_bs + (long long)c * IH * IW;
    float4 o;
    o.x = resize_one(xc, IH, IW, oy - oy0, ox - ox0, RH, RW, mode, r_h, r_w);
    o.y = resize_one(xc, IH, IW, oy - oy0, ox + 1 - ox0, RH, RW, mode, r_h, r_w);
    o.z = resize_one(xc, IH, IW, oy - oy0, ox + 2 - ox0, RH, RW, mode, r_h, r_w);
    o.w = resize_one(xc, IH, IW, oy - oy0, ox + 3 - ox0, RH, RW, mode, r_h, r_w);
    *reinterpret_cast<float4*>(y + (long long)b * y_bs + (long long)i * 4) = o;
}

__global__ void maxpool2_kernel(const float* __restrict__ x, long long x_bs, float* __restrict__ y, long long y_bs,
                                int C, int H, int W)
{
    const int Ho = H >> 1, Wo = W >> 1;
    const long long n = (long long)C * Ho * Wo;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const int ox = (int)(i % Wo);
    const int oy = (int)((i / Wo) % Ho);
    const int c = (int)(i / ((long long)Wo * Ho));
    const float* xi = x + (long long)b * x_bs + ((long long)c * H + 2 * oy) * W + 2 * ox;
    const float m0 = fmaxf(xi[0], xi[1]);
    const float m1 = fmaxf(xi[W], xi[W + 1]);
    y[(long long)b * y_bs + i] = fmaxf(m0, m1);
}

__global__ void axpb_clamp_kernel(const float* __restrict__ x, long long x_bs, const float* __restrict__ r,
                                  long long r_bs, float* __restrict__ y, long long y_bs, long long n, float a, float bb,
                                  float lo, float hi)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    float v = a * x[(long long)b * x_bs + i] + bb;
    if (r) v += r[(long long)b * r_bs + i];
    v = v < lo ? lo : v;
    v = v > hi ? hi : v;
    y[(long long)b * y_bs + i] = v;
}

// the same, four elements per thread with 16-byte accesses (n % 4 == 0 and all views 16-byte aligned: the launcher checks)
__global__ void axpb_clamp4_kernel(const float* __restrict__ x, long long x_bs, const float* __restrict__ r,
                                   long long r_bs, float* __restrict__ y, long long y_bs, long long n4, float a, float bb,
                                   float lo, float hi)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int b = blockIdx.y;
    const float4 xv = *reinterpret_cast<const float4*>(x + (long long)b * x_bs + i * 4);
    float v[4] = {a * xv.x + bb, a * xv.y + bb, a * xv.z + bb, a * xv.w + bb};
    if (r) {
        const float4 rv = *reinterpret_cast<const float4*>(r + (long long)b * r_bs + i * 4);
        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = v[k] < lo ? lo : v[k]; v[k] = v[k] > hi ? hi : v[k]; }
    *reinterpret_cast<float4*>(y + (long long)b * y_bs + i * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

// ---- h2 forms of the learned priors' glue (round 6): the UNet levels that run on the LDS-DMA kernels keep their activations as h2 tensors
// ([B][C/8][plane hi, lo][H][W][8] fp16); pooling and bilinear up-sampling used to go through fp32 (h2_unpack -> maxpool2 -> h2_pack; resize -> h2_pack:
// 4.4 ms of a config-5 pass).  Same float operations as those launches (unpack = hi + lo, fmaxf tree of maxpool2_kernel, resize_one, h2_pack's split with the
// pinned hi value and its range flag), so the results are the same bits (tests/test_hip_ops.py::test_h2_glue_equals_launches).
__device__ __forceinline__ void rs_split2(float v, _Float16& h, _Float16& l)
{
    const float hf = bfsr::pin_f16(v);
    h = (_Float16)hf;
    l = (_Float16)(v - hf);
}

// one thread = one channel octet of one output pixel
__global__ void resize_h2_kernel(const float* __restrict__ x, long long x_bs, int IH, int IW, unsigned short* __restrict__ y, long long y_bs, int OH, int OW, int RH, int RW,
                                 int oy0, int ox0, int C8, unsigned n, int mode, float r_h, float r_w, unsigned* flag)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const unsigned HWo = (unsigned)(OH * OW);
    const unsigned oct = i / HWo, pix = i - oct * HWo;
    const int oy = (int)(pix / (unsigned)OW), ox = (int)(pix - (pix / (unsigned)OW) * (unsigned)OW);
    const float* xb = x + (long long)b * x_bs + (long long)oct * 8 * IH * IW;
    rs_half8 h8, l8;
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = resize_one(xb + (long long)j * IH * IW, IH, IW, oy - oy0, ox - ox0, RH, RW, mode, r_h, r_w);
        _Float16 h, l;
        rs_split2(v, h, l);
        h8[j] = h; l8[j] = l;
        amax = fmaxf(amax, fabsf(v));
    }
    unsigned short* yb = y + (long long)b * y_bs + ((long long)oct * 2 * HWo + pix) * 8;
    *reinterpret_cast<rs_half8*>(yb) = h8;
    *reinterpret_cast<rs_half8*>(yb + (long long)HWo * 8) = l8;
    if (flag && !(amax < 65504.f)) atomicOr(flag, 1u);
}

// 2 x 2 max-pool of an h2 tensor -> h2 tensor and / or fp32 NCHW tensor; one thread = one channel octet of one output pixel
__global__ void maxpool2_h2_kernel(const unsigned short* __restrict__ x, long long x_bs, unsigned short* __restrict__ yh, long long yh_bs, float* __restrict__ yf,
                                   long long yf_bs, int C8, int H, int W, unsigned n, unsigned* flag)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const int Ho = H >> 1, Wo = W >> 1;
    const unsigned HWo = (unsigned)(Ho * Wo);
    const long long HW = (long long)H * W;
    const unsigned oct = i / HWo, pix = i - oct * HWo;
    const int oy = (int)(pix / (unsigned)Wo), ox = (int)(pix - (pix / (unsigned)Wo) * (unsigned)Wo);
    const unsigned short* xb = x + (long long)b * x_bs + (long long)oct * 2 * HW * 8;
    float v[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long p = (long long)(2 * oy + (q >> 1)) * W + 2 * ox + (q & 1);
        const rs_half8 h = *reinterpret_cast<const rs_half8*>(xb + p * 8);
        const rs_half8 l = *reinterpret_cast<const rs_half8*>(xb + (HW + p) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[q][j] = (float)h[j] + (float)l[j];               // h2_unpack
    }
    rs_half8 h8, l8;
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float m = fmaxf(fmaxf(v[0][j], v[1][j]), fmaxf(v[2][j], v[3][j]));      // maxpool2_kernel
        if (yf) yf[(long long)b * yf_bs + ((long long)oct * 8 + j) * HWo + pix] = m;
        _Float16 h, l;
        rs_split2(m, h, l);
        h8[j] = h; l8[j] = l;
        amax = fmaxf(amax, fabsf(m));
    }
    if (yh) {
        unsigned short* yb = yh + (long long)b * yh_bs + ((long long)oct * 2 * HWo + pix) * 8;
        *reinterpret_cast<rs_half8*>(yb) = h8;
        *reinterpret_cast<rs_half8*>(yb + (long long)HWo * 8) = l8;
        if (flag && !(amax < 65504.f)) atomicOr(flag, 1u);
    }
}

// ---- LINF-LP glue, fused (round 6).  The reference's harness does these steps as separate torch ops (LINF-LP/test.py:168-171, 217;
// datasets/wrappers.py:203-228); as separate launches they were 9 % of a config-5 pass (fold 0.84 ms twice, skip resize 0.39, two axpb 0.82; lr_up / down /
// up2 / residual / unfold 2.5 ms), each moving the 906 MB HR image through HBM once more.  Every value below is formed by the SAME float operations in the
// same order as resize_one / axpb_clamp / patch_fold form it (no contraction: -ffp-contract=off), so the fused results are the unfused BITS
// (tests/test_linf_gpu.py::test_fused_glue_equals_launches).
// bilinear, align_corners = False (mode 1 of resize_one) over a source given as a functor src(y, x)
template <class F>
__device__ __forceinline__ float bilerp1(F src, int IH, int IW, int ry, int rx, float r_h, float r_w)
{
    float fy = ((float)ry + 0.5f) * r_h - 0.5f; fy = fy < 0.f ? 0.f : fy;
    float fx = ((float)rx + 0.5f) * r_w - 0.5f; fx = fx < 0.f ? 0.f : fx;
    int y0 = (int)fy; y0 = y0 < IH - 1 ? y0 : IH - 1;
    int x0 = (int)fx; x0 = x0 < IW - 1 ? x0 : IW - 1;
    const int y1 = y0 + (y0 < IH - 1 ? 1 : 0);
    const int x1 = x0 + (x0 < IW - 1 ? 1 : 0);
    const float hl1 = fy - (float)y0, hl0 = 1.f - hl1;
    const float wl1 = fx - (float)x0, wl0 = 1.f - wl1;
    const float p00 = src(y0, x0), p01 = src(y0, x1);
    const float p10 = src(y1, x0), p11 = src(y1, x1);
    return hl0 * (wl0 * p00 + wl1 * p01) + hl1 * (wl0 * p10 + wl1 * p11);
}

// fold + crop + LR skip + output clamp: raw = (1 * fold(p)[.., :H, :W] + 0) + bilinear(inp -> H x W); out01 = clamp(0.5 raw + 0.5, 0, 1).  VEC consecutive
// pixels of a row per thread (VEC = 4: 16-byte stores)
template <int VEC>
__global__ void linf_fold_skip_kernel(const float* __restrict__ p, long long p_bs, const float* __restrict__ inp, long long inp_bs, float* __restrict__ raw,
                                      long long raw_bs, float* __restrict__ out01, long long out_bs, int C, int qh, int qw, int H, int WV, int ps, int h, int w,
                                      unsigned n, float r_h, float r_w)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const unsigned row = i / (unsigned)WV;
    const int x4 = (int)(i - row * (unsigned)WV) * VEC;
    const unsigned c = row / (unsigned)H;
    const int y = (int)(row - c * (unsigned)H);
    const unsigned qy = (unsigned)y / (unsigned)ps;
    const int ky = y - (int)qy * ps;
    const float* pb = p + (long long)b * p_bs;
    const float* ic = inp + (long long)b * inp_bs + (long long)c * h * w;
    float rv[VEC], ov[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const int x = x4 + k;
        const unsigned qx = (unsigned)x / (unsigned)ps;
        const int kx = x - (int)qx * ps;
        const int ch = ((int)c * ps + ky) * ps + kx;
        const float pv = pb[((long long)ch * qh + qy) * qw + qx];
        const float skip = bilerp1([&](int yy, int xx) { return ic[(long long)yy * w + xx]; }, h, w, y, x, r_h, r_w);
        float v = 1.0f * pv + 0.0f;                                      // axpb_clamp(pred, a = 1, b = 0, r = skip)
        v += skip;
        rv[k] = v;
        float o = 0.5f * v + 0.5f;                                       // axpb_clamp(raw, 0.5, 0.5, 0, 1)
        o = o < 0.f ? 0.f : o;
        o = o > 1.f ? 1.f : o;
        ov[k] = o;
    }
    const long long off = (long long)i * VEC;
    if constexpr (VEC == 4) {
        if (raw) *reinterpret_cast<float4*>(raw + (long long)b * raw_bs + off) = make_float4(rv[0], rv[1], rv[2], rv[3]);
        if (out01) *reinterpret_cast<float4*>(out01 + (long long)b * out_bs + off) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    } else {
        if (raw) raw[(long long)b * raw_bs + off] = rv[0];
        if (out01) out01[(long long)b * out_bs + off] = ov[0];
    }
}

// lr_up(y, x) of the input prep: bilinear(2 * inp01 - 1 -> H x W), formed on the fly
struct LrUp {
    const float* ic; int h, w; float r_h, r_w;
    __device__ __forceinline__ float operator()(int y, int x) const
    {
        const float* s = ic; const int ww = w;
        return bilerp1([=](int yy, int xx) { return 2.0f * s[(long long)yy * ww + xx] + (-1.0f); }, h, w, y, x, r_h, r_w);
    }
};

// down = bilinear(lr_up -> h x w): one LR element per thread
__global__ void linf_prep_down_kernel(const float* __restrict__ inp01, long long in_bs, float* __restrict__ down, long long down_bs, int C, int h, int w, int H, int W,
                                      float ru_h, float ru_w, float rd_h, float rd_w)
{
    const unsigned n = (unsigned)(C * h * w);
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const unsigned row = i / (unsigned)w;
    const int x = (int)(i - row * (unsigned)w);
    const unsigned c = row / (unsigned)h;
    const int y = (int)(row - c * (unsigned)h);
    const LrUp up{inp01 + (long long)b * in_bs + (long long)c * h * w, h, w, ru_h, ru_w};
    down[(long long)b * down_bs + i] = bilerp1(up, H, W, y, x, rd_h, rd_w);
}

// gt = unfold(zero-pad(lr_up - up(down))).  One wave = one row (channel ch, patch row qy) of the unfolded tensor [C*ps*ps, qh, qw], lanes stride over qx: the index
// divisions happen once per row, not once per element (one element per thread with five 32-bit divisions ran at 1.07 TB/s: 0.85 ms for the 913 MB of config 5)
__global__ void linf_prep_residual_kernel(const float* __restrict__ inp01, long long in_bs, const float* __restrict__ down, long long down_bs, float* __restrict__ gt,
                                          long long gt_bs, int C, int h, int w, int H, int W, int qh, int qw, int ps, unsigned nrows, float ru_h, float ru_w)
{
    const unsigned row = blockIdx.x * blockDim.y + threadIdx.y;
    if (row >= nrows) return;
    const int b = blockIdx.y;
    const unsigned ch = row / (unsigned)qh;
    const int qy = (int)(row - ch * (unsigned)qh);
    const unsigned pp = (unsigned)(ps * ps);
    const unsigned c = ch / pp, r = ch - c * pp;
    const unsigned ky = r / (unsigned)ps;
    const int kx = (int)(r - ky * (unsigned)ps);
    const int y = qy * ps + (int)ky;
    const LrUp up{inp01 + (long long)b * in_bs + (long long)c * h * w, h, w, ru_h, ru_w};
    const float* dc = down + (long long)b * down_bs + (long long)c * h * w;
    const int ww = w;
    float* grow = gt + (long long)b * gt_bs + (long long)row * qw;
    for (int qx = threadIdx.x; qx < qw; qx += blockDim.x) {
        const int x = qx * ps + kx;
        float v = 0.f;
        if (y < H && x < W) {
            const float lu = up(y, x);
            const float u2 = bilerp1([=](int yy, int xx) { return dc[(long long)yy * ww + xx]; }, h, w, y, x, ru_h, ru_w);
            float t = -1.0f * u2 + 0.0f;                                 // axpb_clamp(up2, a = -1, b = 0, r = lr_up)
            t += lu;
            v = t;
        }
        grow[qx] = v;
    }
}

}  // namespace

extern "C" int bfsr_resize_h2(const float* x, long long x_bs, int IH, int IW, unsigned short* y, long long y_bs, int OH, int OW, int RH, int RW, int oy0, int ox0,
                              int B, int C, int mode, float r_h, float r_w, unsigned* flag, void* stream)
{
    if (!x || !y || mode < 0 || mode > 2 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || B <= 0 || C <= 0 || (C & 7)) return -1;
    if ((reinterpret_cast<unsigned long long>(y) & 15) || (y_bs & 7)) return -1;
    const long long n = (long long)(C / 8) * OH * OW;
    if (n >= (1LL << 31)) return -1;
    hipLaunchKernelGGL(resize_h2_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, IH, IW, y, y_bs, OH, OW,
                       RH, RW, oy0, ox0, C / 8, (unsigned)n, mode, r_h, r_w, flag);
    return (int)hipGetLastError();
}

extern "C" int bfsr_maxpool2_h2(const unsigned short* x, long long x_bs, unsigned short* y_h2, long long yh_bs, float* y_f32, long long yf_bs, int B, int C, int H, int W,
                                unsigned* flag, void* stream)
{
    if (!x || (!y_h2 && !y_f32) || H < 2 || W < 2 || B <= 0 || C <= 0 || (C & 7)) return -1;
    if ((reinterpret_cast<unsigned long long>(x) & 15) || (x_bs & 7)) return -1;
    if (y_h2 && ((reinterpret_cast<unsigned long long>(y_h2) & 15) || (yh_bs & 7))) return -1;
    const long long n = (long long)(C / 8) * (H / 2) * (W / 2);
    if (n >= (1LL << 31)) return -1;
    hipLaunchKernelGGL(maxpool2_h2_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, y_h2, yh_bs, y_f32,
                       yf_bs, C / 8, H, W, (unsigned)n, flag);
    return (int)hipGetLastError();
}

extern "C" int bfsr_linf_fold_skip(const float* p, long long p_bs, const float* inp, long long inp_bs, float* raw, long long raw_bs, float* out01, long long out_bs,
                                   int B, int C, int qh, int qw, int H, int W, int ps, int h, int w, float r_h, float r_w, void* stream)
{
    if (!p || !inp || (!raw && !out01) || B <= 0 || C <= 0 || ps <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0 || H > ps * qh || W > ps * qw) return -1;
    const long long n = (long long)C * H * W;
    if (n >= (1LL << 31)) return -1;
    const unsigned long long al = reinterpret_cast<unsigned long long>(raw) | reinterpret_cast<unsigned long long>(out01);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if ((W & 3) == 0 && (al & 15) == 0 && (((raw ? raw_bs : 0) | (out01 ? out_bs : 0)) & 3) == 0) {
        const unsigned n4 = (unsigned)(n / 4);
        hipLaunchKernelGGL(linf_fold_skip_kernel<4>, dim3((n4 + 255) / 256, (unsigned)B), dim3(256), 0, st, p, p_bs, inp, inp_bs, raw, raw_bs, out01, out_bs, C, qh, qw, H,
                           W / 4, ps, h, w, n4, r_h, r_w);
    } else {
        hipLaunchKernelGGL(linf_fold_skip_kernel<1>, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, st, p, p_bs, inp, inp_bs, raw, raw_bs, out01, out_bs, C, qh,
                           qw, H, W, ps, h, w, (unsigned)n, r_h, r_w);
    }
    return (int)hipGetLastError();
}

extern "C" int bfsr_linf_prep_down(const float* inp01, long long in_bs, float* down, long long down_bs, int B, int C, int h, int w, int H, int W,
                                   float ru_h, float ru_w, float rd_h, float rd_w, void* stream)
{
    if (!inp01 || !down || B <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || (long long)C * h * w >= (1LL << 31)) return -1;
    const unsigned n = (unsigned)(C * h * w);
    hipLaunchKernelGGL(linf_prep_down_kernel, dim3((n + 255) / 256, (unsigned)B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), inp01, in_bs, down, down_bs, C, h, w, H,
                       W, ru_h, ru_w, rd_h, rd_w);
    return (int)hipGetLastError();
}

extern "C" int bfsr_linf_prep_residual(const float* inp01, long long in_bs, const float* down, long long down_bs, float* gt, long long gt_bs, int B, int C, int h, int w,
                                       int H, int W, int qh, int qw, int ps, float ru_h, float ru_w, void* stream)
{
    if (!inp01 || !down || !gt || B <= 0 || C <= 0 || ps <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || H > ps * qh || W > ps * qw) return -1;
    const long long n = (long long)C * ps * ps * qh * qw;
    if (n >= (1LL << 31)) return -1;
    const unsigned nrows = (unsigned)(C * ps * ps * qh);
    hipLaunchKernelGGL(linf_prep_residual_kernel, dim3((nrows + 3) / 4, (unsigned)B), dim3(64, 4), 0, reinterpret_cast<hipStream_t>(stream), inp01, in_bs, down,
                       down_bs, gt, gt_bs, C, h, w, H, W, qh, qw, ps, nrows, ru_h, ru_w);
    return (int)hipGetLastError();
}

extern "C" int bfsr_resize(const float* x, long long x_bs, int IH, int IW, float* y, long long y_bs, int OH, int OW,
                           int RH, int RW, int oy0, int ox0, int B, int C, int mode, float r_h, float r_w, void* stream)
{
    if (!x || !y || mode < 0 || mode > 2 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return -1;
    const long long n = (long long)C * OH * OW;
    if ((OW & 3) == 0 && n < (1LL << 31) && (reinterpret_cast<unsigned long long>(y) & 15) == 0 && (y_bs & 3) == 0) {
        const unsigned n4 = (unsigned)(n / 4);
        hipLaunchKernelGGL(resize4_kernel, dim3((n4 + 255) / 256, (unsigned)B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, IH, IW, y,
                           y_bs, OH, OW / 4, RH, RW, oy0, ox0, n4, mode, r_h, r_w);
        return (int)hipGetLastError();
    }
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)B);
    if (n < (1LL << 31)) {
        hipLaunchKernelGGL(resize32_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, IH, IW, y,
                           y_bs, OH, OW, RH, RW, oy0, ox0, (unsigned)n, mode, r_h, r_w);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(resize_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, IH, IW, y,
                       y_bs, OH, OW, RH, RW, oy0, ox0, C, mode, r_h, r_w);
    return (int)hipGetLastError();
}

extern "C" int bfsr_maxpool2(const float* x, long long x_bs, float* y, long long y_bs, int B, int C, int H, int W,
                             void* stream)
{
    if (!x || !y || H < 2 || W < 2) return -1;
    const long long n = (long long)C * (H / 2) * (W / 2);
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(maxpool2_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, y, y_bs, C,
                       H, W);
    return (int)hipGetLastError();
}

extern "C" int bfsr_axpb_clamp(const float* x, long long x_bs, const float* r, long long r_bs, float* y, long long y_bs,
                               int B, int C, int H, int W, float a, float b, float lo, float hi, void* stream)
{
    if (!x || !y) return -1;
    const long long n = (long long)C * H * W;
    const unsigned long long al = reinterpret_cast<unsigned long long>(x) | reinterpret_cast<unsigned long long>(y) | reinterpret_cast<unsigned long long>(r);
    if ((n & 3) == 0 && (al & 15) == 0 && ((x_bs | y_bs | (r ? r_bs : 0)) & 3) == 0) {
        hipLaunchKernelGGL(axpb_clamp4_kernel, dim3((unsigned)((n / 4 + 255) / 256), (unsigned)B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs,
                           r, r_bs, y, y_bs, n / 4, a, b, lo, hi);
        return (int)hipGetLastError();
    }
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(axpb_clamp_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, r, r_bs,
                       y, y_bs, n, a, b, lo, hi);
    return (int)hipGetLastError();
}

extern "C" int bfsr_abi_version(void) { return BFSR_ABI_VERSION; }
