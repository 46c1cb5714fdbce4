// linf_ops.hip -- LINF-LP specific kernels (gfx950, wave64):
//   bfsr_linf_features : local-ensemble Fourier feature build (LINF-LP/models/linf.py:344-388)
//   bfsr_linf_flow     : local implicit coupling flow forward / inverse (LINF-LP/models/flow.py:44-63)
//   bfsr_patch_fold / bfsr_patch_unfold : ps x ps patch <-> pixel rearrangement (linf.py:401-406, wrappers.py:224-228)
//   bfsr_conv2d_direct : small strided conv (prior `lr_proj.0`: 3 -> in_chans, k3 s3 p1; LINF-LP/models/unet.py:118)
// All tensors NCHW fp32 views over the query grid (one "pixel" = one query point); lanes run along the
// contiguous plane so every per-channel access of a wave is coalesced.
#include <hip/hip_runtime.h>
#include "../../include/bfsr_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// One thread = one (query point, neighbour k).  For hidden_dim HD: cf = [coef(HD) | freq(HD)] on the LR grid.
// out[b][k*HD + c][q]        = (w_k * coef_k[c])        * cos(pi * f_c)      c in [0, HD/2)
// out[b][k*HD + HD/2 + c][q] = (w_k * coef_k[HD/2 + c]) * sin(pi * f_c)
//   f_c = freq_k[c] * rel_y + freq_k[HD/2 + c] * rel_x + phase[c][0]*cell_y*h + phase[c][1]*cell_x*w
//   w_k = area_{3-k} / sum_k area_k,  area_k = |rel_y * rel_x| + 1e-9   (diagonal swap, linf.py:379-380)
__global__ __launch_bounds__(256) void linf_features_kernel(BfsrLinfFeatArgs a)
{
    const long long NQ = (long long)a.qh * a.qw;
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= NQ) return;
    const int k = blockIdx.y;                 // neighbour: (vx,vy) = (-1,-1), (-1,1), (1,-1), (1,1)
    const int b = blockIdx.z;
    const int h = a.h, w = a.w, HD = a.hidden, HH = a.hidden / 2;
    const float cy = a.coord[((long long)b * NQ + q) * 2 + 0];
    const float cx = a.coord[((long long)b * NQ + q) * 2 + 1];
    const float fh = (float)h, fw = (float)w;

    float area[4], rel_y = 0.f, rel_x = 0.f;
    int iy = 0, ix = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float y = cy + ((j & 2) ? a.dy_pos : a.dy_neg);
        float x = cx + ((j & 1) ? a.dx_pos : a.dx_neg);
        y = fminf(fmaxf(y, a.clamp_lo), a.clamp_hi);
        x = fminf(fmaxf(x, a.clamp_lo), a.clamp_hi);
        // F.grid_sample(mode='nearest', align_corners=False): pix = ((c+1)*size-1)/2, nearbyint, zero outside
        int jy = (int)nearbyintf(((y + 1.f) * fh - 1.f) / 2.f);
        int jx = (int)nearbyintf(((x + 1.f) * fw - 1.f) / 2.f);
        jy = min(max(jy, 0), h - 1);
        jx = min(max(jx, 0), w - 1);
        // centre of that LR cell: make_coord = -1 + r + (2r)*i with r = 1/n  (utils.py:113-115)
        const float qy = a.cy0 + a.cy1 * (float)jy;
        const float qx = a.cx0 + a.cx1 * (float)jx;
        const float ry_rel = (cy - qy) * fh, rx_rel = (cx - qx) * fw;
        area[j] = fabsf(ry_rel * rx_rel) + 1e-9f;
        if (j == k) { rel_y = ry_rel; rel_x = rx_rel; iy = jy; ix = jx; }
    }
    const float tot = ((area[0] + area[1]) + area[2]) + area[3];
    const float wk = area[3 - k] / tot;
    const float cell_y = a.cell[b * 2 + 0] * fh, cell_x = a.cell[b * 2 + 1] * fw;

    const long long hw = (long long)h * w;
    const float* cfp = a.cf + (long long)b * a.cf_bs + (long long)iy * w + ix;
    float* op = a.out + (long long)b * a.out_bs + (long long)(k * HD) * NQ + q;
    const float PI = 3.14159274101257324f;       // float32(np.pi)
    for (int c = 0; c < HH; ++c) {
        const float co0 = cfp[(long long)c * hw], co1 = cfp[(long long)(HH + c) * hw];
        const float f0 = cfp[(long long)(HD + c) * hw], f1 = cfp[(long long)(HD + HH + c) * hw];
        float f = f0 * rel_y + f1 * rel_x;
        f = f + (cell_y * a.phase[c * 2 + 0] + cell_x * a.phase[c * 2 + 1]);
        float s, cs;
        sincosf(PI * f, &s, &cs);
        op[(long long)c * NQ] = (wk * co0) * cs;
        op[(long long)(HH + c) * NQ] = (wk * co1) * s;
    }
}

// ---------------------------------------------------------------------------------------------------
// Flow over D-vectors (D = 3*ps*ps), one thread per query point.  lin_w: [L+1][D][D] (forward: W; inverse: W^-1),
// lin_b: [L+1][D]; affine_info ai [B, 2*D*L, qh, qw]: for layer i  s = ai[2Di : 2Di+D], shift = ai[2Di+D : 2D(i+1)].
//   forward : for i<L: x = W_i x + b_i ; x = x*scale_i + shift_i ;  then x = W_L x + b_L
//   inverse : x = Winv_L (x - b_L) ; for i=L-1..0: x = (x - shift_i)/scale_i ; x = Winv_i (x - b_i)
template <int D>
__global__ __launch_bounds__(256) void linf_flow_kernel(BfsrLinfFlowArgs a)
{
    const long long NQ = (long long)a.qh * a.qw;
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= NQ) return;
    const int b = blockIdx.y, L = a.layers;
    const float* xi = a.x + (long long)b * a.x_bs + q;
    const float* ai = a.ai + (long long)b * a.ai_bs + q;
    float x[D], y[D];
    // conditioning values of one flow layer: lv[d] = raw scale, lv[D + d] = shift.  ai_fmt 1 = the quad-major layout the fused MLP
    // kernel writes ([layers][QB quads][NQ][4], a layer's 2*D values padded to QB*4): QB 16-byte loads per layer instead of 2*D
    // four-byte ones (the row-major form streamed at 2.4 TB/s, load-issue bound: 540 loads per query point)
    constexpr int QB = (2 * D + 3) / 4;
    float lv[QB * 4];
    const float4* aq = reinterpret_cast<const float4*>(a.ai + (long long)b * a.ai_bs) + q;
    auto load_layer = [&](int i) {
        if (a.ai_fmt == 1) {
#pragma unroll
            for (int k = 0; k < QB; ++k) {
                const float4 v = aq[(long long)(i * QB + k) * NQ];
                lv[4 * k] = v.x; lv[4 * k + 1] = v.y; lv[4 * k + 2] = v.z; lv[4 * k + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int d = 0; d < 2 * D; ++d) lv[d] = ai[(long long)(2 * D * i + d) * NQ];
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = xi[(long long)d * NQ];

    auto linear = [&](int layer, bool sub_bias_first) {
        const float* __restrict__ W = a.lin_w + (long long)layer * D * D;
        const float* __restrict__ bb = a.lin_b + (long long)layer * D;
        if (sub_bias_first) {
#pragma unroll
            for (int d = 0; d < D; ++d) x[d] -= bb[d];
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < D; ++j) s = fmaf(W[i * D + j], x[j], s);
            y[i] = sub_bias_first ? s : s + bb[i];
        }
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = y[d];
    };

    if (!a.reverse) {
        float ld = a.logdet_const;
        const bool want_lp = a.log_p != nullptr;
        for (int i = 0; i < L; ++i) {
            load_layer(i);
            linear(i, false);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float sr = lv[d], sh = lv[D + d];
                const float sc = 1.f / (1.f + expf(-(sr + 2.f))) + a.eps;
                x[d] = x[d] * sc + sh;
                if (want_lp) ld += logf(sc);
            }
        }
        linear(L, false);
        if (want_lp) {
            float base = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) base += -0.5f * (x[d] * x[d] + 1.8378770664093453f);
            a.log_p[(long long)b * NQ + q] = ld + base;
        }
    } else if (a.reverse == 2) {
        // vector-Jacobian product of the INVERSE flow w.r.t. its input z (the conditioning is fixed, so the inverse is affine in z):
        // J = Winv_0 D_0 ... Winv_{L-1} D_{L-1} Winv_L with D_i = diag(1/scale_i)  =>  g_z = Winv_L^T D_{L-1} Winv_{L-1}^T ... D_0 Winv_0^T g.
        // The caller passes lin_w[i] = Winv_i^T; biases and shifts do not enter.  (LINF-LP/train.py:143: the image-space loss of the
        // latent module back-propagates through query_rgb into z_lr_learned.)
        auto matvec = [&](int layer) {
            const float* __restrict__ W = a.lin_w + (long long)layer * D * D;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < D; ++j) s = fmaf(W[i * D + j], x[j], s);
                y[i] = s;
            }
#pragma unroll
            for (int d = 0; d < D; ++d) x[d] = y[d];
        };
        for (int i = 0; i < L; ++i) {
            load_layer(i);
            matvec(i);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float sr = lv[d];
                x[d] = x[d] / (1.f / (1.f + expf(-(sr + 2.f))) + a.eps);
            }
        }
        matvec(L);
    } else {
        linear(L, true);
        for (int i = L - 1; i >= 0; --i) {
            load_layer(i);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float sr = lv[d], sh = lv[D + d];
                x[d] = (x[d] - sh) / (1.f / (1.f + expf(-(sr + 2.f))) + a.eps);
            }
            linear(i, true);
        }
    }
    float* yo = a.y + (long long)b * a.y_bs + q;
#pragma unroll
    for (int d = 0; d < D; ++d) yo[(long long)d * NQ] = x[d];
}

// ---------------------------------------------------------------------------------------------------
// fold: p [B, C*ps*ps, qh, qw] -> img [B, C, H, W] (H <= ps*qh: cropped), channel c*ps*ps + ky*ps + kx
__global__ void patch_fold_kernel(const float* __restrict__ p, long long p_bs, float* __restrict__ img, long long img_bs,
                                  int C, int qh, int qw, int H, int W, int ps, int unfold)
{
    const long long n = (long long)C * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (!unfold) {
        if (i >= n) return;
        const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)(i / ((long long)W * H));
        const int ch = c * ps * ps + (y % ps) * ps + (x % ps);
        img[(long long)b * img_bs + i] = p[(long long)b * p_bs + ((long long)ch * qh + y / ps) * qw + x / ps];
    } else {
        // unfold: img [B,C,H,W] zero padded to (ps*qh, ps*qw) -> p
        const long long np = (long long)C * ps * ps * qh * qw;
        if (i >= np) return;
        const int qx = (int)(i % qw), qy = (int)((i / qw) % qh);
        const int ch = (int)(i / ((long long)qw * qh));
        const int c = ch / (ps * ps), r = ch % (ps * ps);
        const int y = qy * ps + r / ps, x = qx * ps + r % ps;
        float v = 0.f;
        if (y < H && x < W) v = img[(long long)b * img_bs + ((long long)c * H + y) * W + x];
        const_cast<float*>(p)[(long long)b * p_bs + i] = v;
    }
}

// small direct convolution: y[b,co,oy,ox] = bias[co] + sum x[b,ci,oy*s-pad+ky,ox*s-pad+kx] * w[co,ci,ky,kx]; optional LeakyReLU
__global__ void conv2d_direct_kernel(const float* __restrict__ x, long long x_bs, const float* __restrict__ w,
                                     const float* __restrict__ bias, float* __restrict__ y, long long y_bs, int Cin,
                                     int Cout, int H, int W, int OH, int OW, int KS, int stride, int pad, int act,
                                     float slope)
{
    const long long n = (long long)Cout * OH * OW;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH), co = (int)(i / ((long long)OW * OH));
    float acc = 0.f;
    for (int ci = 0; ci < Cin; ++ci)
        for (int ky = 0; ky < KS; ++ky) {
            const int iy = oy * stride - pad + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < KS; ++kx) {
                const int ix = ox * stride - pad + kx;
                if (ix < 0 || ix >= W) continue;
                acc = fmaf(x[(long long)b * x_bs + ((long long)ci * H + iy) * W + ix],
                           w[((co * Cin + ci) * KS + ky) * KS + kx], acc);
            }
        }
    if (bias) acc += bias[co];
    if (act == BFSR_ACT_RELU) acc = fmaxf(acc, 0.f);
    else if (act == BFSR_ACT_LRELU) acc = acc > 0.f ? acc : acc * slope;
    y[(long long)b * y_bs + i] = acc;
}


// F.grid_sample(x, coord.flip(-1), mode='bilinear', padding_mode='border', align_corners=False) added to `acc`
// (LINF.query_rgb skip, LINF-LP/models/linf.py:193-194): coord [B,qh,qw,2] holds (y, x) in [-1,1].
__global__ void grid_sample_add_kernel(const float* __restrict__ x, long long x_bs, const float* __restrict__ coord,
                                       const float* __restrict__ acc, long long acc_bs, float* __restrict__ out,
                                       long long out_bs, int C, int h, int w, long long NQ)
{
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= NQ) return;
    const int b = blockIdx.y;
    const float cy = coord[((long long)b * NQ + q) * 2 + 0], cx = coord[((long long)b * NQ + q) * 2 + 1];
    float ix = ((cx + 1.f) * (float)w - 1.f) / 2.f, iy = ((cy + 1.f) * (float)h - 1.f) / 2.f;
    ix = fminf((float)(w - 1), fmaxf(ix, 0.f));                 // padding_mode='border': clip the coordinates
    iy = fminf((float)(h - 1), fmaxf(iy, 0.f));
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float nw = ((float)x1 - ix) * ((float)y1 - iy), ne = (ix - (float)x0) * ((float)y1 - iy);
    const float sw = ((float)x1 - ix) * (iy - (float)y0), se = (ix - (float)x0) * (iy - (float)y0);
    const bool x1ok = x1 < w, y1ok = y1 < h;
    for (int c = 0; c < C; ++c) {
        const float* xc = x + (long long)b * x_bs + (long long)c * h * w;
        float v = xc[(long long)y0 * w + x0] * nw;
        if (x1ok) v += xc[(long long)y0 * w + x1] * ne;
        if (y1ok) v += xc[(long long)y1 * w + x0] * sw;
        if (x1ok && y1ok) v += xc[(long long)y1 * w + x1] * se;
        const long long o = (long long)c * NQ + q;
        out[(long long)b * out_bs + o] = acc[(long long)b * acc_bs + o] + v;
    }
}

}  // namespace

extern "C" int bfsr_linf_features(const BfsrLinfFeatArgs* a, void* stream)
{
    if (!a || !a->cf || !a->coord || !a->cell || !a->phase || !a->out || (a->hidden & 1)) return -1;
    const long long NQ = (long long)a->qh * a->qw;
    dim3 grid((unsigned)((NQ + 255) / 256), 4, (unsigned)a->B);
    hipLaunchKernelGGL(linf_features_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a);
    return (int)hipGetLastError();
}

extern "C" int bfsr_linf_flow(const BfsrLinfFlowArgs* a, void* stream)
{
    if (!a || !a->x || !a->y || !a->ai || !a->lin_w || !a->lin_b) return -1;
    if (a->ai_fmt != 0 && a->ai_fmt != 1) return -1;
    if (a->ai_fmt == 1 && ((reinterpret_cast<unsigned long long>(a->ai) & 15) || (a->ai_bs & 3))) return -1;
    const long long NQ = (long long)a->qh * a->qw;
    dim3 grid((unsigned)((NQ + 255) / 256), (unsigned)a->B);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (a->D == 27) hipLaunchKernelGGL((linf_flow_kernel<27>), grid, dim3(256), 0, st, *a);
    else if (a->D == 3) hipLaunchKernelGGL((linf_flow_kernel<3>), grid, dim3(256), 0, st, *a);
    else if (a->D == 12) hipLaunchKernelGGL((linf_flow_kernel<12>), grid, dim3(256), 0, st, *a);
    else return -1;
    return (int)hipGetLastError();
}

extern "C" int bfsr_patch_fold(const float* p, long long p_bs, float* img, long long img_bs, int B, int C, int qh,
                               int qw, int H, int W, int ps, void* stream)
{
    if (!p || !img || ps < 1 || H > ps * qh || W > ps * qw) return -1;
    const long long n = (long long)C * H * W;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(patch_fold_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, p_bs, img, img_bs,
                       C, qh, qw, H, W, ps, 0);
    return (int)hipGetLastError();
}

extern "C" int bfsr_patch_unfold(const float* img, long long img_bs, float* p, long long p_bs, int B, int C, int qh,
                                 int qw, int H, int W, int ps, void* stream)
{
    if (!p || !img || ps < 1 || H > ps * qh || W > ps * qw) return -1;
    const long long n = (long long)C * ps * ps * qh * qw;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(patch_fold_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, p_bs,
                       const_cast<float*>(img), img_bs, C, qh, qw, H, W, ps, 1);
    return (int)hipGetLastError();
}

extern "C" int bfsr_conv2d_direct(const float* x, long long x_bs, const float* w, const float* bias, float* y,
                                  long long y_bs, int B, int Cin, int Cout, int H, int W, int KS, int stride, int pad,
                                  int act, float slope, void* stream)
{
    if (!x || !w || !y || stride < 1 || KS < 1) return -1;
    const int OH = (H + 2 * pad - KS) / stride + 1, OW = (W + 2 * pad - KS) / stride + 1;
    if (OH <= 0 || OW <= 0) return -1;
    const long long n = (long long)Cout * OH * OW;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(conv2d_direct_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, w, bias,
                       y, y_bs, Cin, Cout, H, W, OH, OW, KS, stride, pad, act, slope);
    return (int)hipGetLastError();
}

extern "C" int bfsr_grid_sample_add(const float* x, long long x_bs, const float* coord, const float* acc, long long acc_bs,
                                    float* out, long long out_bs, int B, int C, int h, int w, int qh, int qw, void* stream)
{
    if (!x || !coord || !acc || !out) return -1;
    const long long NQ = (long long)qh * qw;
    dim3 grid((unsigned)((NQ + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(grid_sample_add_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, coord, acc,
                       acc_bs, out, out_bs, C, h, w, NQ);
    return (int)hipGetLastError();
}
