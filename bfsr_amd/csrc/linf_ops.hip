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
// FAST (round 6): the sigmoid of the 2*D*L conditional scales per query point as v_exp_f32 of x * log2(e) and quotients as v_rcp_f32 + one Newton
// step on the quotient (<= 1-2 ulp; coupling_tail_kernel's choices) instead of libm expf + two IEEE divisions -- this kernel is VALU-bound (one query
// point per lane: 270 sigmoids next to 8 019 FMAs) and those were 60 % of its instructions.  Used by the inverse and by the forward pass when no
// log-density is asked for; the log_p form (its logf included) and the vector-Jacobian product keep the libm forms.
template <int D, int FAST>
__global__ __launch_bounds__(256) void linf_flow_kernel(BfsrLinfFlowArgs a)
{
    auto fdiv = [](float n, float d) {
        const float r = __builtin_amdgcn_rcpf(d);
        const float qt = n * r;
        return fmaf(fmaf(-d, qt, n), r, qt);
    };
    const long long NQ = (long long)a.qh * a.qw;
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y, L = a.layers;
    // the (L+1) linears' matrices and biases in LDS, read back as broadcasts (round 6).  As wave-uniform global loads they went through the scalar
    // cache: 32 KB of matrices per pass against a 16 KB cache, ~100 SGPRs to land them in -- every ~64 FMAs waited for a scalar load from L2 and
    // the kernel ran at 1.9 TB/s of its 18 GB, neither VALU- nor HBM-bound (cfg5: 9.3 ms per direction)
    extern __shared__ float sLin[];
    {
        const int nw = (L + 1) * D * D, nb = (L + 1) * D;
        for (int i = threadIdx.x; i < nw; i += 256) sLin[i] = a.lin_w[i];
        for (int i = threadIdx.x; i < nb; i += 256) sLin[nw + i] = a.lin_b[i];
        __syncthreads();
    }
    if (q >= NQ) return;
    // an opaque per-lane zero in the LDS addresses: hipcc otherwise proves the loaded values wave-uniform and moves every one of them into an SGPR
    // (v_readfirstlane per weight + 151 spilled SGPRs); as "divergent" values they stay in the VGPRs the broadcast reads deliver them in
    int lz = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(lz));
#endif
    const float* sMat = sLin + lz;
    const float* sBias = sMat + (L + 1) * D * D;
    const float* xi = a.x + (long long)b * a.x_bs + q;
    const float* ai = a.ai + (long long)b * a.ai_bs + q;
    float x[D], y[D];
    // conditioning values of one flow layer: lv[d] = raw scale, lv[D + d] = shift.  ai_fmt 1 = the quad-major layout the fused MLP
    // kernel writes ([layers][QB quads][NQ][4], a layer's 2*D values padded to QB*4): QB 16-byte loads per layer instead of 2*D
    // four-byte ones (the row-major form streamed at 2.4 TB/s, load-issue bound: 540 loads per query point)
    constexpr int SH = (D + 3) / 4 * 4, QB = 2 * SH / 4;                  // ai_fmt 1: a layer = [SH raw scales (D used) | SH shifts (D used)]
    float lv[QB * 4];
    const int so = a.ai_fmt == 1 ? SH : D;                               // where the shifts start inside lv
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
        const float* __restrict__ W = sMat + layer * D * D;
        const float* __restrict__ bb = sBias + layer * D;
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
                const float sr = lv[d], sh = lv[so + d];
                if constexpr (FAST) {
                    x[d] = x[d] * (fdiv(1.f, 1.f + __expf(-(sr + 2.f))) + a.eps) + sh;
                    continue;
                }
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
            const float* __restrict__ W = sMat + layer * D * D;
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
                const float sr = lv[d], sh = lv[so + d];
                if constexpr (FAST) {
                    const float t = 1.f + __expf(-(sr + 2.f));
                    x[d] = (x[d] - sh) * fdiv(t, fmaf(a.eps, t, 1.f));       // 1 / (1/t + eps) = t / (1 + eps t)
                    continue;
                }
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
// The same flow for D = 27 on the fp32 matrix pipe (round 6): the one-query-per-lane form above spends its time feeding 8 019 FMAs per query
// with wave-uniform weights (scalar loads through a 16 KB cache for 32 KB of matrices; as LDS broadcasts the return path moves 64 copies of
// every weight) and ran at 1.9 TB/s of its 18 GB at BASELINE config 5.  Here a wave owns 64 query points as two 32-column tiles of
// v_mfma_f32_32x32x2_f32: M = output row (27 of 32), N = query point, K = input row.  The accumulator layout of one linear (lane (l31, half):
// register r = row 8(r>>2) + (r&3) + 4*half of query l31) IS the B operand of the next if that linear's K axis is contracted in the order
// k(r, half) -- the chain never leaves that layout (coupling_head_kernel's trick); in it a lane owns whole (scale | shift) QUADS of the conditioning
// ([layer][14 quads][NQ][4]: raw scales of rows 4j..4j+3 in quad j, shifts in quad 7 + j), so the affine stage is 16-byte loads and no exchange.
// A operands ([layer][r][64 lanes], zero beyond 27) and biases live in LDS (46 KB), filled once per block; blocks walk query chunks grid-stride.
// Per layer and wave: 32 MFMAs (2 048 matrix-pipe cycles) instead of 729 x 4 VALU cycles, 16 conflict-free ds_read_b32, 16 x 16-byte loads.
// Sigmoid / quotient: the FAST forms of linf_flow_kernel.  REV 0: forward without log-density, REV 1: inverse.
typedef float flow_f32x16 __attribute__((ext_vector_type(16)));
template <int REV>
__global__ __launch_bounds__(256) void linf_flow27_mfma_kernel(BfsrLinfFlowArgs a, int nchunks)
{
    constexpr int D = 27, QB = 14;
    extern __shared__ float sLin[];
    const int L = a.layers, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int nA = (L + 1) * 1024;
    for (int i = tid; i < nA; i += 256) {
        const int ln = i & 63, r = (i >> 6) & 15, layer = i >> 10;
        const int row = ln & 31, k = 8 * (r >> 2) + (r & 3) + 4 * (ln >> 5);
        sLin[i] = (row < D && k < D) ? a.lin_w[(layer * D + row) * D + k] : 0.f;
    }
    for (int i = tid; i < (L + 1) * 32; i += 256) {
        const int row = i & 31, layer = i >> 5;
        sLin[nA + i] = row < D ? a.lin_b[layer * D + row] : 0.f;
    }
    __syncthreads();
    const float* sB = sLin + nA;
    auto fdiv = [](float n, float d) {
        const float r = __builtin_amdgcn_rcpf(d);
        const float qt = n * r;
        return fmaf(fmaf(-d, qt, n), r, qt);
    };
    const long long NQ = (long long)a.qh * a.qw;
    const int b = blockIdx.y;
    const float* xb = a.x + (long long)b * a.x_bs;
    float* yb = a.y + (long long)b * a.y_bs;
    const float4* aq = reinterpret_cast<const float4*>(a.ai + (long long)b * a.ai_bs);
    const float eps = a.eps;
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const long long q0 = (long long)chunk * 256 + wave * 64;
        if (q0 >= NQ) continue;                                           // (wave-uniform)
        long long qc[2];
        bool ok[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) { const long long qn = q0 + 32 * t + l31; ok[t] = qn < NQ; qc[t] = ok[t] ? qn : NQ - 1; }
        float xv[2][16];                                                  // [tile][r]: row 8(r>>2) + (r&3) + 4*lhi of query qc[tile]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 8 * (r >> 2) + (r & 3) + 4 * lhi;
                xv[t][r] = row < D ? xb[(long long)row * NQ + qc[t]] : 0.f;
            }
        auto matvec = [&](int layer, bool sub_bias_first) {               // x = W x + b  |  x = W (x - b)
            const float* A = sLin + layer * 1024 + lane;
            const float* bb = sB + layer * 32 + 4 * lhi;
            flow_f32x16 acc[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bias = bb[8 * (r >> 2) + (r & 3)];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (sub_bias_first) { xv[t][r] -= bias; acc[t][r] = 0.f; }
                    else acc[t][r] = bias;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ar = A[r * 64];
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, xv[t][r], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) xv[t][r] = acc[t][r];
        };
        auto affine = [&](int layer) {                                    // forward: x = x * scale + shift; inverse: x = (x - shift) / scale
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float4 sc4[4], sh4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bool pad = g == 3 && lhi;                       // rows 28..31 do not exist (their x stays 0: zero weights, zero shift)
                    const int qd = pad ? 6 : 2 * g + lhi;
                    sc4[g] = aq[(long long)(layer * QB + qd) * NQ + qc[t]];
                    sh4[g] = aq[(long long)(layer * QB + 7 + qd) * NQ + qc[t]];
                    if (pad) { sc4[g] = make_float4(0.f, 0.f, 0.f, 0.f); sh4[g] = make_float4(0.f, 0.f, 0.f, 0.f); }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float sr[4] = {sc4[g].x, sc4[g].y, sc4[g].z, sc4[g].w}, sh[4] = {sh4[g].x, sh4[g].y, sh4[g].z, sh4[g].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float tt = 1.f + __expf(-(sr[e] + 2.f));
                        float& v = xv[t][4 * g + e];
                        if (REV) v = (v - sh[e]) * fdiv(tt, fmaf(eps, tt, 1.f));      // 1 / (1/t + eps) = t / (1 + eps t)
                        else v = v * (fdiv(1.f, tt) + eps) + sh[e];
                    }
                }
            }
            // row 27 (r = 15 of the lower half-wave) is padding too: keep it exactly 0 whatever the producer left in the 28th slot
            if (!lhi) { xv[0][15] = 0.f; xv[1][15] = 0.f; }
        };
        if (!REV) {
            for (int i = 0; i < L; ++i) { matvec(i, false); affine(i); }
            matvec(L, false);
        } else {
            matvec(L, true);
            for (int i = L - 1; i >= 0; --i) { affine(i); matvec(i, true); }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 8 * (r >> 2) + (r & 3) + 4 * lhi;
                if (ok[t] && row < D) yb[(long long)row * NQ + qc[t]] = xv[t][r];
            }
    }
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
    const bool fast = (a->reverse == 1) || (a->reverse == 0 && !a->log_p);
    if (a->layers < 0 || a->layers > 15) return -1;
    if (fast && a->D == 27 && a->ai_fmt == 1 && a->layers <= 13 && a->x != a->y) {                 // the matrix-pipe form (46 KB of LDS at 10 layers)
        const long long nchunks = (NQ + 255) / 256;
        if (nchunks > 0x7fffffffLL) return -1;
        long long gx = 1536 / (a->B > 0 ? a->B : 1);
        gx = gx < 1 ? 1 : (gx > nchunks ? nchunks : gx);
        const unsigned ldsm = (unsigned)((a->layers + 1) * (1024 + 32) * 4);
        dim3 gm((unsigned)gx, (unsigned)a->B);
        if (a->reverse) hipLaunchKernelGGL((linf_flow27_mfma_kernel<1>), gm, dim3(256), ldsm, st, *a, (int)nchunks);
        else hipLaunchKernelGGL((linf_flow27_mfma_kernel<0>), gm, dim3(256), ldsm, st, *a, (int)nchunks);
        return (int)hipGetLastError();
    }
    const unsigned lds = (unsigned)((a->layers + 1) * a->D * (a->D + 1) * 4);       // <= 48 KB (D = 27, 16 linears): under the default dynamic-LDS limit
#define BFSR_FLOW_(D_) do { if (fast) hipLaunchKernelGGL((linf_flow_kernel<D_, 1>), grid, dim3(256), lds, st, *a); \
                             else hipLaunchKernelGGL((linf_flow_kernel<D_, 0>), grid, dim3(256), lds, st, *a); } while (0)
    if (a->D == 27) BFSR_FLOW_(27);
    else if (a->D == 3) BFSR_FLOW_(3);
    else if (a->D == 12) BFSR_FLOW_(12);
    else return -1;
#undef BFSR_FLOW_
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
