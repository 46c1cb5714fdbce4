// flow_ops.hip -- HBM-bound pointwise kernels of the SRFlow flow stack (gfx950, wave64).
//   bfsr_flow_pointwise : the fused FlowStep chain (self-conditional affine on z2, feature-conditional
//                         affine on z, CxC invertible 1x1 conv, actnorm) -- one read of z/h_aff/h_ft and
//                         one write of z per step (20*C bytes per pixel for a coupled step)
//   bfsr_squeeze2d / bfsr_unsqueeze2d, bfsr_split2d, bfsr_standardize
// Layout: NCHW fp32 views; lanes run along the contiguous H*W plane so every per-channel access of a
// wave is one coalesced 256 B (VEC=1) .. 1 KiB (VEC=4) segment.
#include <hip/hip_runtime.h>
#include "../../include/bfsr_hip.h"

namespace {

__device__ __forceinline__ float sigmoid_scale(float raw, float eps)
{
    // torch.sigmoid(raw + 2.) + eps   (FlowAffineCouplingsAblation.py:111,118)
    return 1.f / (1.f + expf(-(raw + 2.f))) + eps;
}

template <int VEC> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<2> { typedef float2 T; };
template <> struct VecT<4> { typedef float4 T; };

template <int VEC>
__device__ __forceinline__ void ldv(const float* p, float (&v)[VEC])
{
    typedef typename VecT<VEC>::T T;
    T t = *reinterpret_cast<const T*>(p);
    const float* f = reinterpret_cast<const float*>(&t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = f[i];
}
template <int VEC>
__device__ __forceinline__ void stv(float* p, const float (&v)[VEC])
{
    typedef typename VecT<VEC>::T T;
    T t;
    float* f = reinterpret_cast<float*>(&t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = v[i];
    *reinterpret_cast<T*>(p) = t;
}

// C compile-time so the per-pixel channel vector lives in VGPRs.
template <int C, int VEC>
__global__ __launch_bounds__(256) void flow_pointwise_kernel(BfsrFlowArgs a, long long HW, long long nvec)
{
    constexpr int CN = C / 2;        // channels_for_nn (z1)
    constexpr int CC = C - CN;       // channels_for_co (z2)
    const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
    if (v >= nvec) return;
    const int b = blockIdx.y;
    const long long pix = v * VEC;
    const float eps = a.eps;

    float x[C][VEC];
    {
        const float* zi = a.z_in + (long long)b * a.z_in_bs + pix;
#pragma unroll
        for (int c = 0; c < C; ++c) ldv<VEC>(zi + c * HW, x[c]);
    }
    const float* ha = a.h_aff ? a.h_aff + (long long)b * a.h_aff_bs + pix : nullptr;
    const float* hf = a.h_ft ? a.h_ft + (long long)b * a.h_ft_bs + pix : nullptr;

    if (a.reverse) {
        if (ha) {
#pragma unroll
            for (int j = 0; j < CC; ++j) {
                float sh[VEC], sr[VEC];
                ldv<VEC>(ha + (2 * j) * HW, sh);
                ldv<VEC>(ha + (2 * j + 1) * HW, sr);
#pragma unroll
                for (int i = 0; i < VEC; ++i) x[CN + j][i] = x[CN + j][i] / sigmoid_scale(sr[i], eps) - sh[i];
            }
        }
        if (hf) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float sh[VEC], sr[VEC];
                ldv<VEC>(hf + (2 * c) * HW, sh);
                ldv<VEC>(hf + (2 * c + 1) * HW, sr);
#pragma unroll
                for (int i = 0; i < VEC; ++i) x[c][i] = x[c][i] / sigmoid_scale(sr[i], eps) - sh[i];
            }
        }
    } else {
        if (ha) {
#pragma unroll
            for (int j = 0; j < CC; ++j) {
                float sh[VEC], sr[VEC];
                ldv<VEC>(ha + (2 * j) * HW, sh);
                ldv<VEC>(ha + (2 * j + 1) * HW, sr);
#pragma unroll
                for (int i = 0; i < VEC; ++i) x[CN + j][i] = (x[CN + j][i] + sh[i]) * sigmoid_scale(sr[i], eps);
            }
        }
        if (a.an_bias) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float bb = a.an_bias[c], es = a.an_escale[c];
#pragma unroll
                for (int i = 0; i < VEC; ++i) x[c][i] = (x[c][i] + bb) * es;
            }
        }
    }

    float* zo = a.z_out + (long long)b * a.z_out_bs + pix;
    if (a.w) {
        // y = W x : weights are wave-uniform (scalar loads), rows processed one at a time
        const float* __restrict__ w = a.w;
#pragma unroll 4
        for (int i = 0; i < C; ++i) {
            float y[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) y[k] = 0.f;
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const float wij = w[i * C + j];
#pragma unroll
                for (int k = 0; k < VEC; ++k) y[k] = fmaf(wij, x[j][k], y[k]);
            }
            if (a.reverse) {
                if (a.an_bias) {
                    const float bb = a.an_bias[i], es = a.an_escale[i];
#pragma unroll
                    for (int k = 0; k < VEC; ++k) y[k] = y[k] * es - bb;
                }
            } else if (hf) {
                float sh[VEC], sr[VEC];
                ldv<VEC>(hf + (2 * i) * HW, sh);
                ldv<VEC>(hf + (2 * i + 1) * HW, sr);
#pragma unroll
                for (int k = 0; k < VEC; ++k) y[k] = (y[k] + sh[k]) * sigmoid_scale(sr[k], eps);
            }
            stv<VEC>(zo + i * HW, y);
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (a.reverse) {
                if (a.an_bias) {
                    const float bb = a.an_bias[c], es = a.an_escale[c];
#pragma unroll
                    for (int k = 0; k < VEC; ++k) x[c][k] = x[c][k] * es - bb;
                }
            } else if (hf) {
                float sh[VEC], sr[VEC];
                ldv<VEC>(hf + (2 * c) * HW, sh);
                ldv<VEC>(hf + (2 * c + 1) * HW, sr);
#pragma unroll
                for (int k = 0; k < VEC; ++k) x[c][k] = (x[c][k] + sh[k]) * sigmoid_scale(sr[k], eps);
            }
            stv<VEC>(zo + c * HW, x[c]);
        }
    }
}

template <int C, int VEC>
int launch_flow(const BfsrFlowArgs& a, hipStream_t st)
{
    const long long HW = (long long)a.H * a.W;
    const long long nvec = HW / VEC;
    dim3 grid((unsigned)((nvec + 255) / 256), (unsigned)a.B);
    hipLaunchKernelGGL((flow_pointwise_kernel<C, VEC>), grid, dim3(256), 0, st, a, HW, nvec);
    return (int)hipGetLastError();
}

inline bool aligned16(const void* p) { return (reinterpret_cast<unsigned long long>(p) & 15ull) == 0; }

template <int C>
int dispatch_flow_vec(const BfsrFlowArgs& a, hipStream_t st, int maxvec)
{
    const long long HW = (long long)a.H * a.W;
    bool v4 = maxvec >= 4 && (HW % 4 == 0) && aligned16(a.z_in) && aligned16(a.z_out) && (a.z_in_bs % 4 == 0) &&
              (a.z_out_bs % 4 == 0);
    if (a.h_aff) v4 = v4 && aligned16(a.h_aff) && (a.h_aff_bs % 4 == 0);
    if (a.h_ft) v4 = v4 && aligned16(a.h_ft) && (a.h_ft_bs % 4 == 0);
    if (v4) return launch_flow<C, 4>(a, st);
    return launch_flow<C, 1>(a, st);
}

// ---------------------------------------------------------------------------------------------
__global__ void squeeze2d_kernel(const float* __restrict__ x, long long x_bs, float* __restrict__ y, long long y_bs,
                                 int C, int H, int W)
{
    // one thread per (c, input row r, output column w): reads x[c][r][2w..2w+1] (8 B), writes 2 planes
    const int Wo = W >> 1, Ho = H >> 1;
    const long long n = (long long)C * H * Wo;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const int w = (int)(i % Wo);
    const int r = (int)((i / Wo) % H);
    const int c = (int)(i / ((long long)Wo * H));
    const float2 v = *reinterpret_cast<const float2*>(x + (long long)b * x_bs + ((long long)c * H + r) * W + 2 * w);
    const int fy = r & 1, h = r >> 1;
    float* yo = y + (long long)b * y_bs + ((long long)(c * 4 + fy * 2) * Ho + h) * Wo + w;
    yo[0] = v.x;
    yo[(long long)Ho * Wo] = v.y;
}

__global__ void unsqueeze2d_kernel(const float* __restrict__ x, long long x_bs, float* __restrict__ y, long long y_bs,
                                   int Co, int H, int W)
{
    // x [4*Co][H][W] -> y [Co][2H][2W]; one thread per (co, output row R, input column w)
    const int Ho = 2 * H, Wo = 2 * W;
    const long long n = (long long)Co * Ho * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const int w = (int)(i % W);
    const int R = (int)((i / W) % Ho);
    const int co = (int)(i / ((long long)W * Ho));
    const int fy = R & 1, h = R >> 1;
    const float* xi = x + (long long)b * x_bs + ((long long)(co * 4 + fy * 2) * H + h) * W + w;
    float2 v;
    v.x = xi[0];
    v.y = xi[(long long)H * W];
    *reinterpret_cast<float2*>(y + (long long)b * y_bs + ((long long)co * Ho + R) * Wo + 2 * w) = v;
}

__global__ void split2d_kernel(const float* __restrict__ h, long long h_bs, const float* __restrict__ src,
                               long long src_bs, float* __restrict__ dst, long long dst_bs, int Cc, long long HW,
                               int reverse)
{
    const long long n = (long long)Cc * HW;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const int j = (int)(i / HW);
    const long long pix = i - (long long)j * HW;
    const float mean = h[(long long)b * h_bs + (long long)(2 * j) * HW + pix];
    const float logs = h[(long long)b * h_bs + (long long)(2 * j + 1) * HW + pix];
    const float s = src[(long long)b * src_bs + i];
    const float e = expf(logs);
    dst[(long long)b * dst_bs + i] = reverse ? (mean + e * s) : ((s - mean) / e);
}

__global__ void standardize_kernel(const float* __restrict__ x, long long x_bs, float* __restrict__ y, long long y_bs,
                                   int C, long long HW)
{
    const long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= HW) return;
    const int b = blockIdx.y;
    const float* xi = x + (long long)b * x_bs + pix;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += xi[(long long)c * HW];
    const float mean = s / (float)C;
    float q = 0.f;
    for (int c = 0; c < C; ++c) {
        const float d = xi[(long long)c * HW] - mean;
        q = fmaf(d, d, q);
    }
    const float sd = sqrtf(q / (float)(C - 1)) + 1e-8f;
    float* yo = y + (long long)b * y_bs + pix;
    for (int c = 0; c < C; ++c) yo[(long long)c * HW] = (xi[(long long)c * HW] - mean) / sd;
}

}  // namespace

extern "C" int bfsr_flow_pointwise(const BfsrFlowArgs* a, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || !a->z_in || !a->z_out || a->B <= 0 || a->H <= 0 || a->W <= 0) return -1;
    if (a->an_bias && !a->an_escale) return -1;
    switch (a->C) {
        case 12: return dispatch_flow_vec<12>(*a, st, 4);
        case 24: return dispatch_flow_vec<24>(*a, st, 4);
        case 48: return dispatch_flow_vec<48>(*a, st, 1);
        case 96: return dispatch_flow_vec<96>(*a, st, 1);
        case 3: return dispatch_flow_vec<3>(*a, st, 4);
        case 6: return dispatch_flow_vec<6>(*a, st, 4);
        default: return -1;
    }
}

extern "C" int bfsr_squeeze2d(const float* x, long long x_bs, float* y, long long y_bs, int B, int C, int H, int W,
                              void* stream)
{
    if (!x || !y || (H & 1) || (W & 1)) return -1;
    const long long n = (long long)C * H * (W / 2);
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(squeeze2d_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, y, y_bs, C,
                       H, W);
    return (int)hipGetLastError();
}

extern "C" int bfsr_unsqueeze2d(const float* x, long long x_bs, float* y, long long y_bs, int B, int C, int H, int W,
                                void* stream)
{
    if (!x || !y || (C & 3)) return -1;
    const int Co = C / 4;
    const long long n = (long long)Co * (2 * H) * W;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(unsqueeze2d_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, y, y_bs,
                       Co, H, W);
    return (int)hipGetLastError();
}

extern "C" int bfsr_split2d(const float* h, long long h_bs, const float* src, long long src_bs, float* dst,
                            long long dst_bs, int B, int Cc, int H, int W, int reverse, void* stream)
{
    if (!h || !src || !dst) return -1;
    const long long HW = (long long)H * W, n = HW * Cc;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(split2d_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), h, h_bs, src, src_bs,
                       dst, dst_bs, Cc, HW, reverse);
    return (int)hipGetLastError();
}

extern "C" int bfsr_standardize(const float* x, long long x_bs, float* y, long long y_bs, int B, int C, int H, int W,
                                void* stream)
{
    if (!x || !y || C < 2) return -1;
    const long long HW = (long long)H * W;
    dim3 grid((unsigned)((HW + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(standardize_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, y, y_bs,
                       C, HW);
    return (int)hipGetLastError();
}
