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


// ---------------------------------------------------------------------------------------------
// MFMA variant for wide steps (C = 96): the CxC invertible 1x1 conv is a [C x C] x [C x pixels] GEMM, done on
// v_mfma_f32_32x32x2_f32 with M = output channel, N = 32 pixels, K = input channel.  The B fragment (lane = pixel
// l&31 of channel 2kk + (l>>5)) is loaded straight from global memory -- one coalesced 128-byte row per half wave --
// and the elementwise stages that precede the matvec are applied to it in registers; W (k-major copy, `wt`) comes
// from L1/L2.  The stages that follow the matvec run on the accumulator layout (lane = pixel, 16 channels per tile).
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <int C>
__global__ __launch_bounds__(256) void flow_pointwise_mfma_kernel(BfsrFlowArgs a, long long HW)
{
    constexpr int CN = C / 2, MT = (C + 31) / 32;
    static_assert(CN % 2 == 0, "channel pairs must not straddle the z1/z2 split");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int b = blockIdx.y;
    const long long pix0 = ((long long)blockIdx.x * 4 + wave) * 32;      // 32 pixels per wave
    if (pix0 >= HW) return;
    const long long pix = pix0 + l31;
    const bool ok = pix < HW;
    const long long pc = ok ? pix : HW - 1;                                 // clamped: loads stay in range
    const float eps = a.eps;
    const float* zi = a.z_in + (long long)b * a.z_in_bs + pc;
    const float* ha = a.h_aff ? a.h_aff + (long long)b * a.h_aff_bs + pc : nullptr;
    const float* hf = a.h_ft ? a.h_ft + (long long)b * a.h_ft_bs + pc : nullptr;
    const float* __restrict__ wt = a.wt;                                    // [k][i]

    f32x16_t acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    // phase 1: all B fragments of this wave's 32 pixels (C/2 registers per lane) with the pre-matvec stages applied;
    // fully unrolled so every global load is in flight at once
    float xs[C / 2];
#pragma unroll
    for (int kk = 0; kk < C / 2; ++kk) xs[kk] = zi[(long long)(2 * kk + lhi) * HW];
    if (a.reverse) {
        if (ha) {
#pragma unroll
            for (int kk = CN / 2; kk < C / 2; ++kk) {       // channels >= CN (CN even): z2 = z2/scale - shift
                const int j = 2 * kk + lhi - CN;
                xs[kk] = xs[kk] / sigmoid_scale(ha[(long long)(2 * j + 1) * HW], eps) - ha[(long long)(2 * j) * HW];
            }
        }
        if (hf) {
#pragma unroll
            for (int kk = 0; kk < C / 2; ++kk) {
                const int c = 2 * kk + lhi;
                xs[kk] = xs[kk] / sigmoid_scale(hf[(long long)(2 * c + 1) * HW], eps) - hf[(long long)(2 * c) * HW];
            }
        }
    } else {
        if (ha) {
#pragma unroll
            for (int kk = CN / 2; kk < C / 2; ++kk) {
                const int j = 2 * kk + lhi - CN;
                xs[kk] = (xs[kk] + ha[(long long)(2 * j) * HW]) * sigmoid_scale(ha[(long long)(2 * j + 1) * HW], eps);
            }
        }
        if (a.an_bias) {
#pragma unroll
            for (int kk = 0; kk < C / 2; ++kk) {
                const int c = 2 * kk + lhi;
                xs[kk] = (xs[kk] + a.an_bias[c]) * a.an_escale[c];
            }
        }
    }
    // phase 2: the matvec on the matrix cores
#pragma unroll
    for (int kk = 0; kk < C / 2; ++kk) {
        const int c = 2 * kk + lhi;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int i = m * 32 + l31;
            const float w = i < C ? wt[c * C + i] : 0.f;
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, xs[kk], acc[m], 0, 0, 0);
        }
    }
    if (!ok) return;
    float* zo = a.z_out + (long long)b * a.z_out_bs + pix;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (i >= C) continue;
            float y = acc[m][r];
            if (a.reverse) {
                if (a.an_bias) y = y * a.an_escale[i] - a.an_bias[i];
            } else if (hf) {
                y = (y + hf[(long long)(2 * i) * HW]) * sigmoid_scale(hf[(long long)(2 * i + 1) * HW], eps);
            }
            zo[(long long)i * HW] = y;
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
    // 16-byte accesses only when the grid still has >= 2 blocks per CU; below that 8-byte accesses double the number of
    // waves in flight, which is what a ~35 us latency-bound launch needs (measured: C=24 @ 8x160x160 +10 %)
    if (v4 && (HW / 4 + 255) / 256 * a.B >= 512) return launch_flow<C, 4>(a, st);
    if (v4) return launch_flow<C, 2>(a, st);
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

// ---- likelihood reductions (per-sample sums into double accumulators) --------------------------------------------
__device__ __forceinline__ void block_atomic_add(double v, double* dst)
{
    __shared__ double part[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dst, part[0] + part[1] + part[2] + part[3]);
}

// h [B, 2*Cs, HW] cross split: scale raw = odd channels.  grid (blocks, B); a thread walks pixels with a grid stride.
__global__ __launch_bounds__(256) void logscale_sum_kernel(const float* __restrict__ h, long long h_bs, int Cs, long long HW,
                                                           float eps, double coef, double* __restrict__ out)
{
    const int b = blockIdx.y;
    const float* hb = h + (long long)b * h_bs;
    double acc = 0.0;
    for (int j = 0; j < Cs; ++j) {
        const float* hc = hb + (long long)(2 * j + 1) * HW;
        float s = 0.f;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long long)gridDim.x * 256)
            s += logf(1.f / (1.f + expf(-(hc[i] + 2.f))) + eps);
        acc += (double)s;
    }
    block_atomic_add(acc * coef, out + b);
}

__global__ __launch_bounds__(256) void gaussian_logp_kernel(const float* __restrict__ x, long long x_bs, const float* __restrict__ h,
                                                            long long h_bs, int C, long long HW, double coef, double* __restrict__ out)
{
    constexpr float LOG2PI = 1.8378770664093453f;
    const int b = blockIdx.y;
    const float* xb = x + (long long)b * x_bs;
    const float* hb = h ? h + (long long)b * h_bs : nullptr;
    double acc = 0.0;
    for (int c = 0; c < C; ++c) {
        float s = 0.f;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long long)gridDim.x * 256) {
            const float v = xb[(long long)c * HW + i];
            if (hb) {
                const float mean = hb[(long long)(2 * c) * HW + i], logs = hb[(long long)(2 * c + 1) * HW + i];
                const float d = v - mean;
                s += -0.5f * (logs * 2.f + d * d / expf(logs * 2.f) + LOG2PI);
            } else {
                s += -0.5f * (v * v + LOG2PI);
            }
        }
        acc += (double)s;
    }
    block_atomic_add(acc * coef, out + b);
}

}  // namespace

extern "C" int bfsr_logscale_sum(const float* h, long long h_bs, int B, int Cs, long long HW, float eps, double coef, double* out,
                                 void* stream)
{
    if (!h || !out || B <= 0 || Cs <= 0 || HW <= 0) return -1;
    long long blocks = (HW + 256 * 8 - 1) / (256 * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(logscale_sum_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       h, h_bs, Cs, HW, eps, coef, out);
    return (int)hipGetLastError();
}

extern "C" int bfsr_gaussian_logp(const float* x, long long x_bs, const float* h, long long h_bs, int B, int C, long long HW,
                                  double coef, double* out, void* stream)
{
    if (!x || !out || B <= 0 || C <= 0 || HW <= 0) return -1;
    long long blocks = (HW + 256 * 8 - 1) / (256 * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(gaussian_logp_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       x, x_bs, h, h_bs, C, HW, coef, out);
    return (int)hipGetLastError();
}


extern "C" int bfsr_flow_pointwise(const BfsrFlowArgs* a, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || !a->z_in || !a->z_out || a->B <= 0 || a->H <= 0 || a->W <= 0) return -1;
    if (a->an_bias && !a->an_escale) return -1;
    switch (a->C) {
        case 12: return dispatch_flow_vec<12>(*a, st, 4);
        case 24: return dispatch_flow_vec<24>(*a, st, 4);
        case 48: return dispatch_flow_vec<48>(*a, st, 1);
        case 96:
            if (a->w && a->wt) {       // MFMA path needs the k-major copy of W (in-place is fine: a wave reads all
                                       // channels of its 32 pixels before it stores any)
                const long long HW = (long long)a->H * a->W;
                dim3 grid((unsigned)((HW + 127) / 128), (unsigned)a->B);
                hipLaunchKernelGGL((flow_pointwise_mfma_kernel<96>), grid, dim3(256), 0, st, *a, HW);
                return (int)hipGetLastError();
            }
            return dispatch_flow_vec<96>(*a, st, 1);
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
