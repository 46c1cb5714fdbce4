// metrics.hip -- the step right after the hot path: evaluation metrics and output formatting on
// the device, so the fp32 SR images never travel to the host (LINF-LP/test.py:172-225, utils.py:132-193, imresize.py).
//   bfsr_resample_taps : one pass of a separable resampler with host-built tap tables (MATLAB-style antialiased bicubic
//                        `imresize`: imresize.py:64-88 builds the tables, :110-121 applies them)
//   bfsr_sqdiff_sum    : per-sample sum of squared differences over the shaved window, optional luma conversion
//                        (calc_psnr, utils.py:132-149); PSNR = -10*log10(sum/count) on the host
//   bfsr_ssim_sum      : per-(sample,channel) sum of the SSIM map, 11x11 Gaussian window sigma 1.5, 'valid' region,
//                        fp64 like the reference's cv2 path (utils.py:152-171)
//   bfsr_ssim_sum_w    : the same with a caller-given WS x WS window and covariance normalisation (WS = 7 uniform, NP/(NP-1): the
//                        skimage.metrics.structural_similarity call of SRFlow-LP/code/Measure.py:45-48)
//   bfsr_to_uint8      : round(clamp(x,0,1)*255) -> uint8 (round-half-even like numpy .round(), test.py:210-212)
#include <hip/hip_runtime.h>
#include "../../include/bfsr_hip.h"

namespace {

__device__ __forceinline__ void block_atomic_add_d(double v, double* dst)
{
    __shared__ double part[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dst, part[0] + part[1] + part[2] + part[3]);
}

// dim 0: y[b,c,o,x] = sum_p w[o,p] * x[b,c,idx[o,p],x];  dim 1: y[b,c,y,o] = sum_p w[o,p] * x[b,c,y,idx[o,p]]
__global__ __launch_bounds__(256) void resample_taps_kernel(const float* __restrict__ x, long long x_bs, float* __restrict__ y,
                                                            long long y_bs, const int* __restrict__ idx, const float* __restrict__ w,
                                                            int C, int H, int W, int O, int P, int dim)
{
    const int OH = dim == 0 ? O : H, OW = dim == 0 ? W : O;
    const long long n = (long long)C * OH * OW;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH), c = (int)(i / ((long long)OW * OH));
    const float* xc = x + (long long)b * x_bs + (long long)c * H * W;
    const int o = dim == 0 ? oy : ox;
    float s = 0.f;
    for (int p = 0; p < P; ++p) {
        const int j = idx[o * P + p];
        const float v = dim == 0 ? xc[(long long)j * W + ox] : xc[(long long)oy * W + j];
        s = fmaf(w[o * P + p], v, s);
    }
    y[(long long)b * y_bs + i] = s;
}

__global__ __launch_bounds__(256) void sqdiff_sum_kernel(const float* __restrict__ a, long long a_bs, const float* __restrict__ b_,
                                                         long long b_bs, int C, int H, int W, int shave, int luma, float inv_range,
                                                         double* __restrict__ out)
{
    const int b = blockIdx.y;
    const int VH = H - 2 * shave, VW = W - 2 * shave;
    const long long HW = (long long)H * W;
    const float* pa = a + (long long)b * a_bs;
    const float* pb = b_ + (long long)b * b_bs;
    const long long n = (long long)(luma ? 1 : C) * VH * VW;
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int vx = (int)(i % VW), vy = (int)((i / VW) % VH), c = (int)(i / ((long long)VW * VH));
        const long long o = (long long)(vy + shave) * W + vx + shave;
        float d;
        if (luma) {
            // diff.mul([65.738,129.057,25.064]/256).sum(dim=1)  (utils.py:137-140)
            d = (pa[o] - pb[o]) * inv_range * (65.738f / 256.f) + (pa[HW + o] - pb[HW + o]) * inv_range * (129.057f / 256.f) +
                (pa[2 * HW + o] - pb[2 * HW + o]) * inv_range * (25.064f / 256.f);
        } else {
            d = (pa[c * HW + o] - pb[c * HW + o]) * inv_range;
        }
        acc += (double)d * (double)d;
    }
    block_atomic_add_d(acc, out + b);
}

// one thread per valid pixel of one (sample, channel) plane; images are scaled by `scale` (255 for [0,1] inputs) first.  WS x WS window
// `win` (WS <= 11), 'valid' region (H-WS+1) x (W-WS+1); variances / covariance multiplied by cov_norm (1 for utils.calculate_ssim;
// NP/(NP-1) for skimage's sample covariance).
__global__ __launch_bounds__(256) void ssim_sum_kernel(const float* __restrict__ a, long long a_bs, const float* __restrict__ b_,
                                                       long long b_bs, int C, int H, int W, double scale, int WS, const double* __restrict__ win,
                                                       double cov_norm, double* __restrict__ out)
{
    const int bc = blockIdx.y, b = bc / C, c = bc % C;
    const int VH = H - WS + 1, VW = W - WS + 1;
    const float* pa = a + (long long)b * a_bs + (long long)c * H * W;
    const float* pb = b_ + (long long)b * b_bs + (long long)c * H * W;
    const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)VH * VW; i += (long long)gridDim.x * 256) {
        const int vx = (int)(i % VW), vy = (int)(i / VW);
        double m1 = 0, m2 = 0, s11 = 0, s22 = 0, s12 = 0;
        for (int dy = 0; dy < WS; ++dy)
            for (int dx = 0; dx < WS; ++dx) {
                const double wv = win[dy * WS + dx];
                const double p = (double)pa[(long long)(vy + dy) * W + vx + dx] * scale;
                const double q = (double)pb[(long long)(vy + dy) * W + vx + dx] * scale;
                m1 += wv * p; m2 += wv * q; s11 += wv * p * p; s22 += wv * q * q; s12 += wv * p * q;
            }
        const double v1 = cov_norm * (s11 - m1 * m1), v2 = cov_norm * (s22 - m2 * m2), cv = cov_norm * (s12 - m1 * m2);
        acc += ((2 * m1 * m2 + C1) * (2 * cv + C2)) / ((m1 * m1 + m2 * m2 + C1) * (v1 + v2 + C2));
    }
    block_atomic_add_d(acc, out + bc);
}

__global__ __launch_bounds__(256) void to_uint8_kernel(const float* __restrict__ x, long long x_bs, unsigned char* __restrict__ y,
                                                       long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const float v = fminf(fmaxf(x[(long long)b * x_bs + i], 0.f), 1.f) * 255.f;
    y[(long long)b * n + i] = (unsigned char)rintf(v);
}

}  // namespace

extern "C" int bfsr_resample_taps(const float* x, long long x_bs, float* y, long long y_bs, const int* idx, const float* w,
                                  int B, int C, int H, int W, int O, int P, int dim, void* stream)
{
    if (!x || !y || !idx || !w || B <= 0 || C <= 0 || H <= 0 || W <= 0 || O <= 0 || P <= 0 || (dim != 0 && dim != 1)) return -1;
    const long long n = (long long)C * (dim == 0 ? O : H) * (dim == 0 ? W : O);
    hipLaunchKernelGGL(resample_taps_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, x_bs, y, y_bs, idx, w, C, H, W, O, P, dim);
    return (int)hipGetLastError();
}

extern "C" int bfsr_sqdiff_sum(const float* a, long long a_bs, const float* b, long long b_bs, int B, int C, int H, int W, int shave,
                               int luma, float rgb_range, double* out, void* stream)
{
    if (!a || !b || !out || B <= 0 || C <= 0 || shave < 0 || H - 2 * shave <= 0 || W - 2 * shave <= 0 || (luma && C != 3)) return -1;
    long long blocks = ((long long)C * H * W + 256 * 8 - 1) / (256 * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sqdiff_sum_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       a, a_bs, b, b_bs, C, H, W, shave, luma, 1.f / rgb_range, out);
    return (int)hipGetLastError();
}

extern "C" int bfsr_ssim_sum_w(const float* a, long long a_bs, const float* b, long long b_bs, int B, int C, int H, int W, double scale,
                               int ws, const double* window, double cov_norm, double* out, void* stream)
{
    if (!a || !b || !out || !window || B <= 0 || C <= 0 || ws < 1 || ws > 11 || H < ws || W < ws) return -1;
    long long blocks = ((long long)(H - ws + 1) * (W - ws + 1) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ssim_sum_kernel, dim3((unsigned)blocks, (unsigned)(B * C)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       a, a_bs, b, b_bs, C, H, W, scale, ws, window, cov_norm, out);
    return (int)hipGetLastError();
}

extern "C" int bfsr_ssim_sum(const float* a, long long a_bs, const float* b, long long b_bs, int B, int C, int H, int W, double scale,
                             const double* window121, double* out, void* stream)
{
    if (H <= 10 || W <= 10) return -1;
    return bfsr_ssim_sum_w(a, a_bs, b, b_bs, B, C, H, W, scale, 11, window121, 1.0, out, stream);
}

extern "C" int bfsr_to_uint8(const float* x, long long x_bs, unsigned char* y, int B, long long n, void* stream)
{
    if (!x || !y || B <= 0 || n <= 0) return -1;
    hipLaunchKernelGGL(to_uint8_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, x_bs, y, n);
    return (int)hipGetLastError();
}
