// resample.hip -- resize (nearest / bilinear), 2x2 max-pool and affine-clamp kernels (gfx950).
// Index conventions follow the torch ops the reference calls:
//   nearest                 : src = min(floor(dst * r), in-1)                     (F.interpolate default)
//   bilinear, align False   : src = max(0, (dst+0.5)*r - 0.5), i1 = min(i0+1, in-1)
//   bilinear, align True    : src = dst * r with r = (in-1)/(out-1)
// and the bilinear blend is  hl0*(wl0*p00 + wl1*p01) + hl1*(wl0*p10 + wl1*p11).
#include <hip/hip_runtime.h>
#include "../../include/bfsr_hip.h"

namespace {

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
    const int ry = oy - oy0, rx = ox - ox0;
    float out = 0.f;
    if (ry >= 0 && ry < RH && rx >= 0 && rx < RW) {
        const float* xc = x + (long long)b * x_bs + (long long)c * IH * IW;
        if (mode == 0) {
            int sy = (int)floorf((float)ry * r_h); sy = sy < IH - 1 ? sy : IH - 1;
            int sx = (int)floorf((float)rx * r_w); sx = sx < IW - 1 ? sx : IW - 1;
            out = xc[(long long)sy * IW + sx];
        } else {
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
            out = hl0 * (wl0 * p00 + wl1 * p01) + hl1 * (wl0 * p10 + wl1 * p11);
        }
    }
    y[(long long)b * y_bs + i] = out;
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

}  // namespace

extern "C" int bfsr_resize(const float* x, long long x_bs, int IH, int IW, float* y, long long y_bs, int OH, int OW,
                           int RH, int RW, int oy0, int ox0, int B, int C, int mode, float r_h, float r_w, void* stream)
{
    if (!x || !y || mode < 0 || mode > 2 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return -1;
    const long long n = (long long)C * OH * OW;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)B);
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
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(axpb_clamp_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, r, r_bs,
                       y, y_bs, n, a, b, lo, hi);
    return (int)hipGetLastError();
}

extern "C" int bfsr_abi_version(void) { return BFSR_ABI_VERSION; }
