// resample.hip -- resize (nearest / bilinear), 2x2 max-pool and affine-clamp kernels (gfx950).
// Index conventions follow the torch ops the reference calls:
//   nearest                 : src = min(floor(dst * r), in-1)                     (F.interpolate default)
//   bilinear, align False   : src = max(0, (dst+0.5)*r - 0.5), i1 = min(i0+1, in-1)
//   bilinear, align True    : src = dst * r with r = (in-1)/(out-1)
// and the bilinear blend is  hl0*(wl0*p00 + wl1*p01) + hl1*(wl0*p10 + wl1*p11).
#include <hip/hip_runtime.h>
#include "../../include/bfsr_hip.h"

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

}  // namespace

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
