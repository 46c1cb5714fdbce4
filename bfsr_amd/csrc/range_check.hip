// range_check.hip -- per-channel dynamic-range check of an fp32 tensor that is about to enter a region computed with the two-term fp16 split
// (bfsr_amd/guard.py).  The pair x = hi + lo holds 22 significant bits only while lo = fp16(x - hi) is a normal number,
// |x| >= 2^-3, and degrades to an ABSOLUTE 2^-25 below that: harmless for the small elements of a channel whose large elements dominate, a real
// loss of relative precision for a channel that is tiny EVERYWHERE (a rescaled tensor, a nearly dead channel with compensating weights downstream).
// The split kernels themselves raise bit 0 of the flag for |x| >= 65504; this kernel raises
//   bit 3 when some channel's max |x| over the whole tensor is in (0, tiny)    (exact zeros are exact in any format)
//   bit 0 when some channel's max |x| is >= huge or not finite
// and guard.run_guarded re-runs the pass under the bf16x3 split (fp32's exponent range).  Two launches: per (channel, slice) maxima into a
// scratch array that is fully rewritten every call, then one block that reduces them and tests.
#include <hip/hip_runtime.h>
#include "../../include/bfsr_hip.h"

namespace {

constexpr int SLICES = 32;

__global__ __launch_bounds__(256) void channel_absmax_kernel(const float* __restrict__ x, long long x_bs, int B, long long HW, float* __restrict__ part, int vec4)
{
    const int c = blockIdx.x, s = blockIdx.y;
    const long long n = (long long)B * HW;
    float m = 0.f;
    bool bad = false;
    if (vec4) {                                                          // HW % 4 == 0 and 16-byte aligned planes: one float4 per lane and load
        const long long HW4 = HW >> 2, n4 = (long long)B * HW4;
        for (long long i = (long long)s * 256 + threadIdx.x; i < n4; i += (long long)SLICES * 256) {
            const long long b = i / HW4, p = i - b * HW4;
            const float4 q = reinterpret_cast<const float4*>(x + b * x_bs + (long long)c * HW)[p];
            const float v = fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), fabsf(q.w)));
            bad = bad || !(fabsf(q.x) <= 3.0e38f) || !(fabsf(q.y) <= 3.0e38f) || !(fabsf(q.z) <= 3.0e38f) || !(fabsf(q.w) <= 3.0e38f);
            m = fmaxf(m, v);
        }
    } else {
        for (long long i = (long long)s * 256 + threadIdx.x; i < n; i += (long long)SLICES * 256) {
            const long long b = i / HW, p = i - b * HW;
            const float v = fabsf(x[b * x_bs + (long long)c * HW + p]);
            bad = bad || !(v <= 3.0e38f);                                  // inf or NaN
            m = fmaxf(m, v);
        }
    }
    if (bad) m = __builtin_inff();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    __shared__ float w4[4];
    if ((threadIdx.x & 63) == 0) w4[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[c * SLICES + s] = fmaxf(fmaxf(w4[0], w4[1]), fmaxf(w4[2], w4[3]));
}

__global__ __launch_bounds__(256) void channel_range_test_kernel(const float* __restrict__ part, int C, float tiny, float huge, unsigned* flag)
{
    unsigned bits = 0;
    for (int c = threadIdx.x; c < C; c += 256) {
        float m = 0.f;
        for (int s = 0; s < SLICES; ++s) m = fmaxf(m, part[c * SLICES + s]);
        if (m > 0.f && m < tiny) bits |= 8u;
        if (!(m < huge)) bits |= 1u;
    }
    if (bits) atomicOr(flag, bits);
}

}  // namespace

extern "C" long long bfsr_channel_range_scratch(int C) { return C > 0 ? (long long)C * SLICES : -1; }

extern "C" int bfsr_channel_range_check(const float* x, long long x_bs, int B, int C, int H, int W, float tiny, float huge, float* scratch,
                                        unsigned* flag, void* stream)
{
    if (!x || !scratch || !flag || B <= 0 || C <= 0 || C > 65535 || H <= 0 || W <= 0 || !(tiny >= 0.f) || !(huge > tiny)) return -1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long HW = (long long)H * W;
    const int vec4 = (HW % 4 == 0) && (x_bs % 4 == 0) && ((reinterpret_cast<unsigned long long>(x) & 15) == 0);
    hipLaunchKernelGGL(channel_absmax_kernel, dim3((unsigned)C, SLICES), dim3(256), 0, st, x, x_bs, B, HW, scratch, vec4);
    hipLaunchKernelGGL(channel_range_test_kernel, dim3(1), dim3(256), 0, st, scratch, C, tiny, huge, flag);
    return (int)hipGetLastError();
}
