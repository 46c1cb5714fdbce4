// range_check.hip -- per-SAMPLE, per-channel dynamic-range check of an fp32 tensor that is about to enter a region computed with the two-term
// fp16 split (bfsr_amd/guard.py).  The pair x = hi + lo holds 22 significant bits only while lo = fp16(x - hi) is a normal
// number, |x| >= 2^-3, and degrades to an ABSOLUTE 2^-25 below that.  The absolute error a tiny channel c adds to an output of the conv that reads
// it is at most 2^-25 * G_c, G_c = that conv's weight mass on input channel c; it matters only against the size of the conv's outputs, which the
// largest per-channel contribution max_c'(m_c' * G_c') (m = the channel's max |x| over the sample) estimates from below.  Round 6 rule, per sample:
//   bit 3 when some channel with 0 < m_c < tiny has  g_c * 2^-25 > 2^-20 * max_c'(m_c' * g_c'),  i.e.  max_c'(m_c' * g_c') < g_c * ratio
//         (ratio = 2^-5; g = G / max G in [0, 1], 1 for every channel when the caller passes no gains)
//   bit 0 when some channel's max |x| is >= huge or not finite
// Without gains the rule fires when the WHOLE sample is tiny (an exactly rescaled trunk); with them also for a channel that is tiny but read by
// weights far above the rest (the hostile "compensating weights" case).  A dark crop, a nearly dead channel or a small trained coefficient next to
// normal channels read by comparable weights does NOT fire: its absolute error is below 2^-20 of the outputs it feeds (round 5 flagged every
// channel with max |x| < 2^-7 over the whole BATCH, which made a sample's bits depend on its batch-mates and re-ran realistic inputs at 2x cost:
// ADVICE round 5).  Exact zeros are exact in any format.  guard.run_guarded re-runs a flagged pass under the bf16x3 split (fp32's exponent range).
// Two launches: per (sample, channel, slice) maxima into a scratch array that is fully rewritten every call, then one block per sample that
// reduces them and tests.
#include <hip/hip_runtime.h>
#include "../../include/bfsr_hip.h"

namespace {

constexpr int TARGET_SLICES = 32;                                        // blocks per channel the first launch aims for (B * slices-per-sample)

__host__ __device__ inline int slices_per_sample(int B) { return B >= TARGET_SLICES ? 1 : (TARGET_SLICES + B - 1) / B; }

__global__ __launch_bounds__(256) void channel_absmax_kernel(const float* __restrict__ x, long long x_bs, int C, long long HW, int S, float* __restrict__ part, int vec4)
{
    const int c = blockIdx.x, b = blockIdx.y / S, s = blockIdx.y - b * S;
    const float* xp = x + (long long)b * x_bs + (long long)c * HW;
    float m = 0.f;
    bool bad = false;
    if (vec4) {                                                          // HW % 4 == 0 and 16-byte aligned planes: one float4 per lane and load
        const long long HW4 = HW >> 2;
        for (long long i = (long long)s * 256 + threadIdx.x; i < HW4; i += (long long)S * 256) {
            const float4 q = reinterpret_cast<const float4*>(xp)[i];
            const float v = fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), fabsf(q.w)));
            bad = bad || !(fabsf(q.x) <= 3.0e38f) || !(fabsf(q.y) <= 3.0e38f) || !(fabsf(q.z) <= 3.0e38f) || !(fabsf(q.w) <= 3.0e38f);
            m = fmaxf(m, v);
        }
    } else {
        for (long long i = (long long)s * 256 + threadIdx.x; i < HW; i += (long long)S * 256) {
            const float v = fabsf(xp[i]);
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
    if (threadIdx.x == 0) part[((long long)b * C + c) * S + s] = fmaxf(fmaxf(w4[0], w4[1]), fmaxf(w4[2], w4[3]));
}

// one block per sample
__global__ __launch_bounds__(256) void channel_range_test_kernel(const float* __restrict__ part, int C, int S, float tiny, float huge, float ratio,
                                                                 const float* __restrict__ gain, unsigned* flag)
{
    const float* pb = part + (long long)blockIdx.x * C * S;
    auto cmax = [&](int c) {
        float m = 0.f;
        for (int s = 0; s < S; ++s) m = fmaxf(m, pb[c * S + s]);
        return m;
    };
    // reference size of what the consumer computes: the largest per-channel contribution m_c * g_c of this sample
    float ref = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float m = cmax(c);
        if (m <= 3.0e38f) ref = fmaxf(ref, m * (gain ? gain[c] : 1.f));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ref = fmaxf(ref, __shfl_down(ref, o, 64));
    __shared__ float w4[4];
    if ((threadIdx.x & 63) == 0) w4[threadIdx.x >> 6] = ref;
    __syncthreads();
    ref = fmaxf(fmaxf(w4[0], w4[1]), fmaxf(w4[2], w4[3]));
    unsigned bits = 0;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float m = cmax(c);
        const float g = gain ? gain[c] : 1.f;
        if (m > 0.f && m < tiny && ref < g * ratio) bits |= 8u;
        if (!(m < huge)) bits |= 1u;
    }
    if (bits) atomicOr(flag, bits);
}

}  // namespace

extern "C" long long bfsr_channel_range_scratch(int B, int C) { return (B > 0 && C > 0) ? (long long)B * C * slices_per_sample(B) : -1; }

extern "C" int bfsr_channel_range_check(const float* x, long long x_bs, int B, int C, int H, int W, float tiny, float huge, float ratio,
                                        const float* gain, float* scratch, unsigned* flag, void* stream)
{
    if (!x || !scratch || !flag || B <= 0 || C <= 0 || C > 65535 || H <= 0 || W <= 0 || !(tiny >= 0.f) || !(huge > tiny) || !(ratio >= 0.f)) return -1;
    const int S = slices_per_sample(B);
    if ((long long)B * S > 65535) return -1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long HW = (long long)H * W;
    const int vec4 = (HW % 4 == 0) && (x_bs % 4 == 0) && ((reinterpret_cast<unsigned long long>(x) & 15) == 0);
    hipLaunchKernelGGL(channel_absmax_kernel, dim3((unsigned)C, (unsigned)(B * S)), dim3(256), 0, st, x, x_bs, C, HW, S, scratch, vec4);
    hipLaunchKernelGGL(channel_range_test_kernel, dim3((unsigned)B), dim3(256), 0, st, scratch, C, S, tiny, huge, ratio, gain, flag);
    return (int)hipGetLastError();
}
