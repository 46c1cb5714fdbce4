// coexec_probe.hip -- tiny victim kernels (one instruction class each) for tools/exp/coexec_probe.py: which part of resize4_kernel goes wrong when its
// waves share a SIMD with the MFMA chain of the 1x1-only coupling_head?  Built stand-alone: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/exp/libcoexec.so
#include <hip/hip_runtime.h>

extern "C" {

// integer division / modulo by a run-time divisor (hipcc: v_rcp_iflag_f32 + corrections)
__global__ void k_udiv(unsigned* out, unsigned n, unsigned d)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[2 * i] = i / d;
    out[2 * i + 1] = i % d;
}
// float conversions / floor / select arithmetic of the bilinear index computation (no memory gathers)
__global__ void k_index(float* out, unsigned n, float r)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int rx = (int)(i & 1023u);
    const float fx = (float)rx * r;
    int x0 = (int)fx; x0 = x0 < 500 ? x0 : 500;
    const int x1 = x0 + (x0 < 500 ? 1 : 0);
    const float w1 = fx - (float)x0, w0 = 1.f - w1;
    out[i] = w0 * (float)x0 + w1 * (float)x1 + floorf(fx * 0.37f);
}
// four data-dependent gathers + blend with index arithmetic by shifts only (no division, no float->int)
__global__ void k_gather(const float* __restrict__ x, float* out, unsigned n, unsigned mask)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned a = (i * 2654435761u) & mask, b = (a + 1) & mask, c = (a + 97) & mask, d = (a + 98) & mask;
    out[i] = 0.25f * (x[a] + x[b]) + 0.5f * (x[c] - x[d]);
}
// IEEE float division (v_div_scale / v_rcp_f32 / v_div_fmas / v_div_fixup)
__global__ void k_fdiv(const float* __restrict__ x, float* out, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = (x[i] + 3.f) / (fabsf(x[(i + 1) % n]) + 0.5f);
}

static int launch(void (*k)(), void* s) { (void)k; (void)s; return 0; }
int run_udiv(unsigned* out, unsigned n, unsigned d, void* s) { hipLaunchKernelGGL(k_udiv, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, out, n, d); return (int)hipGetLastError(); }
int run_index(float* out, unsigned n, float r, void* s) { hipLaunchKernelGGL(k_index, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, out, n, r); return (int)hipGetLastError(); }
int run_gather(const float* x, float* out, unsigned n, unsigned mask, void* s) { hipLaunchKernelGGL(k_gather, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, x, out, n, mask); return (int)hipGetLastError(); }
int run_fdiv(const float* x, float* out, unsigned n, void* s) { hipLaunchKernelGGL(k_fdiv, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, x, out, n); return (int)hipGetLastError(); }
}
