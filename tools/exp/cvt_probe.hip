// cvt_probe.hip -- do v_cvt_f16_f32 and v_cvt_pk_f16_f32 return the same fp16 for the same fp32 on gfx950?  (hipcc uses them interchangeably: DESIGN.md section 5, round 5, pin_f16.)
// Build and run on a GPU box: hipcc --offload-arch=gfx950 -O3 -o tools/exp/cvt_probe tools/exp/cvt_probe.hip && tools/exp/cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(unsigned base, unsigned stride, unsigned long long* nbad, unsigned* ex)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned bits = base + i * stride;
    const float x = __uint_as_float(bits);
    unsigned a, b;
    asm volatile("v_cvt_f16_f32 %0, %2\n\tv_cvt_pk_f16_f32 %1, %2, %2" : "=&v"(a), "=&v"(b) : "v"(x));
    const unsigned ha = a & 0xffffu, hb = b & 0xffffu, hb2 = b >> 16;
    if (ha != hb || hb != hb2) {
        const unsigned long long n = atomicAdd(nbad, 1ull);
        if (n < 16) { ex[3 * n] = bits; ex[3 * n + 1] = ha; ex[3 * n + 2] = hb; }
    }
}
int main()
{
    unsigned long long* nbad; unsigned* ex;
    hipMalloc(&nbad, 8); hipMalloc(&ex, 16 * 3 * 4);
    struct { const char* name; unsigned base, stride; } sweeps[] = {
        {"every 64th fp32 bit pattern (all exponents, both signs)", 0u, 64u},
        {"dense: 2^26 consecutive patterns from 1.0f", 0x3f800000u, 1u},
        {"dense: 2^26 consecutive patterns around the fp16 denormal range (6e-5)", 0x38000000u, 1u},
        {"dense: 2^26 consecutive patterns below 65504 (0x477f0000)", 0x477e0000u, 1u}};
    for (auto& s : sweeps) {
        hipMemset(nbad, 0, 8); hipMemset(ex, 0, 16 * 3 * 4);
        hipLaunchKernelGGL(k, dim3((1u << 26) / 256), dim3(256), 0, 0, s.base, s.stride, nbad, ex);
        unsigned long long n; unsigned e[48];
        hipMemcpy(&n, nbad, 8, hipMemcpyDeviceToHost); hipMemcpy(e, ex, sizeof(e), hipMemcpyDeviceToHost);
        printf("%-72s mismatches: %llu of %u\n", s.name, n, 1u << 26);
        for (unsigned j = 0; j < (n < 6 ? n : 6); ++j) { float f; memcpy(&f, &e[3 * j], 4); printf("    x = %.9g (0x%08x): v_cvt_f16_f32 -> 0x%04x, v_cvt_pk_f16_f32 -> 0x%04x\n", f, e[3 * j], e[3 * j + 1], e[3 * j + 2]); }
    }
    return 0;
}
