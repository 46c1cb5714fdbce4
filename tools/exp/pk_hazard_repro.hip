// pk_hazard_repro.hip -- STAND-ALONE reproducer (no library): the swizzled packed-fp32 add `v_pk_add_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]` next to MFMA waves of ANOTHER kernel
// on the same SIMD (two streams).  Victim: 4M threads x 64 repetitions of the instruction on per-thread data; aggressor: a small-footprint kernel (4 waves per workgroup, two
// workgroups per CU, no LDS) that keeps the matrix pipe busy.  Prints how many victim results differ from the victim run alone, per round.
// Build / run on an MI355X: hipcc --offload-arch=gfx950 -O3 -o tools/exp/pk_hazard_repro tools/exp/pk_hazard_repro.hip && tools/exp/pk_hazard_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int SWZ> __global__ void victim(const float* __restrict__ in, float* out, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f2 v; v.x = in[2 * i]; v.y = in[2 * i + 1];
    f2 c; c.x = 0.4947f; c.y = 1.25f;
    f2 acc; acc.x = 0.f; acc.y = 0.f;
#pragma unroll 4
    for (int k = 0; k < 64; ++k) {
        f2 r;
        if (SWZ) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 4" : "=&v"(r) : "v"(v), "v"(c));
        else     asm volatile("v_pk_add_f32 %0, %1, %2\n\ts_nop 4" : "=&v"(r) : "v"(v), "v"(c));
        acc.x += r.x; acc.y += r.y;
        v.x += 0.001f; v.y -= 0.001f;
    }
    out[2 * i] = acc.x; out[2 * i + 1] = acc.y;
}

// aggressor variants: 0 = MFMAs only; 1 = MFMAs + v_permlane32_swap_b32 on their results; 2 = v_permlane32_swap_b32 only; 3 = MFMAs + LDS reads
template <int V> __global__ __launch_bounds__(256, 2) void aggressor(float* sink, int iters)
{
    __shared__ float lds[1024];
    const int lane = threadIdx.x & 63;
    if (V == 3) { for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = 0.001f * i; __syncthreads(); }
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane - e)); }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float p = 0.5f * lane, q = 0.25f * lane;
    for (int k = 0; k < iters; ++k) {
        if (V != 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
        }
        if (V == 1 || V == 2) {
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
            p += 1.f; q -= 1.f;
        }
        if (V == 3) p += lds[(lane * 4 + k) & 1023];
    }
    float t = p + q;
    for (int r = 0; r < 16; ++r) t += acc0[r] + acc1[r];
    if (t == 1234.5f) sink[threadIdx.x] = t;
}

int main()
{
    const unsigned n = 1u << 22;
    std::vector<float> h(2 * n);
    unsigned s = 1u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) / (float)(1 << 22); }
    float *din, *dout, *dsink;
    hipMalloc(&din, 8ull * n); hipMalloc(&dout, 8ull * n); hipMalloc(&dsink, 4096);
    hipMemcpy(din, h.data(), 8ull * n, hipMemcpyHostToDevice);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    std::vector<float> ref(2 * n), got(2 * n);
    for (int av = 0; av < 4; ++av)
    for (int swz = 1; swz >= (av == 0 ? 0 : 1); --swz) {
        auto run_aggr = [&]() {
            if (av == 0) hipLaunchKernelGGL(aggressor<0>, dim3(2 * cus), dim3(256), 0, s1, dsink, 2000);
            else if (av == 1) hipLaunchKernelGGL(aggressor<1>, dim3(2 * cus), dim3(256), 0, s1, dsink, 2000);
            else if (av == 2) hipLaunchKernelGGL(aggressor<2>, dim3(2 * cus), dim3(256), 0, s1, dsink, 8000);
            else hipLaunchKernelGGL(aggressor<3>, dim3(2 * cus), dim3(256), 0, s1, dsink, 2000);
        };
        static const char* an[4] = {"MFMAs only", "MFMAs + v_permlane32_swap_b32", "v_permlane32_swap_b32 only", "MFMAs + LDS reads"};
        printf("aggressor: %s | ", an[av]);
        auto run_victim = [&](hipStream_t st) { if (swz) hipLaunchKernelGGL(victim<1>, dim3(n / 256), dim3(256), 0, st, din, dout, n); else hipLaunchKernelGGL(victim<0>, dim3(n / 256), dim3(256), 0, st, din, dout, n); };
        run_victim(s2); hipDeviceSynchronize();
        hipMemcpy(ref.data(), dout, 8ull * n, hipMemcpyDeviceToHost);
        printf("%s\n", swz ? "victim: v_pk_add_f32 WITH op_sel:[0,1] op_sel_hi:[1,0]" : "victim: v_pk_add_f32 without op_sel (control)");
        for (int round = 0; round < 5; ++round) {
            hipEvent_t ev; hipEventCreate(&ev);
            for (int k = 0; k < 4; ++k) run_victim(s2);
            hipEventRecord(ev, s2);
            int launches = 0;
            while (hipEventQuery(ev) != hipSuccess) { for (int k = 0; k < 8; ++k) run_aggr(); launches += 8; }
            hipDeviceSynchronize();
            hipMemcpy(got.data(), dout, 8ull * n, hipMemcpyDeviceToHost);
            unsigned long long bad = 0;
            for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
            printf("   round %d: %llu of %u values differ from the victim run alone (%d aggressor launches meanwhile)\n", round, bad, 2 * n, launches);
            hipEventDestroy(ev);
        }
    }
    return 0;
}
