// probe of buffer_load_dwordx4 ... lds semantics on gfx950: lane stride, LDS offsets > 64 KiB, out-of-range lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void probe(const unsigned* src, unsigned* dst, int lds_off, int nbytes_valid, unsigned soff)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 40960; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, (unsigned)nbytes_valid, 0x00020000);
    if (threadIdx.x < 64) {
        unsigned voff = threadIdx.x * 16u;
        if (threadIdx.x == 5) voff = 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + lds_off), 16, voff, soff, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256 + 64; i += blockDim.x) dst[i] = reinterpret_cast<unsigned*>(smem + lds_off - 128)[i];
}
int main()
{
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    unsigned *s, *d;
    hipMalloc(&s, 4096 * 4); hipMalloc(&d, 4096 * 4);
    hipMemcpy(s, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    int offs[3] = {1024, 60000 - 60000 % 16, 100000 - 100000 % 16};
    for (int t = 0; t < 3; ++t) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 163840, 0, s, d, offs[t], 1000, t == 2 ? 2048u : 0u);
        hipDeviceSynchronize();
        std::vector<unsigned> o(320);
        hipMemcpy(o.data(), d, 320 * 4, hipMemcpyDeviceToHost);
        printf("lds_off %d (soff %u): before: %x %x | lanes0-7 dwords:", offs[t], t == 2 ? 2048u : 0u, o[30], o[31]);
        for (int i = 32; i < 32 + 32; ++i) printf(" %x", o[i]);
        printf(" ... lane61-63:");
        for (int i = 32 + 61 * 4; i < 32 + 64 * 4; ++i) printf(" %x", o[i]);
        printf(" | after: %x %x\n", o[32 + 256], o[32 + 257]);
    }
    return 0;
}
