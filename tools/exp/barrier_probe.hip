// barrier_probe.hip -- stand-alone probe for the round-4 fault study: can a wave of an 8-wave workgroup observe LDS
// contents "from the future" under the protocol of round 3's coupling_head
//     write tile(i) -> barrier B1 -> read tile(i) (ds_read_b128 + MFMA, loads in flight) -> barrier B2 -> register-only work + stores -> write tile(i+1)
// i.e. does a wave that passed B2 ever overwrite the tile while another wave is still in its reads?  Every tile word carries the
// iteration number; every fragment a wave reads between B1 and B2 is checked against it.  Per-wave work is made uneven on purpose.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/barrier_probe.hip -o /tmp/barrier_probe && /tmp/barrier_probe [iters] [waves]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PW = 34;

template <int NWV>
__global__ __launch_bounds__(NWV * 64, 2) void probe(const float* __restrict__ src, float* __restrict__ dst, unsigned* __restrict__ bad, int iters, int n)
{
    constexpr int NPOS = (NWV + 2) * PW, NT = NWV * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sZ = smem;                               // [3 planes][NPOS][16 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    unsigned nbad = 0, first = 0;
    float pre[16];
    const float* sp = src + ((size_t)blockIdx.x * NT + tid) * 16 % n;
#pragma unroll
    for (int r = 0; r < 16; ++r) pre[r] = sp[r];
    f32x16 acc;
    for (int i = 1; i <= iters; ++i) {
        if (tid < NPOS) {
            const uint4 tok = make_uint4((unsigned)i, (unsigned)i, (unsigned)i, (unsigned)i);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint4*>(sZ + (pl * NPOS + tid) * 16) = tok;
        }
        __syncthreads();                                    // B1
        float nxt[16];
        const float* np_ = src + (((size_t)blockIdx.x * NT + tid) * 16 + (size_t)i * 4096) % n;
#pragma unroll
        for (int r = 0; r < 16; ++r) nxt[r] = np_[r];       // loads in flight under the reads, as the head's prefetch
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int t0 = 2 * j, t1 = 2 * j + 1 < 9 ? 2 * j + 1 : 2 * j;
            const int a0 = ((t0 / 3) * PW + (t0 % 3)) * 16, a1 = ((t1 / 3) * PW + (t1 % 3)) * 16;
            const unsigned char* bp = sZ + (lhi ? a1 : a0) + (wave * PW + l31) * 16;
            uint4 f[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) f[pl] = *reinterpret_cast<const uint4*>(bp + pl * NPOS * 16);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const unsigned v = f[pl].x | f[pl].y | f[pl].z | f[pl].w, w = f[pl].x & f[pl].y & f[pl].z & f[pl].w;
                if (v != (unsigned)i || w != (unsigned)i) { if (!nbad) first = ((unsigned)i << 16) | (f[pl].x & 0xffffu); ++nbad; }
                const bf16x8 b = __builtin_bit_cast(bf16x8, f[pl]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, acc, 0, 0, 0);
            }
        }
        __syncthreads();                                    // B2
        // register-only phase of uneven length (wave- and iteration-dependent), then stores
        float v = 0.f;
        const int reps = 1 + ((wave * 7 + i * 3 + blockIdx.x) & 15);
        for (int k = 0; k < reps; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) v = fmaf(v, 1.0001f, acc[r] + pre[r]);
        float* dp = dst + ((size_t)blockIdx.x * NT + tid) * 4;
        dp[i & 3] = v;
#pragma unroll
        for (int r = 0; r < 16; ++r) pre[r] = nxt[r];
    }
    if (nbad) {
        const unsigned k = atomicAdd(bad, 1u);
        if (k < 64) { bad[1 + 3 * k] = (blockIdx.x << 8) | tid; bad[2 + 3 * k] = nbad; bad[3 + 3 * k] = first; }
    }
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, waves = argc > 2 ? atoi(argv[2]) : 8;
    const int n = 64 << 20;
    float *src, *dst; unsigned* bad;
    hipMalloc(&src, (size_t)n * 4); hipMalloc(&dst, (size_t)1024 * 512 * 4 * 4); hipMalloc(&bad, 4096);
    hipMemset(src, 0, (size_t)n * 4); hipMemset(bad, 0, 4096);
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (waves == 8) hipLaunchKernelGGL(probe<8>, dim3(cus), dim3(512), 3 * 10 * PW * 16, 0, src, dst, bad, iters, n);
    else            hipLaunchKernelGGL(probe<4>, dim3(cus * 2), dim3(256), 3 * 6 * PW * 16, 0, src, dst, bad, iters, n);
    hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(1024);
    hipMemcpy(h.data(), bad, 4096, hipMemcpyDeviceToHost);
    printf("barrier_probe waves=%d: %d workgroups x %d iterations in %.1f ms: %u threads saw a foreign token\n", waves, waves == 8 ? cus : 2 * cus, iters, ms, h[0]);
    for (unsigned k = 0; k < h[0] && k < 8; ++k)
        printf("  block %u thread %u: %u bad fragments, first at iteration %u saw token %u\n", h[1 + 3 * k] >> 8, h[1 + 3 * k] & 255u | (h[1 + 3 * k] & 0x100u),
               h[2 + 3 * k], h[3 + 3 * k] >> 16, h[3 + 3 * k] & 0xffffu);
    return 0;
}
