// mfma_src_war_probe.hip -- does v_mfma_f32_32x32x16_f16 read its SrcA / SrcB registers at issue?  hipcc lets a VALU write of a source register
// follow the MFMA with no wait state.  Each wave issues MFMAS matrix instructions that read one B (or A) fragment, overwrites ONE register of
// that fragment NOPS issue slots behind the last of them (all in one asm block), lets the pipe drain and compares the accumulators with a run
// whose overwrite comes after the drain.  DEPTH = independent MFMAs queued ahead (a longer queue = a later start of the last one); WAVES per
// workgroup of 2*WAVES*64 threads... (two waves per SIMD when WAVES = 8).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_src_war_probe mfma_src_war_probe.hip ; run: ./mfma_src_war_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// WHICH: 0 = overwrite a SrcB register, 1 = a SrcA register; REG: 0..3; NOPS: issue slots between the last MFMA and the overwrite;
// DEPTH: 1, 4 or 8 MFMAs on separate accumulators reading the same A and B
template <int WHICH, int REG, int NOPS, int DEPTH>
__global__ __launch_bounds__(512) void probe(const unsigned* __restrict__ ab, unsigned* __restrict__ bad, int iters)
{
    const unsigned lane = threadIdx.x & 63;
    const unsigned a0 = ab[lane * 8 + 0], a1 = ab[lane * 8 + 1], a2 = ab[lane * 8 + 2], a3 = ab[lane * 8 + 3];
    const unsigned b0 = ab[lane * 8 + 4], b1 = ab[lane * 8 + 5], b2 = ab[lane * 8 + 6], b3 = ab[lane * 8 + 7];
    unsigned wrong = 0;
    for (int it = 0; it < iters; ++it) {
        float r_ref, r_tst;
        // the accumulators are a[0:15] .. of the LAST queued MFMA; every MFMA reads v[40:43] (B) and v[44:47] (A)
#define BODY_(NOPS_, OUT_)                                                                                                  \
        asm volatile(                                                                                                       \
            "v_mov_b32 v44, %1\n\tv_mov_b32 v45, %2\n\tv_mov_b32 v46, %3\n\tv_mov_b32 v47, %4\n\t"                          \
            "v_mov_b32 v40, %5\n\tv_mov_b32 v41, %6\n\tv_mov_b32 v42, %7\n\tv_mov_b32 v43, %8\n\t"                          \
            "s_nop 4\n\t"                                                                                                   \
            ".if %9 >= 8\n\t"                                                                                               \
            "v_mfma_f32_32x32x16_f16 a[64:79], v[44:47], v[40:43], 0\n\tv_mfma_f32_32x32x16_f16 a[80:95], v[44:47], v[40:43], 0\n\t"      \
            "v_mfma_f32_32x32x16_f16 a[96:111], v[44:47], v[40:43], 0\n\tv_mfma_f32_32x32x16_f16 a[112:127], v[44:47], v[40:43], 0\n\t"   \
            ".endif\n\t"                                                                                                    \
            ".if %9 >= 4\n\t"                                                                                               \
            "v_mfma_f32_32x32x16_f16 a[16:31], v[44:47], v[40:43], 0\n\tv_mfma_f32_32x32x16_f16 a[32:47], v[44:47], v[40:43], 0\n\t"      \
            "v_mfma_f32_32x32x16_f16 a[48:63], v[44:47], v[40:43], 0\n\t"                                                   \
            ".endif\n\t"                                                                                                    \
            "v_mfma_f32_32x32x16_f16 a[0:15], v[44:47], v[40:43], 0\n\t"                                                    \
            ".rept " #NOPS_ "\n\ts_nop 0\n\t.endr\n\t"                                                                      \
            "v_mov_b32 v%10, 0x7bff7bff\n\t"                                                                                \
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"              \
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"              \
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"              \
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"              \
            "v_accvgpr_read_b32 %0, a5\n\t"                                                                                 \
            : "=v"(OUT_) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "n"(DEPTH), "n"((WHICH ? 44 : 40) + REG)           \
            : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12",  \
              "a13", "a14", "a15", "a16", "a31", "a32", "a47", "a48", "a63", "a64", "a79", "a80", "a95", "a96", "a111", "a112", "a127", "memory")
        BODY_(512, r_ref);          // reference: the overwrite comes 512 issue slots later -- long after the pipe has drained
        if (NOPS == 0) { BODY_(0, r_tst); } else if (NOPS == 1) { BODY_(1, r_tst); } else if (NOPS == 2) { BODY_(2, r_tst); } else if (NOPS == 4) { BODY_(4, r_tst); } else { BODY_(8, r_tst); }
#undef BODY_
        wrong += __float_as_uint(r_ref) != __float_as_uint(r_tst);
        __syncthreads();
    }
    if (wrong) atomicAdd(bad, wrong);
}

template <int WHICH, int REG, int NOPS, int DEPTH>
static void run(const unsigned* ab, unsigned* bad, int iters)
{
    (void)hipMemset(bad, 0, 4);
    hipLaunchKernelGGL((probe<WHICH, REG, NOPS, DEPTH>), dim3(256), dim3(512), 0, 0, ab, bad, iters);
    (void)hipDeviceSynchronize();
    unsigned h = 0;
    (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("overwrite Src%c register %d, %d issue slot(s) behind the last of %d queued MFMAs: %u of %lld lane-results differ\n", WHICH ? 'A' : 'B', REG, NOPS, DEPTH, h,
           256LL * 512 * iters);
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    std::vector<unsigned> h(64 * 8);
    srand(7);
    for (auto& v : h) {                                                   // two fp16 values per dword, magnitudes ~1
        const unsigned short lo = (unsigned short)(0x3800 + rand() % 0x400), hi = (unsigned short)(0x3800 + rand() % 0x400);
        v = lo | ((unsigned)hi << 16);
    }
    unsigned *ab, *bad;
    (void)hipMalloc(&ab, h.size() * 4);
    (void)hipMalloc(&bad, 4);
    (void)hipMemcpy(ab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0, 0, 0, 1>(ab, bad, iters); run<0, 3, 0, 1>(ab, bad, iters); run<1, 0, 0, 1>(ab, bad, iters); run<1, 3, 0, 1>(ab, bad, iters);
    run<0, 0, 0, 4>(ab, bad, iters); run<0, 3, 0, 4>(ab, bad, iters); run<1, 3, 0, 4>(ab, bad, iters);
    run<0, 0, 0, 8>(ab, bad, iters); run<0, 3, 0, 8>(ab, bad, iters); run<1, 0, 0, 8>(ab, bad, iters); run<1, 3, 0, 8>(ab, bad, iters);
    run<0, 3, 1, 8>(ab, bad, iters); run<0, 3, 2, 8>(ab, bad, iters); run<0, 3, 4, 8>(ab, bad, iters); run<0, 3, 8, 8>(ab, bad, iters);
    return 0;
}
