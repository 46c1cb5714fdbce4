// xcd_handoff_probe.hip -- round 5: which in-launch hand-off forms are safe for the fused dense-block chain (conv_chain.hip)?
// The chain kernel hands conv outputs from one workgroup to another INSIDE a launch.  Two things must hold on gfx950 (8 XCDs with
// private L2s, per-CU L1 never refreshed by other CUs' stores):
//   (1) data written by a producer (write-through `sc1` 16-byte stores, `s_waitcnt vmcnt(0)`, then an agent-scope atomic on a counter)
//       is what a consumer on ANY XCD reads after it has seen the counter, and
//   (2) this also holds for a cache line the consumer's CU/XCD has ALREADY cached before the producer wrote part of it (false sharing:
//       tile rows of neighbouring tiles share 128-byte lines when W % 8 != 0, and the ring buffers are rewritten every 4 dense blocks).
// Protocol per iteration `it` and pair (producer wg p, consumer wg c):  producer writes the A halves (bytes 0..63) of its lines = it,
// publishes flagA; consumer reads WHOLE lines (so the B halves, still it-1, are cached on its side), acks; producer writes the B halves
// (bytes 64..127) = it, publishes flagB; consumer reads the B halves with the flavour under test and counts words != it (stale).
//   consumer flavours: 0 plain loads, no fence (control: must show stale)   1 buffer loads aux=sc1   2 acquire fence (agent) + plain loads
//                      3 LDS-DMA (buffer_load ... lds) aux=sc1               4 acquire fence + plain LDS-DMA       5 buffer loads aux=sc0|sc1
//   producer flavours: 0 sc1 16-byte stores + vmcnt(0)                      1 plain stores + release fence (agent)
//   hipcc --offload-arch=gfx950 -O3 tools/exp/xcd_handoff_probe.hip -o tools/exp/xcd_handoff_probe && tools/exp/xcd_handoff_probe [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int NL = 64;                       // lines per pair
constexpr int LINE = 128;
constexpr unsigned SPIN_MAX = 4u << 20;

__device__ __forceinline__ unsigned ld_relaxed(unsigned* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_relaxed(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one lane polls; returns false on timeout / abort
__device__ bool wait_eq(unsigned* flag, unsigned want, unsigned* abort_)
{
    bool ok = true;
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (ld_relaxed(flag) != want) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 1023u) == 0 && ld_relaxed(abort_)) { ok = false; break; }
            if (spins > SPIN_MAX) { st_relaxed(abort_, 1u); ok = false; break; }
        }
    }
    return __syncthreads_and((int)ok) != 0;
}

template <int CF, int PF>
__global__ __launch_bounds__(256) void probe(unsigned char* buf, unsigned* flags, unsigned* stale, unsigned* abort_, int iters, int shift)
{
    __shared__ __attribute__((aligned(16))) unsigned char lbuf[NL * LINE];
    const int G = gridDim.x, tid = threadIdx.x, lane = tid & 63;
    // roles: workgroups whose (blockIdx / shift) is even produce for blockIdx + shift
    const int grp = blockIdx.x / shift;
    const bool producer = (grp & 1) == 0;
    const int pair = producer ? blockIdx.x : blockIdx.x - shift;
    if (pair + shift >= G) return;
    unsigned char* reg = buf + (size_t)pair * NL * LINE;
    unsigned* fA = flags + pair * 64, *aA = fA + 16, *fB = fA + 32, *aB = fA + 48;      // one line each
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reg, 0, NL * LINE, 0x00020000);
    unsigned nstale = 0, nstaleA = 0;
    for (int it = 1; it <= iters; ++it) {
        const u32x4 tok = {(unsigned)it, (unsigned)it, (unsigned)it, (unsigned)it};
        if (producer) {
            if (it > 1 && !wait_eq(aB, (unsigned)(it - 1), abort_)) return;
            // A halves: 64 lines x 4 pieces of 16 B = 256 pieces, one per thread
            {
                const unsigned off = (unsigned)((tid >> 2) * LINE + (tid & 3) * 16);
                if (PF == 0) __builtin_amdgcn_raw_buffer_store_b128(tok, rs, off, 0, 16);
                else *reinterpret_cast<u32x4*>(reg + off) = tok;
            }
            if (PF == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (PF == 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                st_relaxed(fA, (unsigned)it);
            }
            if (!wait_eq(aA, (unsigned)it, abort_)) return;
            {
                const unsigned off = (unsigned)((tid >> 2) * LINE + 64 + (tid & 3) * 16);
                if (PF == 0) __builtin_amdgcn_raw_buffer_store_b128(tok, rs, off, 0, 16);
                else *reinterpret_cast<u32x4*>(reg + off) = tok;
            }
            if (PF == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (PF == 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                st_relaxed(fB, (unsigned)it);
            }
        } else {
            auto read2 = [&](int piece, u32x4& v) {                       // piece: 16-byte index inside the region
                const unsigned off = (unsigned)piece * 16u;
                if (CF == 0 || CF == 2) v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);        // plain (a volatile access would be emitted sc0 sc1)
                else if (CF == 1) v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
                else if (CF == 5) v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 17);
            };
            auto read_all = [&](u32x4 (&v)[2]) {                          // whole lines: 512 pieces, two per thread
                if (CF == 3 || CF == 4) {
                    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int p0 = (wave * 2 + j) * 64;               // 64 consecutive pieces per instruction
                        if (CF == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lbuf + p0 * 16), 16, (unsigned)(p0 + lane) * 16u, 0, 0, 16);
                        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lbuf + p0 * 16), 16, (unsigned)(p0 + lane) * 16u, 0, 0, 0);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < 2; ++j) v[j] = *reinterpret_cast<u32x4*>(lbuf + ((wave * 2 + j) * 64 + lane) * 16);
                    __syncthreads();
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) read2(tid * 2 + j, v[j]);
                }
            };
            if (!wait_eq(fA, (unsigned)it, abort_)) return;
            if (CF == 2 || CF == 4) { if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __syncthreads(); }
            u32x4 v[2];
            read_all(v);
            // pieces tid*2, tid*2+1 (or the LDS-DMA mapping): piece index -> (line, piece-in-line); A half = pieces 0..3 of a line
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int piece = (CF == 3 || CF == 4) ? ((tid >> 6) * 2 + j) * 64 + lane : tid * 2 + j;
                if ((piece & 7) < 4) nstaleA += (v[j].x != (unsigned)it) + (v[j].w != (unsigned)it);
            }
            __syncthreads();
            if (tid == 0) st_relaxed(aA, (unsigned)it);
            if (!wait_eq(fB, (unsigned)it, abort_)) return;
            if (CF == 2 || CF == 4) { if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __syncthreads(); }
            read_all(v);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int piece = (CF == 3 || CF == 4) ? ((tid >> 6) * 2 + j) * 64 + lane : tid * 2 + j;
                if ((piece & 7) >= 4) nstale += (v[j].x != (unsigned)it) + (v[j].w != (unsigned)it);
            }
            __syncthreads();
            if (tid == 0) st_relaxed(aB, (unsigned)it);
        }
    }
    if (nstale) atomicAdd(stale, nstale);
    if (nstaleA) atomicAdd(stale + 1, nstaleA);
}

template <int CF, int PF>
static void run(int iters, int shift, unsigned char* buf, unsigned* flags, unsigned* stale, unsigned* abort_)
{
    const int G = 256;
    hipMemset(buf, 0, (size_t)G * NL * LINE);
    hipMemset(flags, 0, (size_t)G * 64 * 4);
    hipMemset(stale, 0, 8);
    hipMemset(abort_, 0, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<CF, PF>), dim3(G), dim3(256), 0, 0, buf, flags, stale, abort_, iters, shift);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned h[2] = {0, 0}, ab = 0;
    hipMemcpy(h, stale, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&ab, abort_, 4, hipMemcpyDeviceToHost);
    // words checked per pair and iteration in the B phase: 256 pieces x 2 words
    const double checked = (double)(G / 2) * iters * 512.0;
    printf("consumer %d producer %d shift %d (%s): stale B words %u of %.3g (%.4f %%), stale A words %u, abort %u, %.2f us per iteration\n", CF, PF, shift,
           shift % 8 ? "cross-XCD" : "same XCD", h[0], checked, 100.0 * h[0] / checked, h[1], ab, 1e3 * ms / iters);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned char* buf; unsigned *flags, *stale, *abort_;
    hipMalloc(&buf, (size_t)256 * NL * LINE);
    hipMalloc(&flags, 256 * 64 * 4);
    hipMalloc(&stale, 8);
    hipMalloc(&abort_, 4);
    for (int shift : {1, 8}) {
        run<0, 0>(iters, shift, buf, flags, stale, abort_);
        run<1, 0>(iters, shift, buf, flags, stale, abort_);
        run<5, 0>(iters, shift, buf, flags, stale, abort_);
        run<2, 0>(iters, shift, buf, flags, stale, abort_);
        run<3, 0>(iters, shift, buf, flags, stale, abort_);
        run<4, 0>(iters, shift, buf, flags, stale, abort_);
        run<0, 1>(iters, shift, buf, flags, stale, abort_);
        run<1, 1>(iters, shift, buf, flags, stale, abort_);
        run<2, 1>(iters, shift, buf, flags, stale, abort_);
        run<3, 1>(iters, shift, buf, flags, stale, abort_);
    }
    return 0;
}
