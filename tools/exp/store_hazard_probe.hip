// store_hazard_probe.hip -- stand-alone reproduction of the gfx950 store-data hazard of DESIGN.md section 3 item 8.
// Each lane stores 16 (or 8) bytes with a buffer store and overwrites one of the data registers with a marker NOPS issue slots later (all in one asm
// block, so nothing can be scheduled in between); the host counts markers that reached memory.  Variants: soffset in an SGPR or the literal 0;
// dwordx4 / dwordx2; NOPS = 0..3.  A background stream of loads and stores in the same waves varies the memory pipeline's timing.
// Build: hipcc --offload-arch=gfx950 -O3 -o store_hazard_probe store_hazard_probe.hip ; run: ./store_hazard_probe [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MARK 0xdeadbeefu

template <int SREG, int X4, int NOPS>
__global__ __launch_bounds__(256) void probe(unsigned* out, const unsigned* bg, unsigned* sink, int iters, unsigned bytes)
{
    const unsigned lane = threadIdx.x, wg = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, bytes, 0x00020000);
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned slot = (wg * 256u + lane) * 16u;                      // 16 bytes per lane, rewritten every iteration
        const unsigned a = 0x10000000u | it, b = 0x20000000u | lane, c = 0x30000000u | wg, d = 0x40000000u | (it ^ lane);
        unsigned soff = SREG ? 0u : 0u;
        acc += bg[(wg * 977u + lane * 13u + it * 7919u) & 0xfffffu];          // unrelated traffic of the same wave
        unsigned vo = slot, so = 0;
        if (SREG) { so = 64u * 1024u * 1024u; vo = slot; }                   // the SGPR form addresses the second half of the buffer
        (void)soff;
        if (X4) {
            if (SREG)
                asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %2\n\tv_mov_b32 v23, %3\n\ts_nop 4\n\t"
                             "buffer_store_dwordx4 v[20:23], %4, %5, %6 offen\n\t"
                             ".rept %7\n\ts_nop 0\n\t.endr\n\t"
                             "v_mov_b32 v22, 0xdeadbeef\n\ts_waitcnt vmcnt(0)"
                             :: "v"(a), "v"(b), "v"(c), "v"(d), "v"(vo), "s"(rs), "s"(so), "n"(NOPS) : "v20", "v21", "v22", "v23", "memory");
            else
                asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %2\n\tv_mov_b32 v23, %3\n\ts_nop 4\n\t"
                             "buffer_store_dwordx4 v[20:23], %4, %5, 0 offen\n\t"
                             ".rept %6\n\ts_nop 0\n\t.endr\n\t"
                             "v_mov_b32 v22, 0xdeadbeef\n\ts_waitcnt vmcnt(0)"
                             :: "v"(a), "v"(b), "v"(c), "v"(d), "v"(vo), "s"(rs), "n"(NOPS) : "v20", "v21", "v22", "v23", "memory");
        } else {
            if (SREG)
                asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, %1\n\ts_nop 4\n\t"
                             "buffer_store_dwordx2 v[20:21], %2, %3, %4 offen\n\t"
                             ".rept %5\n\ts_nop 0\n\t.endr\n\t"
                             "v_mov_b32 v21, 0xdeadbeef\n\ts_waitcnt vmcnt(0)"
                             :: "v"(a), "v"(b), "v"(vo), "s"(rs), "s"(so), "n"(NOPS) : "v20", "v21", "memory");
            else
                asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, %1\n\ts_nop 4\n\t"
                             "buffer_store_dwordx2 v[20:21], %2, %3, 0 offen\n\t"
                             ".rept %4\n\ts_nop 0\n\t.endr\n\t"
                             "v_mov_b32 v21, 0xdeadbeef\n\ts_waitcnt vmcnt(0)"
                             :: "v"(a), "v"(b), "v"(vo), "s"(rs), "n"(NOPS) : "v20", "v21", "memory");
        }
        // read the slot back at once: a marker in memory is a store that picked up the overwritten register
        const unsigned base = (SREG ? 16u * 1024u * 1024u : 0u) + (wg * 256u + lane) * 4u;
        const unsigned got = __builtin_nontemporal_load(out + base + (X4 ? 2 : 1));
        if (got == MARK) atomicAdd(sink + 1, 1u);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int SREG, int X4, int NOPS>
static void run(const char* name, unsigned* out, unsigned* bg, unsigned* sink, int iters)
{
    hipMemset(sink, 0, 8);
    hipLaunchKernelGGL((probe<SREG, X4, NOPS>), dim3(1024), dim3(256), 0, 0, out, bg, sink, iters, 128u * 1024u * 1024u);
    hipDeviceSynchronize();
    unsigned h[2];
    hipMemcpy(h, sink, 8, hipMemcpyDeviceToHost);
    printf("%-44s overwrite %d issue slot(s) behind the store: %8u markers in memory of %lld stores\n", name, NOPS, h[1], 1024LL * 256 * iters);
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned *out, *bg, *sink;
    hipMalloc(&out, 128u * 1024u * 1024u);
    hipMalloc(&bg, 4u * 1024u * 1024u);
    hipMalloc(&sink, 8);
    hipMemset(out, 0, 128u * 1024u * 1024u);
    hipMemset(bg, 0, 4u * 1024u * 1024u);
    run<1, 1, 0>("buffer_store_dwordx4, soffset = SGPR,", out, bg, sink, iters);
    run<1, 1, 1>("buffer_store_dwordx4, soffset = SGPR,", out, bg, sink, iters);
    run<1, 1, 2>("buffer_store_dwordx4, soffset = SGPR,", out, bg, sink, iters);
    run<1, 1, 3>("buffer_store_dwordx4, soffset = SGPR,", out, bg, sink, iters);
    run<0, 1, 0>("buffer_store_dwordx4, soffset = literal 0,", out, bg, sink, iters);
    run<0, 1, 1>("buffer_store_dwordx4, soffset = literal 0,", out, bg, sink, iters);
    run<0, 1, 2>("buffer_store_dwordx4, soffset = literal 0,", out, bg, sink, iters);
    run<1, 0, 0>("buffer_store_dwordx2, soffset = SGPR,", out, bg, sink, iters);
    run<0, 0, 0>("buffer_store_dwordx2, soffset = literal 0,", out, bg, sink, iters);
    return 0;
}
