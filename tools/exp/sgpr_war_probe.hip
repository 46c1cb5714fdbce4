// sgpr_war_probe.hip -- is an SALU write to an SGPR (pair) that directly follows a VALU instruction READING it safe on gfx950 when the VALU of the SIMD is
// kept busy by MFMAs of another wave?  (hipcc -O3 emits exactly `v_pk_mul_f32 v[4:5], s[12:13], v[4:5]; s_mov_b64 s[12:13], -1` in resize_kernel.)
// Victim kernels for tools/exp/sgpr_war_probe.py; every thread repeats the pattern REP times and accumulates.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/exp/libsgprwar.so tools/exp/sgpr_war_probe.hip   (inline asm: not affected by the library's NOPK flag)
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 64
// mode 0: v_pk_mul_f32 reading s[12:13], next instruction overwrites s[12:13]      (the compiled pattern)
// mode 1: the same with s_nop 7 between them
// mode 2: v_mul_f32 reading s12 (32-bit), next instruction overwrites s12
// mode 3: v_pk_mul_f32 reading a pair that is NOT overwritten (control)
template <int MODE> __device__ __forceinline__ f2 step(f2 v, float a, float b)
{
    f2 r;
    if (MODE == 0)
        asm volatile("s_mov_b32 s12, %2\n\ts_mov_b32 s13, %3\n\ts_nop 7\n\tv_pk_mul_f32 %0, s[12:13], %1\n\ts_mov_b64 s[12:13], -1\n\ts_nop 0" : "=v"(r) : "v"(v), "s"(a), "s"(b) : "s12", "s13");
    else if (MODE == 1)
        asm volatile("s_mov_b32 s12, %2\n\ts_mov_b32 s13, %3\n\ts_nop 7\n\tv_pk_mul_f32 %0, s[12:13], %1\n\ts_nop 7\n\ts_mov_b64 s[12:13], -1\n\ts_nop 0" : "=v"(r) : "v"(v), "s"(a), "s"(b) : "s12", "s13");
    else if (MODE == 2) {
        float x, y;
        asm volatile("s_mov_b32 s12, %2\n\ts_nop 7\n\tv_mul_f32 %0, s12, %1\n\ts_mov_b32 s12, -1\n\ts_nop 0" : "=v"(x) : "v"(v.x), "s"(a) : "s12");
        asm volatile("s_mov_b32 s12, %2\n\ts_nop 7\n\tv_mul_f32 %0, s12, %1\n\ts_mov_b32 s12, -1\n\ts_nop 0" : "=v"(y) : "v"(v.y), "s"(b) : "s12");
        r.x = x; r.y = y;
    } else if (MODE == 3)
        asm volatile("s_mov_b32 s12, %2\n\ts_mov_b32 s13, %3\n\ts_nop 7\n\tv_pk_mul_f32 %0, s[12:13], %1\n\ts_nop 0" : "=v"(r) : "v"(v), "s"(a), "s"(b) : "s12", "s13");
    else {
        // modes 4 .. 7: the dependent packed chain of the compiled bilinear blend: pk_mul -> (s_nop N) -> pk_add (op_sel swizzle) -> (s_nop N) -> pk_mul -> (s_nop N) -> add
        f2 t0, t1;
        f2 c; c.x = a; c.y = b;
        if (MODE == 4)
            asm volatile("v_pk_mul_f32 %0, %2, %3\n\tv_pk_mul_f32 %1, %3, %2\n\ts_nop 0\n\tv_pk_add_f32 %1, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 0\n\tv_pk_mul_f32 %1, %2, %1"
                         : "=&v"(t0), "=&v"(t1) : "v"(v), "v"(c));
        else if (MODE == 5)
            asm volatile("v_pk_mul_f32 %0, %2, %3\n\tv_pk_mul_f32 %1, %3, %2\n\ts_nop 4\n\tv_pk_add_f32 %1, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 4\n\tv_pk_mul_f32 %1, %2, %1"
                         : "=&v"(t0), "=&v"(t1) : "v"(v), "v"(c));
        else if (MODE == 6)      // independent VGPR x VGPR packed multiplies only
            asm volatile("v_pk_mul_f32 %0, %2, %3\n\tv_pk_mul_f32 %1, %3, %2\n\ts_nop 4" : "=&v"(t0), "=&v"(t1) : "v"(v), "v"(c));
        else if (MODE == 7)      // packed multiply -> dependent packed multiply (no op_sel anywhere)
            asm volatile("v_pk_mul_f32 %0, %2, %3\n\ts_nop 4\n\tv_pk_mul_f32 %1, %0, %3\n\ts_nop 4" : "=&v"(t0), "=&v"(t1) : "v"(v), "v"(c));
        else if (MODE == 8)      // packed add with the op_sel swizzle on independent inputs
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 4" : "=&v"(t1) : "v"(v), "v"(c));
        else if (MODE == 9)      // packed add WITHOUT op_sel, dependent on a packed multiply
            asm volatile("v_pk_mul_f32 %0, %2, %3\n\ts_nop 4\n\tv_pk_add_f32 %1, %0, %3\n\ts_nop 4" : "=&v"(t0), "=&v"(t1) : "v"(v), "v"(c));
        r = t1;
    }
    return r;
}
template <int MODE> __global__ void k_war(const float* __restrict__ in, float* out, unsigned n, float a, float b)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f2 v; v.x = in[2 * i]; v.y = in[2 * i + 1];
    f2 acc; acc.x = 0.f; acc.y = 0.f;
#pragma unroll 4
    for (int k = 0; k < REP; ++k) {
        const f2 r = step<MODE>(v, a, b);
        acc.x += r.x; acc.y += r.y;
        v.x += 0.001f; v.y -= 0.001f;
    }
    out[2 * i] = acc.x; out[2 * i + 1] = acc.y;
}
extern "C" int run_war(int mode, const float* in, float* out, unsigned n, float a, float b, void* s)
{
    dim3 g((n + 255) / 256), t(256);
    hipStream_t st = (hipStream_t)s;
    if (mode == 0) hipLaunchKernelGGL(k_war<0>, g, t, 0, st, in, out, n, a, b);
    else if (mode == 1) hipLaunchKernelGGL(k_war<1>, g, t, 0, st, in, out, n, a, b);
    else if (mode == 2) hipLaunchKernelGGL(k_war<2>, g, t, 0, st, in, out, n, a, b);
    else if (mode == 3) hipLaunchKernelGGL(k_war<3>, g, t, 0, st, in, out, n, a, b);
    else if (mode == 4) hipLaunchKernelGGL(k_war<4>, g, t, 0, st, in, out, n, a, b);
    else if (mode == 5) hipLaunchKernelGGL(k_war<5>, g, t, 0, st, in, out, n, a, b);
    else if (mode == 6) hipLaunchKernelGGL(k_war<6>, g, t, 0, st, in, out, n, a, b);
    else if (mode == 7) hipLaunchKernelGGL(k_war<7>, g, t, 0, st, in, out, n, a, b);
    else if (mode == 8) hipLaunchKernelGGL(k_war<8>, g, t, 0, st, in, out, n, a, b);
    else hipLaunchKernelGGL(k_war<9>, g, t, 0, st, in, out, n, a, b);
    return (int)hipGetLastError();
}
