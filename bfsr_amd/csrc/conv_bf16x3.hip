// conv_bf16x3.hip -- fp32-accurate conv on the bf16 matrix pipe ("3xBF16" split, as in the BLAS libraries' fp32-emulation
// modes).  Every fp32 operand is split EXACTLY into three bf16 terms, x = xh + xm + xl (round-to-nearest at each level: 8+8+8
// significant bits), and the product is evaluated as the six terms
//        xh*wh + xh*wm + xm*wh + xm*wm + xh*wl + xl*wh        (each exact in the fp32 accumulator)
// on v_mfma_f32_32x32x16_bf16; the three dropped terms are <= 2^-24 + 2^-24 + 2^-32 relative to |x*w|, i.e. at the level of
// the one rounding an fp32 multiply makes anyway.  Accumulation and the whole epilogue are fp32, tensors in HBM are fp32.
// Six bf16 MFMAs cost 6/16 of the fp32 MFMA (32x32x2) time for the same contraction: the effective peak of this scheme is
// 2.5 PFLOP/s / 6 = 417 TFLOP/s of fp32-equivalent work against 157 TFLOP/s for the native fp32 MFMA.
// Error against an fp64 conv is measured next to the native fp32 kernel in tests/test_hip_ops.py (same magnitude).
//
// Structure = conv_f16.hip: M = cout (MR tiles of 32), N = 32 pixels of a row, K = 16 input channels per MFMA; LDS holds the
// three planes of the input tile ([plane][position][16 bf16]) and of the weight slab ([plane][tap][cout][16 bf16]); weights
// are split once at pack time, activations while they are staged into LDS.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/bfsr_hip.h"
#include "launch_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef f32x16 f32x16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int CK = 16;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// Split arithmetic of the kernels below (template parameter PL = planes per operand):
//   PL = 3: exact three-term bf16 split, six products hl+lh+mm+mh+hm+hh on v_mfma_f32_32x32x16_bf16 (BFSR_SPLIT=bf16x3)
//   PL = 2: two-term fp16 split (22 significant bits), three products lh+hl+hh on v_mfma_f32_32x32x16_f16; the weights were
//           multiplied by a power of two at pack time (their lo terms stay normal fp16 numbers) and the accumulators are
//           multiplied by p.acc_scale = 1/that first thing in the epilogue; activations must stay below 65504 in magnitude.
//           Half the matrix instructions and 2/3 of the LDS bytes; end to end indistinguishable from fp32.
template <int PL> struct Sp;
template <> struct Sp<3> {
    typedef __bf16 elt;
    typedef __bf16 frag __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x16_t mfma(frag a, frag b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Sp<2> {
    typedef _Float16 elt;
    typedef _Float16 frag __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x16_t mfma(frag a, frag b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
// the eight values of one (position, k half) unit -> PL fragments
// amax (PL = 2 only): running max |x| of everything this thread hands to the fp16 split -- the range guard of that split: a value
// >= 65504 becomes inf in its hi plane.  (A NaN input does not raise amax -- v_max ignores it -- but it stays NaN through the conv.)
template <int PL>
__device__ __forceinline__ void split_unit(const float (&v)[8], typename Sp<PL>::frag (&o)[PL], float& amax)
{
    if constexpr (PL == 2) {
#pragma unroll
        for (int c = 0; c < 8; c += 2) amax = fmaxf(amax, fmaxf(fabsf(v[c]), fabsf(v[c + 1])));
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float r = v[c];
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) {
            if constexpr (PL == 2) {                                  // fp16 pair: the rounded value pinned (launch_util.h, pin_f16)
                const float hf = bfsr::pin_f16(r);
                o[pl][c] = (typename Sp<PL>::elt)hf;
                r -= hf;
            } else {
                const typename Sp<PL>::elt h = (typename Sp<PL>::elt)r;   // round to nearest; the residual below is exact
                o[pl][c] = h;
                r -= (float)h;
            }
        }
    }
}
// products, smallest terms first: T_(plane of A = weights, plane of B = activations)
#define BFSR_PRODUCTS(PL_, T_) if constexpr ((PL_) == 3) { T_(2, 0) T_(0, 2) T_(1, 1) T_(1, 0) T_(0, 1) T_(0, 0) } else { T_(1, 0) T_(0, 1) T_(0, 0) }

__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l)
{
    h = (__bf16)v;
    const float r1 = v - (float)h;        // exact
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);          // exact residual, <= 8 significant bits
}

template <int PL, int KS, int MR, int NR, int MINB, int NW>
__global__ __launch_bounds__(NW * 64, MINB) void conv_bf16x3_kernel(BfsrConvArgs p, int tiles_x, int tiles_xy, int groups)
{
    constexpr int NT = NW * 64;
    constexpr int TH = NW * NR, TW = 32, HALO = KS - 1;
    constexpr int IH = TH + HALO, PW = TW + HALO, NPOS = IH * PW, PPT = (NPOS + NT - 1) / NT;
    constexpr int TAPS = KS * KS, MW = MR * 32;
    constexpr int WPL = TAPS * MW * CK;              // bf16 elements of one weight plane per chunk
    constexpr int WSLAB = PL * WPL;
    constexpr int WV = (WSLAB / 8 + NT - 1) / NT;      // 16-byte weight loads per thread
    constexpr int IPL = NPOS * CK;                   // bf16 elements of one input plane

    float amax = 0.f;                                                    // range guard of the fp16 split (split_unit)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    typedef typename Sp<PL>::elt elt;
    typedef typename Sp<PL>::frag frag;
    elt* sW = reinterpret_cast<elt*>(smem_raw);                     // [3][TAPS][k half][MW][8]
    elt* sIn = sW + WSLAB;                                             // [3][k half][NPOS][8]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    int bid = blockIdx.x;
    const int cg = bid % groups; bid /= groups;
    const int tile = bid % tiles_xy; const int b = bid / tiles_xy;
    const int x0 = (tile % tiles_x) * TW, y0 = (tile / tiles_x) * TH;

    const int H = p.H, W = p.W, sh = p.in_shift, Ws = W >> sh;
    const long long cs_in = (long long)(H >> sh) * Ws;
    const float* __restrict__ xin = p.x + (long long)b * p.x_bs;
    const int Cin = p.Cin;
    const int nchunk = (Cin + CK - 1) / CK;
    const elt* __restrict__ wg = reinterpret_cast<const elt*>(p.w) + (long long)cg * nchunk * WSLAB;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0,
                                                                           (unsigned)((long long)Cin * cs_in * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<elt*>(wg), 0,
                                                                          (unsigned)((long long)nchunk * WSLAB * 2), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned voff[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int pos = tid + i * NT;
        const int r = pos / PW, c = pos - r * PW;
        const int gy = y0 + r - HALO / 2, gx = x0 + c - HALO / 2;
        const bool ok = (pos < NPOS) && gy >= 0 && gy < H && gx >= 0 && gx < W;
        voff[i] = ok ? (unsigned)((gy >> sh) * Ws + (gx >> sh)) * 4u : OOB;
    }
    const unsigned cs_bytes = (unsigned)(cs_in * 4);

    f32x16 acc[MR][NR];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    float vin[PPT][CK];
    uint4 vw[WV];
    auto load_chunk = [&](int k) {
        const unsigned sbase = (unsigned)(k * CK) * cs_bytes;
#pragma unroll
        for (int c = 0; c < CK; ++c)
#pragma unroll
            for (int i = 0; i < PPT; ++i)
                vin[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, voff[i], sbase + (unsigned)c * cs_bytes, 0));
        const unsigned wbase = (unsigned)k * (WSLAB * 2);
#pragma unroll
        for (int i = 0; i < WV; ++i)
            vw[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(tid + i * NT) * 16u, wbase, 0));
    };
    load_chunk(0);

    for (int k = 0; k < nchunk; ++k) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int pos = tid + i * NT;
            if (i < PPT - 1 || pos < NPOS) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    float u8[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) u8[c] = vin[i][hf * 8 + c];
                    frag s8[PL];
                    split_unit<PL>(u8, s8, amax);
#pragma unroll
                    for (int pl = 0; pl < PL; ++pl) *reinterpret_cast<frag*>(sIn + pl * IPL + (hf * NPOS + pos) * 8) = s8[pl];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int idx = tid + i * NT;
            if (i < WV - 1 || idx < WSLAB / 8) reinterpret_cast<uint4*>(sW)[idx] = vw[i];
        }
        __syncthreads();
        if (k + 1 < nchunk) load_chunk(k + 1);
        const elt* inB = sIn + (lhi * NPOS + (wave * NR) * PW + l31) * 8;
        const elt* wA = sW + (lhi * MW + l31) * 8;
        // software pipeline over the taps (dx-major so a B row set serves the KS vertical taps): the A fragments of tap t+1
        // and, at a column change, the B fragments of column dx+1 are requested from LDS before the MFMAs of tap t issue
        constexpr bool PFB = (MINB == 1 && NW == 4);     // room for a second B row set only with the 512-register budget
        frag bfr[PFB ? 2 : 1][PL][NR + HALO], afr[2][PL][MR];
        auto load_b = [&](int buf, int dx) {
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int r = 0; r < NR + HALO; ++r)
                    bfr[buf][pl][r] = *reinterpret_cast<const frag*>(inB + pl * IPL + (r * PW + dx) * 8);
        };
        auto load_a = [&](int buf, int dx, int dy) {
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int m = 0; m < MR; ++m)
                    afr[buf][pl][m] = *reinterpret_cast<const frag*>(wA + pl * WPL + ((dy * KS + dx) * 2 * MW + m * 32) * 8);
        };
        load_b(0, 0);
        load_a(0, 0, 0);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int dx = t / KS, dy = t % KS;
            const int ab = t & 1, bb = PFB ? (dx & 1) : 0;
            if (!PFB && t > 0 && dy == 0) load_b(0, dx);
            if (t + 1 < TAPS) {
                load_a(ab ^ 1, (t + 1) / KS, (t + 1) % KS);
                if (PFB && dy == KS - 1) load_b(bb ^ 1, dx + 1);
            }
            __builtin_amdgcn_sched_barrier(0);          // keep the prefetch ahead of this tap's MFMAs
            // small terms first, the leading term last; MR*NR independent accumulators between dependent MFMAs
#define BFSR_TERM(PA_, PB_)                                                                                            \
    _Pragma("unroll") for (int m = 0; m < MR; ++m) _Pragma("unroll") for (int n = 0; n < NR; ++n)                       \
        acc[m][n] = Sp<PL>::mfma(afr[ab][PA_][m], bfr[bb][PB_][n + dy], acc[m][n]);
            BFSR_PRODUCTS(PL, BFSR_TERM)
#undef BFSR_TERM
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    if (PL == 2 && p.flag && __any((int)!(amax < 65504.f))) { if ((threadIdx.x & 63) == 0) atomicOr(p.flag, 1u); }   // range guard of the fp16 split
    // ---- epilogue (fp32, identical stage order to conv_mfma_kernel)
    const long long HW = (long long)H * W;
    const int gx = x0 + l31;
    if (gx >= W) return;
    const float slope = p.act == BFSR_ACT_NONE ? 1.f : (p.act == BFSR_ACT_RELU ? 0.f : p.slope);
    const float4* __restrict__ epi = reinterpret_cast<const float4*>(p.epi);
    const unsigned out_bytes = (unsigned)((long long)p.Cout * HW * 4);
    auto tensor_rsrc = [&](const float* t, long long bs) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t ? t + (long long)b * bs : p.y), 0, t ? out_bytes : 0u, 0x00020000);
    };
    const bool tensors = p.pre_add || p.res1 || p.res2;
    auto run_epilogue = [&](auto with_tensors) {
        constexpr bool T = decltype(with_tensors)::value;
        const __amdgpu_buffer_rsrc_t rs_pre = tensor_rsrc(p.pre_add, p.pre_add_bs);
        const __amdgpu_buffer_rsrc_t rs_r1 = tensor_rsrc(p.res1, p.res1_bs);
        const __amdgpu_buffer_rsrc_t rs_r2 = tensor_rsrc(p.res2, p.res2_bs);
        const float a1 = p.res1 ? p.alpha1 : 1.f, a2 = p.res2 ? p.alpha2 : 1.f;
        if (p.y_fmt == 1) {
            // quad-major output (and pre_add) [Cout/4][H][W][4]: the four channels (r&3) of an accumulator group are one 16-byte access
            // (the private layout coupling_head / coupling_tail read, include/bfsr_hip.h); no residuals in this form (launcher)
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co0 = (cg * MR + m) * 32 + 8 * g + 4 * lhi;
                    if (co0 >= p.Cout) continue;
                    float4 q0[4]; float q1[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        q0[e] = make_float4(0.f, 0.f, 1.f, 0.f); q1[e] = 1.f;
                        if (epi) { q0[e] = epi[(co0 + e) * 2]; q1[e] = epi[(co0 + e) * 2 + 1].x; }
                    }
#pragma unroll
                    for (int n = 0; n < NR; ++n) {
                        const int gy = y0 + wave * NR + n;
                        if (gy >= H) continue;
                        const unsigned ob = (unsigned)((((long long)(co0 >> 2) * HW + (long long)gy * W + gx) * 16));
                        float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
                        if constexpr (T) if (p.pre_add) pv = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_pre, ob, 0, 0));
                        const float pa_[4] = {pv.x, pv.y, pv.z, pv.w};
                        float o4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = PL == 2 ? acc[m][n][4 * g + e] * p.acc_scale : acc[m][n][4 * g + e];
                            v += q0[e].x;
                            v += pa_[e];
                            v += q0[e].y; v *= q0[e].z; v += q0[e].w;
                            v = v > 0.f ? v : v * slope;
                            o4[e] = v * q1[e];
                        }
                        *reinterpret_cast<float4*>(reinterpret_cast<char*>(p.y + (long long)b * p.y_bs) + ob) = make_float4(o4[0], o4[1], o4[2], o4[3]);
                    }
                }
            return;
        }
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (cg * MR + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (co >= p.Cout) continue;
                float4 q0 = make_float4(0.f, 0.f, 1.f, 0.f); float q1 = 1.f;
                if (epi) { q0 = epi[co * 2]; q1 = epi[co * 2 + 1].x; }
                const long long cbase = (long long)co * HW;
#pragma unroll
                for (int n = 0; n < NR; ++n) {
                    const int gy = y0 + wave * NR + n;
                    if (gy >= H) continue;
                    const long long o = cbase + (long long)gy * W + gx;
                    float v = PL == 2 ? acc[m][n][r] * p.acc_scale : acc[m][n][r];
                    v += q0.x;
                    if constexpr (T) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_pre, (unsigned)o * 4u, 0, 0));
                    v += q0.y; v *= q0.z; v += q0.w;
                    v = v > 0.f ? v : v * slope;
                    v *= q1;
                    if constexpr (T) {
                        v = a1 * v + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r1, (unsigned)o * 4u, 0, 0));
                        v = a2 * v + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r2, (unsigned)o * 4u, 0, 0));
                    }
                    p.y[(long long)b * p.y_bs + o] = v;
                }
            }
    };
    if (tensors) run_epilogue(std::true_type{});
    else run_epilogue(std::false_type{});
}

template <int PL, int KS, int MR, int NR, int MINB, int NW>
int launch_x3p(const BfsrConvArgs& a, hipStream_t st)
{
    constexpr int TH = NW * NR, HALO = KS - 1;
    constexpr int LDS = PL * (KS * KS * MR * 32 * CK + (TH + HALO) * (32 + HALO) * CK) * 2;
    static std::atomic<unsigned long long> lds_done{0};
    if (LDS > 65536 && bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_bf16x3_kernel<PL, KS, MR, NR, MINB, NW>), LDS, lds_done) != 0)
        return -1;
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + TH - 1) / TH;
    const int groups = ((a.Cout + 31) / 32 + MR - 1) / MR;
    const long long nblk = (long long)tiles_x * tiles_y * groups * a.B;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return -1;
    hipLaunchKernelGGL((conv_bf16x3_kernel<PL, KS, MR, NR, MINB, NW>), dim3((unsigned)nblk), dim3(NW * 64), LDS, st, a, tiles_x, tiles_x * tiles_y, groups);
    return (int)hipGetLastError();
}

template <int KS, int MR, int NR, int MINB, int NW>
int launch_x3(const BfsrConvArgs& a, hipStream_t st)
{
    return a.arith == 1 ? launch_x3p<2, KS, MR, NR, MINB, NW>(a, st) : launch_x3p<3, KS, MR, NR, MINB, NW>(a, st);
}


// ---- conv over nearest_up2(x) with the 16 parity-pre-summed matrices (see conv_up2_kernel in conv_mfma.hip), 3xBF16 split.
// x [B,Cin,H/2,W/2] -> y [B,Cout,H,W]; tap index t = (a*2+b)*4 + i*2+j (output row/column parity a,b; source offset i,j).
// A workgroup produces the output rows of ONE row parity `a` (8 of the 16 matrices -> half the weight slab in LDS, two
// workgroups per CU): NW waves, source tile (NW*NR) rows x 32 columns, 32 output channels; each wave owns NR source rows:
// acc[column parity][row].  Column offset d = b+j is the outer loop so one set of B rows serves every (b,j) with that d.
// The two column parities of a source pixel are adjacent output pixels: the epilogue moves float2 (fully coalesced rows).
// Channels that already live at the output resolution go through the plain kernel first and arrive here as `pre_add`.
template <int PL, int NW, int NR, int MR>
__global__ __launch_bounds__(NW * 64, 2) void conv_up2_bf16x3_kernel(BfsrConvArgs p, int tiles_x, int tiles_xy, int groups)
{
    constexpr int NT = NW * 64, SR = NW * NR, PW = 34, NPOS = (SR + 1) * PW, PPT = (NPOS + NT - 1) / NT;
    constexpr int TAPS = 16, MW = 32 * MR, HT = 8;                       // HT = matrices of one row parity; MR = 32-cout M tiles per workgroup
    constexpr int WPL = HT * MW * CK, WSLAB = PL * WPL, WV = (WSLAB / 8 + NT - 1) / NT, IPL = NPOS * CK;
    constexpr int GPL = TAPS * MW * CK;                                  // one plane of all 16 matrices in global memory
    float amax = 0.f;                                                    // range guard of the fp16 split (split_unit)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    typedef typename Sp<PL>::elt elt;
    typedef typename Sp<PL>::frag frag;
    elt* sW = reinterpret_cast<elt*>(smem_raw);                     // [3][8][k half][32][8]
    elt* sIn = sW + WSLAB;                                             // [3][k half][NPOS][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    int bid = blockIdx.x;
    const int pa = bid & 1; bid >>= 1;                                    // output row parity of this workgroup
    const int cg = bid % groups; bid /= groups;
    const int tile = bid % tiles_xy; const int b = bid / tiles_xy;
    const int x0 = (tile % tiles_x) * 32, y0 = (tile / tiles_x) * SR;     // source coordinates
    const int H = p.H, W = p.W, Hs = H >> 1, Ws = W >> 1;
    const long long cs_in = (long long)Hs * Ws;
    const float* __restrict__ xin = p.x + (long long)b * p.x_bs;
    const int Cin = p.Cin, nchunk = (Cin + CK - 1) / CK;
    const elt* __restrict__ wg = reinterpret_cast<const elt*>(p.w) + (long long)cg * nchunk * PL * GPL;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0,
                                                                           (unsigned)((long long)Cin * cs_in * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<elt*>(wg), 0,
                                                                          (unsigned)((long long)nchunk * PL * GPL * 2), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned voff[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int pos = tid + i * NT;
        const int r = pos / PW, c = pos - r * PW;
        const int gy = y0 + r - 1 + pa, gx = x0 + c - 1;                  // parity a needs source rows sy-1+a, sy+a
        const bool ok = pos < NPOS && gy >= 0 && gy < Hs && gx >= 0 && gx < Ws;
        voff[i] = ok ? (unsigned)(gy * Ws + gx) * 4u : OOB;
    }
    // weight slab of this parity: per plane the 8 matrices t = pa*8 .. pa*8+7 are contiguous in global memory
    unsigned woff[WV];
#pragma unroll
    for (int i = 0; i < WV; ++i) {
        const int idx = tid + i * NT;                                     // 16-byte unit inside [3][8][32][16]
        const int pl = idx / (WPL / 8), rem = idx - pl * (WPL / 8);
        woff[i] = idx < WSLAB / 8 ? (unsigned)((pl * GPL + pa * WPL) * 2 + rem * 16) : OOB;
    }
    const unsigned cs_bytes = (unsigned)(cs_in * 4);
    f32x16 acc[MR][2][NR];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int n = 0; n < NR; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][q][n][r] = 0.f;

    float vin[PPT][CK];
    uint4 vw[WV];
    auto load_chunk = [&](int k) {
        const unsigned sbase = (unsigned)(k * CK) * cs_bytes;
#pragma unroll
        for (int c = 0; c < CK; ++c)
#pragma unroll
            for (int i = 0; i < PPT; ++i)
                vin[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, voff[i], sbase + (unsigned)c * cs_bytes, 0));
        const unsigned wbase = (unsigned)k * (PL * GPL * 2);
#pragma unroll
        for (int i = 0; i < WV; ++i)
            vw[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woff[i], wbase, 0));
    };
    load_chunk(0);
    for (int k = 0; k < nchunk; ++k) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int pos = tid + i * NT;
            if (i < PPT - 1 || pos < NPOS) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    float u8[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) u8[c] = vin[i][hf * 8 + c];
                    frag s8[PL];
                    split_unit<PL>(u8, s8, amax);
#pragma unroll
                    for (int pl = 0; pl < PL; ++pl) *reinterpret_cast<frag*>(sIn + pl * IPL + (hf * NPOS + pos) * 8) = s8[pl];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int idx = tid + i * NT;
            if (i < WV - 1 || idx < WSLAB / 8) reinterpret_cast<uint4*>(sW)[idx] = vw[i];
        }
        __syncthreads();
        if (k + 1 < nchunk) load_chunk(k + 1);
        const elt* inB = sIn + (lhi * NPOS + (wave * NR) * PW + l31) * 8;
        const elt* wA = sW + (lhi * MW + l31) * 8;
        frag bfr[PL][NR + 1], afr[2][PL][MR];
        auto load_b = [&](int d) {
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int e = 0; e < NR + 1; ++e)
                    bfr[pl][e] = *reinterpret_cast<const frag*>(inB + pl * IPL + (e * PW + d) * 8);
        };
        // step s = 0..7 in the order (d; bq,j with bq+j = d; i); local matrix index = bq*4 + i*2 + j
        auto load_a = [&](int buf, int s_) {
            const int blk = s_ >> 1, bq = blk >> 1, j = blk & 1, i = s_ & 1;
            const int t = bq * 4 + i * 2 + j;
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int m = 0; m < MR; ++m) afr[buf][pl][m] = *reinterpret_cast<const frag*>(wA + pl * WPL + (t * 2 * MW + m * 32) * 8);
        };
        load_a(0, 0);
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            const int blk = s_ >> 1, bq = blk >> 1, j = blk & 1, i = s_ & 1, d = bq + j;
            const int ab = s_ & 1;
            if (s_ == 0 || s_ == 2 || s_ == 6) load_b(d);
            if (s_ + 1 < 8) load_a(ab ^ 1, s_ + 1);
            __builtin_amdgcn_sched_barrier(0);
#define BFSR_TERM(PA_, PB_)                                                                                            \
    _Pragma("unroll") for (int m = 0; m < MR; ++m) _Pragma("unroll") for (int n = 0; n < NR; ++n)                       \
        acc[m][bq][n] = Sp<PL>::mfma(afr[ab][PA_][m], bfr[PB_][n + i], acc[m][bq][n]);
            BFSR_PRODUCTS(PL, BFSR_TERM)
#undef BFSR_TERM
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    if (PL == 2 && p.flag && __any((int)!(amax < 65504.f))) { if ((threadIdx.x & 63) == 0) atomicOr(p.flag, 1u); }   // range guard of the fp16 split
    // ---- epilogue (same stage order as conv_mfma_kernel); lane = source column -> two adjacent output pixels
    const int sx = x0 + l31;
    if (sx >= Ws) return;
    const long long HW = (long long)H * W;
    const float slope = p.act == BFSR_ACT_NONE ? 1.f : (p.act == BFSR_ACT_RELU ? 0.f : p.slope);
    const float4* __restrict__ epi = reinterpret_cast<const float4*>(p.epi);
    const bool tensors = p.pre_add || p.res1 || p.res2;
    auto run_epilogue = [&](auto with_tensors) {
        constexpr bool T = decltype(with_tensors)::value;
        const float a1 = p.res1 ? p.alpha1 : 1.f, a2 = p.res2 ? p.alpha2 : 1.f;
        const float* pre = p.pre_add ? p.pre_add + (long long)b * p.pre_add_bs : nullptr;
        const float* r1 = p.res1 ? p.res1 + (long long)b * p.res1_bs : nullptr;
        const float* r2 = p.res2 ? p.res2 + (long long)b * p.res2_bs : nullptr;
        float* yb = p.y + (long long)b * p.y_bs;
        if (p.y_fmt == 1) {
            // quad-major output (and pre_add) [Cout/4][H][W][4]: a lane's two adjacent output pixels x the four channels of an accumulator
            // group are 32 contiguous bytes (two 16-byte accesses instead of four 8-byte ones); no residuals in this form (launcher)
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co0 = (cg * MR + m) * 32 + 8 * g + 4 * lhi;
                if (co0 >= p.Cout) continue;
                float4 q0[4]; float q1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    q0[e] = make_float4(0.f, 0.f, 1.f, 0.f); q1[e] = 1.f;
                    if (epi) { q0[e] = epi[(co0 + e) * 2]; q1[e] = epi[(co0 + e) * 2 + 1].x; }
                }
#pragma unroll
                for (int n = 0; n < NR; ++n) {
                    const int sy = y0 + wave * NR + n;
                    if (sy >= Hs) continue;
                    const long long o = ((long long)(co0 >> 2) * HW + (long long)(2 * sy + pa) * W + 2 * sx) * 4;
                    float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;
                    if constexpr (T) if (pre) { t0 = *reinterpret_cast<const float4*>(pre + o); t1 = *reinterpret_cast<const float4*>(pre + o + 4); }
                    const float pa0[4] = {t0.x, t0.y, t0.z, t0.w}, pa1[4] = {t1.x, t1.y, t1.z, t1.w};
                    float v0[4], v1[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = acc[m][0][n][4 * g + e], c = acc[m][1][n][4 * g + e];
                        if (PL == 2) { a *= p.acc_scale; c *= p.acc_scale; }
                        a += q0[e].x; c += q0[e].x;
                        a += pa0[e]; c += pa1[e];
                        a = (a + q0[e].y) * q0[e].z + q0[e].w; c = (c + q0[e].y) * q0[e].z + q0[e].w;
                        a = a > 0.f ? a : a * slope; c = c > 0.f ? c : c * slope;
                        v0[e] = a * q1[e]; v1[e] = c * q1[e];
                    }
                    *reinterpret_cast<float4*>(yb + o) = make_float4(v0[0], v0[1], v0[2], v0[3]);
                    *reinterpret_cast<float4*>(yb + o + 4) = make_float4(v1[0], v1[1], v1[2], v1[3]);
                }
            }
            return;
        }
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (cg * MR + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (co >= p.Cout) continue;
            float4 q0 = make_float4(0.f, 0.f, 1.f, 0.f); float q1 = 1.f;
            if (epi) { q0 = epi[co * 2]; q1 = epi[co * 2 + 1].x; }
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const int sy = y0 + wave * NR + n;
                if (sy >= Hs) continue;
                const long long o = (long long)co * HW + (long long)(2 * sy + pa) * W + 2 * sx;
                float2 v = make_float2(acc[m][0][n][r], acc[m][1][n][r]);
                if (PL == 2) { v.x *= p.acc_scale; v.y *= p.acc_scale; }
                v.x += q0.x; v.y += q0.x;
                if constexpr (T) if (pre) { const float2 t = *reinterpret_cast<const float2*>(pre + o); v.x += t.x; v.y += t.y; }
                v.x = (v.x + q0.y) * q0.z + q0.w; v.y = (v.y + q0.y) * q0.z + q0.w;
                v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
                v.x *= q1; v.y *= q1;
                if constexpr (T) {
                    if (r1) { const float2 t = *reinterpret_cast<const float2*>(r1 + o); v.x = a1 * v.x + t.x; v.y = a1 * v.y + t.y; }
                    if (r2) { const float2 t = *reinterpret_cast<const float2*>(r2 + o); v.x = a2 * v.x + t.x; v.y = a2 * v.y + t.y; }
                }
                *reinterpret_cast<float2*>(yb + o) = v;
            }
        }
    };
    if (tensors) run_epilogue(std::true_type{});
    else run_epilogue(std::false_type{});
}

template <int PL, int NW, int NR, int MR>
int launch_up2_x3p(const BfsrConvArgs& a, hipStream_t st)
{
    constexpr int SR = NW * NR;
    constexpr int LDS = PL * (8 * 32 * MR * CK + (SR + 1) * 34 * CK) * 2;
    static std::atomic<unsigned long long> lds_done{0};
    if (LDS > 65536 && bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_up2_bf16x3_kernel<PL, NW, NR, MR>), LDS, lds_done) != 0) return -1;
    const int Hs = a.H / 2, Ws = a.W / 2;
    const int tiles_x = (Ws + 31) / 32, tiles_y = (Hs + SR - 1) / SR;
    const int groups = (a.Cout + 32 * MR - 1) / (32 * MR);
    const long long nblk = 2LL * tiles_x * tiles_y * groups * a.B;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return -1;
    hipLaunchKernelGGL((conv_up2_bf16x3_kernel<PL, NW, NR, MR>), dim3((unsigned)nblk), dim3(NW * 64), LDS, st, a, tiles_x, tiles_x * tiles_y, groups);
    return (int)hipGetLastError();
}

template <int NW, int NR>
int launch_up2_x3(const BfsrConvArgs& a, hipStream_t st)
{
    if (a.mtile == 2) return a.arith == 1 ? launch_up2_x3p<2, NW, NR, 2>(a, st) : -1;      // 64-cout workgroups: the fp16 pair only
    return a.arith == 1 ? launch_up2_x3p<2, NW, NR, 1>(a, st) : launch_up2_x3p<3, NW, NR, 1>(a, st);
}

// ---- conv over nearest_up4(x): x [B,Cin,H/4,W/4] -> y [B,Cout,H,W] (the level-1 conditional of the 8x model: LR-resolution
// RRDB taps under a 4x finer flow level).  Along each axis the 3x3 window of output phase p = 0..3 touches the source offsets
//   p=0: {-1 <- w[-1], 0 <- w[0]+w[+1]}     p=1,2: {0 <- w[-1]+w[0]+w[+1]}     p=3: {0 <- w[-1]+w[0], +1 <- w[+1]}
// i.e. 5 pre-summed "entries" per axis in 3 classes (phases 1 and 2 see the same window): 25 matrices produce the 9 distinct
// outputs of a source pixel (25 vs 144 tap products of the materialised conv), which the epilogue replicates to the 16 pixels.
// entry e = 0..4 -> (class, staged offset): (0,0) (0,1) (1,1) (2,1) (2,2) with staged offset 0,1,2 = source offset -1,0,+1.
// A workgroup handles one ROW class rc (10, 5 or 10 matrices in LDS; staged rows start at y0-1 for rc=0, y0 otherwise, so row
// entry ie of the class reads staged row n+ie) and keeps acc[column class][row]; lane = source column -> float4 of 4 output px.
template <int PL, int NW, int NR, int RC>
__global__ __launch_bounds__(NW * 64, 4) void conv_up4_bf16x3_kernel(BfsrConvArgs p, int tiles_x, int tiles_xy, int groups)
{
    constexpr int NT = NW * 64, SR = NW * NR, PW = 34, NPOS = (SR + 1) * PW, PPT = (NPOS + NT - 1) / NT;
    constexpr int MW = 32, MAXM = 10, MEL = 2 * MW * 8;                   // MEL = bf16 elements of one matrix chunk
    constexpr int WPL = MAXM * MEL, WSLAB = PL * WPL, WV = (WSLAB / 8 + NT - 1) / NT, IPL = NPOS * CK;
    constexpr int GPL = 25 * MEL;                                         // one plane of all 25 matrices in global memory
    float amax = 0.f;                                                    // range guard of the fp16 split (split_unit)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    typedef typename Sp<PL>::elt elt;
    typedef typename Sp<PL>::frag frag;
    elt* sW = reinterpret_cast<elt*>(smem_raw);                     // [3][<=10][k half][32][8]
    elt* sIn = sW + WSLAB;                                             // [3][k half][NPOS][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    int bid = blockIdx.x;
    constexpr int rc = RC;                                                // row class of this launch
    const int cg = bid % groups; bid /= groups;
    const int tile = bid % tiles_xy; const int b = bid / tiles_xy;
    const int x0 = (tile % tiles_x) * 32, y0 = (tile / tiles_x) * SR;     // source coordinates
    const int H = p.H, W = p.W, Hs = H >> 2, Ws = W >> 2;
    constexpr int nrt = rc == 1 ? 1 : 2, e0 = rc == 0 ? 0 : (rc == 1 ? 2 : 3), nm = nrt * 5;
    const long long cs_in = (long long)Hs * Ws;
    const float* __restrict__ xin = p.x + (long long)b * p.x_bs;
    const int Cin = p.Cin, nchunk = (Cin + CK - 1) / CK;
    const elt* __restrict__ wg = reinterpret_cast<const elt*>(p.w) + (long long)cg * nchunk * PL * GPL;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0,
                                                                           (unsigned)((long long)Cin * cs_in * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<elt*>(wg), 0,
                                                                          (unsigned)((long long)nchunk * PL * GPL * 2), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned voff[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int pos = tid + i * NT;
        const int r = pos / PW, c = pos - r * PW;
        const int gy = y0 + r - (rc == 0 ? 1 : 0), gx = x0 + c - 1;
        const bool ok = pos < NPOS && gy >= 0 && gy < Hs && gx >= 0 && gx < Ws;
        voff[i] = ok ? (unsigned)(gy * Ws + gx) * 4u : OOB;
    }
    unsigned woff[WV];
#pragma unroll
    for (int i = 0; i < WV; ++i) {
        const int idx = tid + i * NT;                                     // 16-byte unit inside [3][MAXM][MEL]
        const int pl = idx / (WPL / 8), rem = idx - pl * (WPL / 8);
        woff[i] = (idx < WSLAB / 8 && rem < nm * (MEL / 8)) ? (unsigned)((pl * GPL + e0 * 5 * MEL) * 2 + rem * 16) : OOB;
    }
    const unsigned cs_bytes = (unsigned)(cs_in * 4);
    f32x16 acc[3][NR];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][n][r] = 0.f;

    float vin[PPT][CK];
    uint4 vw[WV];
    auto load_chunk = [&](int k) {
        const unsigned sbase = (unsigned)(k * CK) * cs_bytes;
#pragma unroll
        for (int c = 0; c < CK; ++c)
#pragma unroll
            for (int i = 0; i < PPT; ++i)
                vin[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, voff[i], sbase + (unsigned)c * cs_bytes, 0));
        const unsigned wbase = (unsigned)k * (PL * GPL * 2);
#pragma unroll
        for (int i = 0; i < WV; ++i)
            vw[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woff[i], wbase, 0));
    };
    // one chunk of MFMAs for a row class with NRT row entries; column entries in staged-offset order d = 0 | 1,1,1 | 2
    auto mfma_chunk = [&](auto nrt_tag) {
        constexpr int NRT = decltype(nrt_tag)::value;
        constexpr int STEPS = 5 * NRT;
        const elt* inB = sIn + (lhi * NPOS + (wave * NR) * PW + l31) * 8;
        const elt* wA = sW + (lhi * MW + l31) * 8;
        frag bfr[PL][NR + 1], afr[2][PL];
        auto load_b = [&](int d) {
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int e = 0; e < NR + NRT - 1; ++e)
                    bfr[pl][e] = *reinterpret_cast<const frag*>(inB + pl * IPL + (e * PW + d) * 8);
        };
        auto load_a = [&](int buf, int s_) {                              // step s_ -> (column entry ce, row entry ie)
            const int ce = s_ / NRT, ie = s_ % NRT;
            const int t = ie * 5 + ce;
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) afr[buf][pl] = *reinterpret_cast<const frag*>(wA + pl * WPL + t * MEL);
        };
        load_a(0, 0);
#pragma unroll
        for (int s_ = 0; s_ < STEPS; ++s_) {
            const int ce = s_ / NRT, ie = s_ % NRT;
            const int d = ce == 0 ? 0 : (ce == 4 ? 2 : 1), cc = ce == 0 ? 0 : (ce <= 1 ? 0 : (ce == 2 ? 1 : 2));
            const int ab = s_ & 1;
            if (ie == 0 && (ce == 0 || ce == 1 || ce == 4)) load_b(d);
            if (s_ + 1 < STEPS) load_a(ab ^ 1, s_ + 1);
            __builtin_amdgcn_sched_barrier(0);
#define BFSR_TERM(PA_, PB_)                                                                                            \
    _Pragma("unroll") for (int n = 0; n < NR; ++n)                                                                      \
        acc[cc][n] = Sp<PL>::mfma(afr[ab][PA_], bfr[PB_][n + ie], acc[cc][n]);
            BFSR_PRODUCTS(PL, BFSR_TERM)
#undef BFSR_TERM
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    load_chunk(0);
    for (int k = 0; k < nchunk; ++k) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int pos = tid + i * NT;
            if (i < PPT - 1 || pos < NPOS) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    float u8[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) u8[c] = vin[i][hf * 8 + c];
                    frag s8[PL];
                    split_unit<PL>(u8, s8, amax);
#pragma unroll
                    for (int pl = 0; pl < PL; ++pl) *reinterpret_cast<frag*>(sIn + pl * IPL + (hf * NPOS + pos) * 8) = s8[pl];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int idx = tid + i * NT;
            if (i < WV - 1 || idx < WSLAB / 8) reinterpret_cast<uint4*>(sW)[idx] = vw[i];
        }
        __syncthreads();
        if (k + 1 < nchunk) load_chunk(k + 1);
        mfma_chunk(std::integral_constant<int, nrt>{});
    }

    if (PL == 2 && p.flag && __any((int)!(amax < 65504.f))) { if ((threadIdx.x & 63) == 0) atomicOr(p.flag, 1u); }   // range guard of the fp16 split
    // ---- epilogue: lane = source column -> float4 of output columns 4sx..4sx+3 = classes (0, 1, 1, 2); rows of this class
    const int sx = x0 + l31;
    if (sx >= Ws) return;
    const long long HW = (long long)H * W;
    const float slope = p.act == BFSR_ACT_NONE ? 1.f : (p.act == BFSR_ACT_RELU ? 0.f : p.slope);
    const float4* __restrict__ epi = reinterpret_cast<const float4*>(p.epi);
    const float* pre = p.pre_add ? p.pre_add + (long long)b * p.pre_add_bs : nullptr;
    float* yb = p.y + (long long)b * p.y_bs;
    constexpr int row0 = rc == 0 ? 0 : (rc == 1 ? 1 : 3), nrow = rc == 1 ? 2 : 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (co >= p.Cout) continue;
        float4 q0 = make_float4(0.f, 0.f, 1.f, 0.f); float q1 = 1.f;
        if (epi) { q0 = epi[co * 2]; q1 = epi[co * 2 + 1].x; }
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            const int sy = y0 + wave * NR + n;
            if (sy >= Hs) continue;
            for (int rr = 0; rr < nrow; ++rr) {
                const long long o = (long long)co * HW + (long long)(4 * sy + row0 + rr) * W + 4 * sx;
                float v[4] = {acc[0][n][r], acc[1][n][r], acc[1][n][r], acc[2][n][r]};
                if (PL == 2) { v[0] *= p.acc_scale; v[1] *= p.acc_scale; v[2] *= p.acc_scale; v[3] *= p.acc_scale; }
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pre) t = *reinterpret_cast<const float4*>(pre + o);
                const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float u = v[j] + q0.x + tv[j];
                    u = (u + q0.y) * q0.z + q0.w;
                    u = u > 0.f ? u : u * slope;
                    v[j] = u * q1;
                }
                *reinterpret_cast<float4*>(yb + o) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

template <int PL, int NW, int NR>
int launch_up4_x3p(const BfsrConvArgs& a, hipStream_t st)
{
    constexpr int SR = NW * NR;
    constexpr int LDS = PL * (10 * 2 * 32 * 8 + (SR + 1) * 34 * CK) * 2;
    const int Hs = a.H / 4, Ws = a.W / 4;
    const int tiles_x = (Ws + 31) / 32, tiles_y = (Hs + SR - 1) / SR;
    const int groups = (a.Cout + 31) / 32;
    const long long nblk = (long long)tiles_x * tiles_y * groups * a.B;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return -1;
    // one launch per row class (phase 0 | phases 1,2 | phase 3): they write disjoint output rows
    hipLaunchKernelGGL((conv_up4_bf16x3_kernel<PL, NW, NR, 0>), dim3((unsigned)nblk), dim3(NW * 64), LDS, st, a, tiles_x, tiles_x * tiles_y, groups);
    hipLaunchKernelGGL((conv_up4_bf16x3_kernel<PL, NW, NR, 1>), dim3((unsigned)nblk), dim3(NW * 64), LDS, st, a, tiles_x, tiles_x * tiles_y, groups);
    hipLaunchKernelGGL((conv_up4_bf16x3_kernel<PL, NW, NR, 2>), dim3((unsigned)nblk), dim3(NW * 64), LDS, st, a, tiles_x, tiles_x * tiles_y, groups);
    return (int)hipGetLastError();
}

template <int NW, int NR>
int launch_up4_x3(const BfsrConvArgs& a, hipStream_t st)
{
    return a.arith == 1 ? launch_up4_x3p<2, NW, NR>(a, st) : launch_up4_x3p<3, NW, NR>(a, st);
}

inline void split3_host(float v, unsigned short out[3])
{
    float r = v;
    for (int i = 0; i < 3; ++i) {
        const __bf16 h = (__bf16)r;               // round-to-nearest-even, the conversion the kernel applies to activations
        __builtin_memcpy(&out[i], &h, 2);
        r -= (float)h;
    }
}

}  // namespace

extern "C" long long bfsr_conv_packed_size_taps_bf16x3(int Cout, int Cin, int T, int mtile)
{
    const int nchunk = (Cin + CK - 1) / CK;
    const int groups = ((Cout + 31) / 32 + mtile - 1) / mtile;
    return (long long)groups * nchunk * 3 * T * mtile * 32 * CK;            // number of bf16 elements
}

extern "C" int bfsr_pack_conv_weight_taps_bf16x3(const float* w, int Cout, int Cin, int T, int mtile, unsigned short* packed)
{
    // w [Cout][Cin][T] fp32 -> bf16 [cout_group][chunk][plane h,m,l][tap][k half][mtile*32][8], zero padded
    // (k-half-major so the 64 lanes of an MFMA operand read 2 x 512 contiguous bytes of LDS: no bank conflicts)
    if (T < 1 || mtile < 1) return -1;
    const int nchunk = (Cin + CK - 1) / CK, MW = mtile * 32;
    const int groups = ((Cout + 31) / 32 + mtile - 1) / mtile;
    const long long n = (long long)groups * nchunk * 3 * T * MW * CK;
    for (long long i = 0; i < n; ++i) packed[i] = 0;
    for (int co = 0; co < Cout; ++co) {
        const int g = co / MW, m = co % MW;
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < T; ++t) {
                unsigned short s3[3];
                split3_host(w[((long long)co * Cin + ci) * T + t], s3);
                for (int pl = 0; pl < 3; ++pl)
                    packed[((((((long long)g * nchunk + ci / CK) * 3 + pl) * T + t) * 2 + (ci % CK) / 8) * MW + m) * 8 + ci % 8] = s3[pl];
            }
    }
    return 0;
}

extern "C" long long bfsr_conv_packed_size_taps_f16x2(int Cout, int Cin, int T, int mtile)
{
    const int nchunk = (Cin + CK - 1) / CK;
    const int groups = ((Cout + 31) / 32 + mtile - 1) / mtile;
    return (long long)groups * nchunk * 2 * T * mtile * 32 * CK;            // number of fp16 elements
}

extern "C" int bfsr_pack_conv_weight_taps_f16x2(const float* w, int Cout, int Cin, int T, int mtile, float scale, unsigned short* packed)
{
    // w [Cout][Cin][T] fp32 -> fp16 [cout_group][chunk][plane hi,lo][tap][k half][mtile*32][8] of w*scale (scale: a power of two
    // chosen by the caller, see bfsr_hip.h), zero padded -- the layout of bfsr_pack_conv_weight_taps_bf16x3 with two planes
    if (T < 1 || mtile < 1 || !(scale > 0.f)) return -1;
    const int nchunk = (Cin + CK - 1) / CK, MW = mtile * 32;
    const int groups = ((Cout + 31) / 32 + mtile - 1) / mtile;
    const long long n = (long long)groups * nchunk * 2 * T * MW * CK;
    for (long long i = 0; i < n; ++i) packed[i] = 0;
    for (int co = 0; co < Cout; ++co) {
        const int g = co / MW, m = co % MW;
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < T; ++t) {
                float r = w[((long long)co * Cin + ci) * T + t] * scale;
                for (int pl = 0; pl < 2; ++pl) {
                    const _Float16 h = (_Float16)r;
                    unsigned short bits;
                    __builtin_memcpy(&bits, &h, 2);
                    packed[((((((long long)g * nchunk + ci / CK) * 2 + pl) * T + t) * 2 + (ci % CK) / 8) * MW + m) * 8 + ci % 8] = bits;
                    r -= (float)h;
                }
            }
    }
    return 0;
}

extern "C" long long bfsr_conv_packed_size_bf16x3(int Cout, int Cin, int KS, int mtile)
{
    return bfsr_conv_packed_size_taps_bf16x3(Cout, Cin, KS * KS, mtile);
}

extern "C" int bfsr_pack_conv_weight_bf16x3(const float* w, int Cout, int Cin, int KS, int mtile, unsigned short* packed)
{
    if (KS != 1 && KS != 3) return -1;
    return bfsr_pack_conv_weight_taps_bf16x3(w, Cout, Cin, KS * KS, mtile, packed);
}

extern "C" int bfsr_conv2d_up2_bf16x3(const BfsrConvArgs* a, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || !a->x || !a->w || !a->y || a->w2 || a->x2 || (a->mtile != 1 && !(a->mtile == 2 && a->arith == 1))) return -1;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || (a->H & 1) || (a->W & 1) || a->Cin <= 0 || a->Cout <= 0) return -1;
    if ((long long)a->Cin * (a->H / 2) * (a->W / 2) * 4 >= (1LL << 31)) return -1;
    // the float2 epilogue needs 8-byte aligned rows: even W and even plane/batch strides are given by H,W even + NCHW views
    if ((reinterpret_cast<unsigned long long>(a->y) & 7) || (a->y_bs & 1)) return -1;
    if (a->pre_add && ((reinterpret_cast<unsigned long long>(a->pre_add) & 7) || (a->pre_add_bs & 1))) return -1;
    if (a->res1 && ((reinterpret_cast<unsigned long long>(a->res1) & 7) || (a->res1_bs & 1))) return -1;
    if (a->res2 && ((reinterpret_cast<unsigned long long>(a->res2) & 7) || (a->res2_bs & 1))) return -1;
    if (a->y_fmt != 0 && a->y_fmt != 1) return -1;
    if (a->y_fmt == 1 && ((a->Cout & 3) || a->res1 || a->res2 || (reinterpret_cast<unsigned long long>(a->y) & 15) || (a->y_bs & 3) ||
                          (a->pre_add && ((reinterpret_cast<unsigned long long>(a->pre_add) & 15) || (a->pre_add_bs & 3))))) return -1;
    const int v = a->tune ? a->tune : 801;
    switch (v) {
        case 402: return launch_up2_x3<4, 2>(*a, st);
        case 401: return launch_up2_x3<4, 1>(*a, st);
        case 801: return launch_up2_x3<8, 1>(*a, st);
        default: return -1;
    }
}

extern "C" int bfsr_conv2d_up4_bf16x3(const BfsrConvArgs* a, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || !a->x || !a->w || !a->y || a->w2 || a->x2 || a->res1 || a->res2 || a->mtile != 1 || a->y_fmt != 0) return -1;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || (a->H & 3) || (a->W & 3) || a->Cin <= 0 || a->Cout <= 0) return -1;
    if ((long long)a->Cin * (a->H / 4) * (a->W / 4) * 4 >= (1LL << 31)) return -1;
    if ((reinterpret_cast<unsigned long long>(a->y) & 15) || (a->y_bs & 3)) return -1;           // float4 rows
    if (a->pre_add && ((reinterpret_cast<unsigned long long>(a->pre_add) & 15) || (a->pre_add_bs & 3))) return -1;
    return launch_up4_x3<8, 1>(*a, st);
}

extern "C" int bfsr_conv2d_bf16x3(const BfsrConvArgs* a, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || !a->x || !a->w || !a->y || a->w2 || a->x2) return -1;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->Cin <= 0 || a->Cout <= 0 || a->in_shift < 0 || a->in_shift > 4) return -1;
    if (a->in_shift && (((a->H >> a->in_shift) << a->in_shift) != a->H || ((a->W >> a->in_shift) << a->in_shift) != a->W)) return -1;
    if ((long long)a->Cin * (a->H >> a->in_shift) * (a->W >> a->in_shift) * 4 >= (1LL << 31)) return -1;
    if ((a->pre_add || a->res1 || a->res2) && (long long)a->Cout * a->H * a->W * 4 >= (1LL << 31)) return -1;
    if (a->y_fmt != 0 && a->y_fmt != 1) return -1;
    if (a->y_fmt == 1 && ((a->Cout & 3) || a->res1 || a->res2 || (long long)a->Cout * a->H * a->W * 4 >= (1LL << 31) ||
                          (reinterpret_cast<unsigned long long>(a->y) & 15) || (a->y_bs & 3) ||
                          (a->pre_add && ((reinterpret_cast<unsigned long long>(a->pre_add) & 15) || (a->pre_add_bs & 3))))) return -1;
    // 3x3: 8-wave workgroups (2 waves per SIMD hide each other's LDS latency); one tile row per wave for 32-cout layers
    // (2 workgroups per CU) and for small grids, two rows otherwise.  1x1: 4 waves x 4 rows.
    int NR, NW = 8;
    if (a->KS == 1) { NR = 4; NW = 4; }
    else if (a->mtile == 1) NR = 1;
    else {
        const long long groups_ = ((a->Cout + 31) / 32 + 1) / 2;
        NR = (long long)((a->W + 31) / 32) * ((a->H + 15) / 16) * a->B * groups_ >= 160 ? 2 : 1;
    }
    if (a->tune) { NR = a->tune / 100; NW = (a->tune % 100) ? a->tune % 100 : 4; }
    const int key = a->KS * 1000 + a->mtile * 100 + NR * 10 + NW;
    switch (key) {
#define V(KS_, MR_, NR_, MB_, NW_) case KS_ * 1000 + MR_ * 100 + NR_ * 10 + NW_: return launch_x3<KS_, MR_, NR_, MB_, NW_>(*a, st);
        V(3, 1, 2, 2, 4) V(3, 2, 2, 2, 4) V(3, 1, 4, 1, 4) V(3, 2, 4, 1, 4) V(1, 1, 4, 2, 4) V(1, 2, 4, 2, 4) V(1, 1, 2, 2, 4) V(1, 2, 2, 2, 4)
        V(3, 1, 2, 1, 8) V(3, 2, 2, 1, 8) V(3, 2, 1, 1, 8) V(3, 1, 1, 2, 8) V(3, 2, 1, 2, 4) V(3, 1, 1, 2, 4)
#undef V
        default: return -1;
    }
}
