// conv1x1.hip -- 1x1 convolutions with wide outputs (the LINF shared MLP, linf.py:313-314: 1024->256->256->256->540 per query
// point) as a plain GEMM  Y[Cout][P] = W[Cout][Cin] * X[Cin][P]  over the flattened pixel axis P = H*W.
// The general conv kernels give each workgroup 32-64 output channels and re-stage the activations once per channel group;
// a 1x1 has no halo and no tap reuse, so here a workgroup owns 128 pixels x 256 output channels: the four waves split the
// OUTPUT CHANNELS (64 each), share one staged activation tile, and every activation is read from HBM once per 256 couts.
// Two arithmetic modes, same structure: F16 (operands rounded to fp16: LINF precision='fp16', BASELINE config 5) and X3 (exact
// 3-term bf16 split, six products: fp32-accurate, the default path).  Epilogue, packing conventions (k-half-major 8-channel
// octets) and error behaviour as in conv_bf16x3.hip / conv_f16.hip.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/bfsr_hip.h"
#include "launch_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int NPX = 128, MWG = 256;          // pixels and output channels per workgroup

template <bool X3> struct Mode;
template <> struct Mode<false> { typedef half8 V; static constexpr int PL = 1, CK = 64; };
template <> struct Mode<true> { typedef bf16x8 V; static constexpr int PL = 3, CK = 32; };

template <bool X3>
__global__ __launch_bounds__(256, 2) void conv1x1_kernel(BfsrConvArgs p, long long P, int ptiles, int groups)
{
    typedef typename Mode<X3>::V V;
    constexpr int PL = Mode<X3>::PL, CK = Mode<X3>::CK, OCT = CK / 8, KST = CK / 16;
    constexpr int WPL = OCT * MWG * 8, IPL = OCT * NPX * 8;              // 16-bit elements per plane
    constexpr int WV = PL * WPL / 8 / 256;                              // 16-byte weight units per thread
    constexpr int IV = OCT / 2;                                         // activation octets per thread (2 thread halves)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short* sW = reinterpret_cast<unsigned short*>(smem_raw);   // [PL][OCT][256][8]
    unsigned short* sIn = sW + PL * WPL;                                // [PL][OCT][128][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    int bid = blockIdx.x;
    const int cg = bid % groups; bid /= groups;
    const int pt = bid % ptiles; const int b = bid / ptiles;
    const long long p0 = (long long)pt * NPX;
    const int Cin = p.Cin, nchunk = (Cin + CK - 1) / CK;
    const float* __restrict__ xin = p.x + (long long)b * p.x_bs;
    const unsigned short* __restrict__ wg = reinterpret_cast<const unsigned short*>(p.w) + (long long)cg * nchunk * PL * WPL;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, (unsigned)((long long)Cin * P * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wg), 0,
                                                                          (unsigned)((long long)nchunk * PL * WPL * 2), 0x00020000);
    // staging role: pixel = tid & 127, octet half = tid >> 7
    const int spx = tid & 127, soh = tid >> 7;
    const unsigned voff = (p0 + spx < P) ? (unsigned)(p0 + spx) * 4u : 0x80000000u;
    const unsigned cs_bytes = (unsigned)(P * 4);

    f32x16 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    float vin[IV][8];
    uint4 vw[WV];
    auto load_chunk = [&](int k) {
#pragma unroll
        for (int o = 0; o < IV; ++o)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const unsigned ch = (unsigned)(k * CK + (soh + 2 * o) * 8 + c);
                vin[o][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, voff, ch * cs_bytes, 0));
            }
#pragma unroll
        for (int i = 0; i < WV; ++i)
            vw[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(tid + i * 256) * 16u,
                                                                                    (unsigned)k * (PL * WPL * 2), 0));
    };
    load_chunk(0);
    for (int k = 0; k < nchunk; ++k) {
        __syncthreads();
#pragma unroll
        for (int o = 0; o < IV; ++o) {
            const int oct = soh + 2 * o;
            if constexpr (!X3) {
                half8 h;
#pragma unroll
                for (int c = 0; c < 8; ++c) h[c] = (_Float16)vin[o][c];
                *reinterpret_cast<half8*>(sIn + (oct * NPX + spx) * 8) = h;
            } else {
                bf16x8 h8, m8, l8;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float v = vin[o][c];
                    const __bf16 h = (__bf16)v;
                    const float r1 = v - (float)h;
                    const __bf16 m = (__bf16)r1;
                    h8[c] = h; m8[c] = m; l8[c] = (__bf16)(r1 - (float)m);
                }
                *reinterpret_cast<bf16x8*>(sIn + (oct * NPX + spx) * 8) = h8;
                *reinterpret_cast<bf16x8*>(sIn + IPL + (oct * NPX + spx) * 8) = m8;
                *reinterpret_cast<bf16x8*>(sIn + 2 * IPL + (oct * NPX + spx) * 8) = l8;
            }
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) reinterpret_cast<uint4*>(sW)[tid + i * 256] = vw[i];
        __syncthreads();
        if (k + 1 < nchunk) load_chunk(k + 1);
#pragma unroll
        for (int s = 0; s < KST; ++s) {
            V a[PL][2], bq[PL][4];
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    a[pl][m] = *reinterpret_cast<const V*>(sW + pl * WPL + ((2 * s + lhi) * MWG + wave * 64 + m * 32 + l31) * 8);
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    bq[pl][n] = *reinterpret_cast<const V*>(sIn + pl * IPL + ((2 * s + lhi) * NPX + n * 32 + l31) * 8);
            }
            if constexpr (!X3) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][m], bq[0][n], acc[m][n], 0, 0, 0);
            } else {
#define BFSR_TERM(PA_, PB_)                                                                                            \
    _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int n = 0; n < 4; ++n)                         \
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA_][m], bq[PB_][n], acc[m][n], 0, 0, 0);
                BFSR_TERM(2, 0) BFSR_TERM(0, 2) BFSR_TERM(1, 1) BFSR_TERM(1, 0) BFSR_TERM(0, 1) BFSR_TERM(0, 0)
#undef BFSR_TERM
            }
        }
    }

    // ---- epilogue (fp32, same stage order as conv_mfma_kernel); lane = pixel
    const float slope = p.act == BFSR_ACT_NONE ? 1.f : (p.act == BFSR_ACT_RELU ? 0.f : p.slope);
    const float4* __restrict__ epi = reinterpret_cast<const float4*>(p.epi);
    const float* pre = p.pre_add ? p.pre_add + (long long)b * p.pre_add_bs : nullptr;
    const float* r1 = p.res1 ? p.res1 + (long long)b * p.res1_bs : nullptr;
    const float* r2 = p.res2 ? p.res2 + (long long)b * p.res2_bs : nullptr;
    const float a1 = p.res1 ? p.alpha1 : 1.f, a2 = p.res2 ? p.alpha2 : 1.f;
    float* yb = p.y + (long long)b * p.y_bs;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cg * MWG + wave * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (co >= p.Cout) continue;
            float4 q0 = make_float4(0.f, 0.f, 1.f, 0.f); float q1 = 1.f;
            if (epi) { q0 = epi[co * 2]; q1 = epi[co * 2 + 1].x; }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const long long px = p0 + n * 32 + l31;
                if (px >= P) continue;
                const long long o = (long long)co * P + px;
                float v = acc[m][n][r] + q0.x;
                if (pre) v += pre[o];
                v = (v + q0.y) * q0.z + q0.w;
                v = v > 0.f ? v : v * slope;
                v *= q1;
                if (r1) v = a1 * v + r1[o];
                if (r2) v = a2 * v + r2[o];
                yb[o] = v;
            }
        }
}

template <bool X3>
int launch_1x1(const BfsrConvArgs& a, hipStream_t st)
{
    constexpr int PL = Mode<X3>::PL, CK = Mode<X3>::CK;
    constexpr int LDS = PL * (CK / 8) * (MWG + NPX) * 8 * 2;
    static std::atomic<unsigned long long> lds_done{0};
    if (LDS > 65536 && bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&conv1x1_kernel<X3>), LDS, lds_done) != 0) return -1;
    const long long P = (long long)a.H * a.W;
    const long long ptiles = (P + NPX - 1) / NPX;
    const int groups = (a.Cout + MWG - 1) / MWG;
    const long long nblk = ptiles * groups * a.B;
    if (nblk <= 0 || nblk > 0x7fffffffLL || ptiles > 0x7fffffffLL) return -1;
    hipLaunchKernelGGL((conv1x1_kernel<X3>), dim3((unsigned)nblk), dim3(256), LDS, st, a, P, (int)ptiles, groups);
    return (int)hipGetLastError();
}

template <bool X3>
int pack_1x1(const float* w, int Cout, int Cin, unsigned short* packed)
{
    // w [Cout][Cin] fp32 -> 16-bit [cout_group(256)][chunk][plane][octet][256][8], zero padded
    constexpr int PL = Mode<X3>::PL, CK = Mode<X3>::CK, OCT = CK / 8;
    const int nchunk = (Cin + CK - 1) / CK, groups = (Cout + MWG - 1) / MWG;
    const long long n = (long long)groups * nchunk * PL * OCT * MWG * 8;
    for (long long i = 0; i < n; ++i) packed[i] = 0;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const float v = w[(long long)co * Cin + ci];
            unsigned short s3[3] = {0, 0, 0};
            if (X3) {
                float r = v;
                for (int i = 0; i < 3; ++i) { const __bf16 h = (__bf16)r; __builtin_memcpy(&s3[i], &h, 2); r -= (float)h; }
            } else {
                const _Float16 h = (_Float16)v; __builtin_memcpy(&s3[0], &h, 2);
            }
            const int g = co / MWG, m = co % MWG, ch = ci / CK, oc = (ci % CK) / 8, e = ci % 8;
            for (int pl = 0; pl < PL; ++pl)
                packed[(((((long long)g * nchunk + ch) * PL + pl) * OCT + oc) * MWG + m) * 8 + e] = s3[pl];
        }
    return 0;
}

bool args_ok(const BfsrConvArgs* a)
{
    if (!a || !a->x || !a->w || !a->y || a->w2 || a->x2 || a->KS != 1 || a->in_shift) return false;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->Cin <= 0 || a->Cout <= 0) return false;
    return (long long)a->Cin * a->H * a->W * 4 < (1LL << 31);
}

}  // namespace

extern "C" long long bfsr_conv1x1_packed_size(int Cout, int Cin, int x3)
{
    const int CK = x3 ? 32 : 64, PL = x3 ? 3 : 1;
    return (long long)((Cout + MWG - 1) / MWG) * ((Cin + CK - 1) / CK) * PL * (CK / 8) * MWG * 8;
}

extern "C" int bfsr_pack_conv1x1_weight(const float* w, int Cout, int Cin, int x3, unsigned short* packed)
{
    if (!w || !packed || Cout <= 0 || Cin <= 0) return -1;
    return x3 ? pack_1x1<true>(w, Cout, Cin, packed) : pack_1x1<false>(w, Cout, Cin, packed);
}

extern "C" int bfsr_conv1x1(const BfsrConvArgs* a, int x3, void* stream)
{
    if (!args_ok(a)) return -1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return x3 ? launch_1x1<true>(*a, st) : launch_1x1<false>(*a, st);
}
