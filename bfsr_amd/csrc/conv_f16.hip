// conv_f16.hip -- the reduced-precision conv path (BASELINE config 5, "fp16 MFMA path"): same implicit-GEMM structure as
// conv_mfma.hip but the contraction runs on v_mfma_f32_32x32x16_f16 (16x the fp32 MFMA rate): activations and weights
// are rounded to fp16 when they are staged into LDS, accumulation and the whole epilogue stay fp32, tensors in HBM stay
// fp32 (drop-in for bfsr_conv2d; NOT within the 1e-4 fp32 tolerance -- parity tests report the max-abs vs fp32).
//
// GEMM view: M = cout (MR tiles of 32), N = 32 pixels of a row, K = 16 input channels per MFMA.  LDS holds the input tile
// k-half-major ([k half][position][8 halfs]) and the weight slab as [tap][k half][cout][8 halfs]: each MFMA operand is one
// ds_read_b128 per lane and the 64 lanes read 2 x 512 contiguous bytes (no bank conflicts).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/bfsr_hip.h"
#include "launch_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

namespace {

constexpr int CK = 16;           // input channels per LDS stage = one MFMA k-step per tap

template <int KS, int MR, int NR, int NW>
__global__ __launch_bounds__(NW * 64, 2) void conv_f16_kernel(BfsrConvArgs p, int tiles_x, int tiles_xy, int groups)
{
    constexpr int NT = NW * 64;
    constexpr int TH = NW * NR, TW = 32, HALO = KS - 1;
    constexpr int IH = TH + HALO, PW = TW + HALO, NPOS = IH * PW, PPT = (NPOS + NT - 1) / NT;
    constexpr int TAPS = KS * KS, MW = MR * 32;
    constexpr int WSLAB = TAPS * MW * CK;            // halfs of weights per chunk
    constexpr int WV = (WSLAB / 8 + NT - 1) / NT;      // 16-byte weight loads per thread

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16* sW = reinterpret_cast<_Float16*>(smem_raw);                 // [TAPS][k half][MW][8]
    _Float16* sIn = sW + WSLAB;                                           // [k half][NPOS][8]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    // XCD-aware block order (round 6): the cout groups of a tile run on ONE XCD, i.e. behind one L2 -- in plain block order they were dealt to all eight and every
    // XCD fetched the fp32 input tile for itself (PMC: 8.8 GB fetched for a 1.07 GB input at 128 x 64^2, 512 -> 256: the launch was HBM-bound on re-reads).
    // config 5: conv_f16 4.26 -> 3.64 ms per pass (profiles/r06z_block_order.txt).  The split-precision twins (conv_bf16x3.hip, conv_mfma.hip, conv1x1.hip) keep
    // the plain order: the same change measured 2-4 % SLOWER there (their weight tensors are 2-3x larger: with plain order an XCD keeps re-using one group's weights).
    int bid = (int)bfsr::xcd_order(blockIdx.x, gridDim.x);
    const int cg = bid % groups; bid /= groups;
    const int tile = bid % tiles_xy; const int b = bid / tiles_xy;
    const int x0 = (tile % tiles_x) * TW, y0 = (tile / tiles_x) * TH;

    const int H = p.H, W = p.W, sh = p.in_shift, Ws = W >> sh;
    const long long cs_in = (long long)(H >> sh) * Ws;
    const float* __restrict__ xin = p.x + (long long)b * p.x_bs;
    const int Cin = p.Cin;
    const int nchunk = (Cin + CK - 1) / CK;
    const _Float16* __restrict__ wg = reinterpret_cast<const _Float16*>(p.w) + (long long)cg * nchunk * WSLAB;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0,
                                                                           (unsigned)((long long)Cin * cs_in * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(wg), 0,
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
                half8 lo, hi;
#pragma unroll
                for (int c = 0; c < 8; ++c) { lo[c] = (_Float16)vin[i][c]; hi[c] = (_Float16)vin[i][8 + c]; }
                *reinterpret_cast<half8*>(sIn + pos * 8) = lo;                 // k-half-major: conflict-free b128 accesses
                *reinterpret_cast<half8*>(sIn + (NPOS + pos) * 8) = hi;
            }
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int idx = tid + i * NT;
            if (i < WV - 1 || idx < WSLAB / 8) reinterpret_cast<uint4*>(sW)[idx] = vw[i];
        }
        __syncthreads();
        if (k + 1 < nchunk) load_chunk(k + 1);
        const _Float16* inB = sIn + (lhi * NPOS + (wave * NR) * PW + l31) * 8;
        const _Float16* wA = sW + (lhi * MW + l31) * 8;
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) {
            half8 brow[NR + HALO];
#pragma unroll
            for (int r = 0; r < NR + HALO; ++r) brow[r] = *reinterpret_cast<const half8*>(inB + (r * PW + dx) * 8);
#pragma unroll
            for (int dy = 0; dy < KS; ++dy) {
                half8 a[MR];
#pragma unroll
                for (int m = 0; m < MR; ++m) a[m] = *reinterpret_cast<const half8*>(wA + ((dy * KS + dx) * 2 * MW + m * 32) * 8);
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int n = 0; n < NR; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], brow[n + dy], acc[m][n], 0, 0, 0);
            }
        }
    }

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
                    float v = acc[m][n][r];
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

template <int KS, int MR, int NR, int NW>
int launch_f16(const BfsrConvArgs& a, hipStream_t st)
{
    constexpr int TH = NW * NR, HALO = KS - 1;
    constexpr int LDS = (KS * KS * MR * 32 * CK + (TH + HALO) * (32 + HALO) * CK) * 2;
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + TH - 1) / TH;
    const int groups = ((a.Cout + 31) / 32 + MR - 1) / MR;
    const long long nblk = (long long)tiles_x * tiles_y * groups * a.B;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return -1;
    hipLaunchKernelGGL((conv_f16_kernel<KS, MR, NR, NW>), dim3((unsigned)nblk), dim3(NW * 64), LDS, st, a, tiles_x, tiles_x * tiles_y, groups);
    return (int)hipGetLastError();
}

inline unsigned short f32_to_f16_bits(float f)
{
    const _Float16 h = (_Float16)f;                   // round-to-nearest-even, same conversion the kernel applies
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}

}  // namespace

extern "C" long long bfsr_conv_packed_size_f16(int Cout, int Cin, int KS, int mtile)
{
    const int nchunk = (Cin + CK - 1) / CK;
    const int groups = ((Cout + 31) / 32 + mtile - 1) / mtile;
    return (long long)groups * nchunk * KS * KS * mtile * 32 * CK;      // number of fp16 elements
}

extern "C" int bfsr_pack_conv_weight_f16(const float* w, int Cout, int Cin, int KS, int mtile, unsigned short* packed)
{
    // w [Cout][Cin][KS][KS] fp32 -> fp16 [cout_group][chunk][tap][k half][mtile*32][8], zero padded
    if ((KS != 1 && KS != 3) || mtile < 1) return -1;
    const int nchunk = (Cin + CK - 1) / CK, MW = mtile * 32, T = KS * KS;
    const int groups = ((Cout + 31) / 32 + mtile - 1) / mtile;
    const long long n = (long long)groups * nchunk * T * MW * CK;
    for (long long i = 0; i < n; ++i) packed[i] = 0;
    for (int co = 0; co < Cout; ++co) {
        const int g = co / MW, m = co % MW;
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < T; ++t)
                packed[(((((long long)g * nchunk + ci / CK) * T + t) * 2 + (ci % CK) / 8) * MW + m) * 8 + ci % 8] =
                    f32_to_f16_bits(w[((long long)co * Cin + ci) * T + t]);
    }
    return 0;
}

extern "C" int bfsr_conv2d_f16(const BfsrConvArgs* a, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || !a->x || !a->w || !a->y || a->w2 || a->x2) return -1;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->Cin <= 0 || a->Cout <= 0 || a->in_shift < 0 || a->in_shift > 4) return -1;
    if (a->in_shift && (((a->H >> a->in_shift) << a->in_shift) != a->H || ((a->W >> a->in_shift) << a->in_shift) != a->W)) return -1;
    if ((long long)a->Cin * (a->H >> a->in_shift) * (a->W >> a->in_shift) * 4 >= (1LL << 31)) return -1;
    if ((a->pre_add || a->res1 || a->res2) && (long long)a->Cout * a->H * a->W * 4 >= (1LL << 31)) return -1;
    const long long groups_ = ((a->Cout + 31) / 32 + a->mtile - 1) / a->mtile;
    const long long tiles4 = (long long)((a->W + 31) / 32) * ((a->H + 15) / 16) * a->B;
    // 3x3: 8-wave workgroups, one tile row per wave (measured best or tied on the LINF encoder shapes, +10-17 % over the
    // 4-wave tiles); 1x1: 4 waves x 2-4 rows
    int NR = tiles4 * groups_ >= 1024 ? 4 : 2, NW = 4;
    if (a->KS == 3) { NR = 1; NW = 8; }
    if (a->tune) { NR = a->tune / 100; NW = a->tune % 100; }
    const int key = a->KS * 10000 + a->mtile * 1000 + NR * 100 + NW;
    switch (key) {
#define V(KS_, MR_, NR_, NW_) case KS_ * 10000 + MR_ * 1000 + NR_ * 100 + NW_: return launch_f16<KS_, MR_, NR_, NW_>(*a, st);
        V(3, 1, 2, 4) V(3, 1, 4, 4) V(3, 2, 2, 4) V(3, 2, 4, 4) V(1, 1, 2, 4) V(1, 1, 4, 4) V(1, 2, 2, 4) V(1, 2, 4, 4)
        V(3, 1, 1, 8) V(3, 1, 2, 8) V(3, 2, 1, 8) V(3, 2, 2, 8) V(3, 1, 1, 4) V(3, 2, 1, 4)
#undef V
        default: return -1;
    }
}
