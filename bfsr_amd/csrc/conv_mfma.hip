// conv_mfma.hip -- fp32 implicit-GEMM convolution (3x3 'same' / 1x1, stride 1) on the CDNA4 matrix
// cores (v_mfma_f32_32x32x2_f32: exact fp32, 157 TFLOP/s peak on MI355X), fused epilogue.
//
// GEMM view per workgroup:  D[cout][pixel] = sum_{cin,tap} Wt[cout][cin,tap] * X[cin,tap][pixel]
//   M = cout  (MR tiles of 32 per workgroup)            -> A operand = weights
//   N = pixel (32 consecutive pixels of one image row)  -> B operand = input
//   K = cin * taps, walked CK input channels at a time through LDS
// With M = cout the accumulator layout puts 32 consecutive pixels of one (cout,row) in lanes 0..31,
// so epilogue loads/stores are 128-byte coalesced in NCHW.
//
// Workgroup = 256 threads = 4 wave64; pixel tile = (4*NR rows) x 32 cols; each wave owns NR rows and
// all MR cout tiles: acc[MR][NR] of 16 VGPRs each.  LDS holds the CK-channel input tile with halo
// ([CK][TH+2][34]) and the weight slab ([CK][taps][MR*32]); two workgroups per CU overlap one
// group's staging with the other's MFMA stream.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/bfsr_hip.h"
#include "launch_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int CIN_ALIGN = 16;      // packed weights pad Cin to a multiple of this (any CK in {8,16} divides it)

template <int KS, int MR, int NR, int CK, bool FUSE2 = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(BfsrConvArgs p, int tiles_x, int tiles_xy, int groups)
{
    constexpr int TH = 4 * NR, TW = 32;
    constexpr int HALO = KS - 1;
    constexpr int IH = TH + HALO, PW = TW + HALO;
    constexpr int NPOS = IH * PW;
    constexpr int PPT = (NPOS + 255) / 256;          // staged positions per thread
    constexpr int TAPS = KS * KS;
    constexpr int MW = MR * 32;
    constexpr int WCHUNK = CK * TAPS * MW;           // floats of weights per cin chunk

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW = smem;                                // [CK][TAPS][MW]  (16B aligned for float4 stores)
    float* sIn = smem + WCHUNK;                      // [CK][IH][PW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;

    // block id -> (cout group fastest, spatial tile, batch): groups of one tile run back to back
    int bid = blockIdx.x;
    const int cg = bid % groups; bid /= groups;
    const int tile = bid % tiles_xy; const int b = bid / tiles_xy;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;

    const int H = p.H, W = p.W, sh = p.in_shift;
    const int Ws = W >> sh;
    const long long cs_in = (long long)(H >> sh) * Ws;   // input channel stride
    const float* __restrict__ xin = p.x + (long long)b * p.x_bs;
    const int Cin = p.Cin;
    const int cin_pad = (Cin + CIN_ALIGN - 1) / CIN_ALIGN * CIN_ALIGN;   // stride of the packed weights
    const int cin_loop = (Cin + CK - 1) / CK * CK;                       // channels actually walked
    const float* __restrict__ wg = p.w + (long long)cg * cin_pad * TAPS * MW;

    // Staging uses raw buffer loads (one instruction each: VGPR byte offset + SGPR channel offset, no address
    // arithmetic in the loop).  The descriptor's range check returns 0 beyond `num_records`, which implements both
    // the conv zero padding (out-of-image positions get an out-of-range offset) and the channel padding to CK.
    const unsigned in_bytes = (unsigned)((long long)Cin * cs_in * 4);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wg), 0,
                                                                          (unsigned)(cin_pad * TAPS * MW * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;            // > any valid offset (views are < 2 GiB), no 32-bit wrap
    unsigned voff[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int pos = tid + i * 256;
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

    // register-staged software pipeline: the loads of chunk k+1 are in flight during the MFMAs of chunk k
    constexpr int WV = (WCHUNK / 4 + 255) / 256;     // float4 weight loads per thread
    float vin[PPT][CK];
    float4 vw[WV];
    auto load_chunk = [&](int c0) {
        const unsigned sbase = (unsigned)c0 * cs_bytes;
#pragma unroll
        for (int c = 0; c < CK; ++c) {
            const unsigned so = sbase + (unsigned)c * cs_bytes;
#pragma unroll
            for (int i = 0; i < PPT; ++i)
                vin[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, voff[i], so, 0));
        }
        const unsigned wbase = (unsigned)c0 * (TAPS * MW * 4);
#pragma unroll
        for (int i = 0; i < WV; ++i)
            vw[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(tid + i * 256) * 16u, wbase, 0));
    };
    load_chunk(0);

    for (int c0 = 0; c0 < cin_loop; c0 += CK) {
        __syncthreads();
        // ---- registers -> LDS (input tile zero padded outside the image; weight slab contiguous)
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int pos = tid + i * 256;
            if (i < PPT - 1 || pos < NPOS) {
#pragma unroll
                for (int c = 0; c < CK; ++c) sIn[c * NPOS + pos] = vin[i][c];
            }
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int idx = tid + i * 256;
            if (i < WV - 1 || idx < WCHUNK / 4) reinterpret_cast<float4*>(sW)[idx] = vw[i];
        }
        __syncthreads();
        if (c0 + CK < cin_loop) load_chunk(c0 + CK);
        // ---- MFMA stream
#pragma unroll
        for (int kk = 0; kk < CK / 2; ++kk) {
            const int c = 2 * kk + lhi;
            const float* inC = sIn + c * NPOS + (wave * NR) * PW + l31;
            const float* wC = sW + c * TAPS * MW + l31;
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) {
                float brow[NR + HALO];
#pragma unroll
                for (int r = 0; r < NR + HALO; ++r) brow[r] = inC[r * PW + dx];
#pragma unroll
                for (int dy = 0; dy < KS; ++dy) {
                    float a[MR];
#pragma unroll
                    for (int m = 0; m < MR; ++m) a[m] = wC[(dy * KS + dx) * MW + m * 32];
#pragma unroll
                    for (int m = 0; m < MR; ++m)
#pragma unroll
                        for (int n = 0; n < NR; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], brow[n + dy], acc[m][n], 0, 0, 0);
                }
            }
        }
    }

    const long long HW = (long long)H * W;
    const int gx = x0 + l31;
    const bool col_ok = gx < W;
    // Epilogue parameters are packed per output channel as 8 floats {bias, aff_shift, aff_scale, aff_post, post_scale}
    // with neutral defaults, so the element math is branch-free:
    //   v = ((acc + bias + pre_add) + aff_shift) * aff_scale + aff_post ; v = v > 0 ? v : v*slope ; v *= post_scale ;
    //   v = alpha1*v + res1 ; v = alpha2*v + res2
    // (slope 1 = no activation, 0 = ReLU).  Optional tensors are fetched with range-checked buffer loads: a missing
    // tensor gets an empty descriptor and reads as 0, an out-of-image pixel gets an out-of-range offset.
    const float slope = p.act == BFSR_ACT_NONE ? 1.f : (p.act == BFSR_ACT_RELU ? 0.f : p.slope);
    const float4* __restrict__ epi = reinterpret_cast<const float4*>(p.epi);
    auto chan_params = [&](const float4* __restrict__ e, int co, float4& q0, float& q1) {
        if (e) { q0 = e[co * 2]; q1 = e[co * 2 + 1].x; }
        else { q0 = make_float4(0.f, 0.f, 1.f, 0.f); q1 = 1.f; }
    };
    const unsigned out_bytes = (unsigned)((long long)p.Cout * HW * 4);
    auto tensor_rsrc = [&](const float* t, long long bs) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t ? t + (long long)b * bs : p.y), 0, t ? out_bytes : 0u,
                                                 0x00020000);
    };

    if constexpr (FUSE2) {
        // ---- fused second stage: y = act2((W2 . act1(epi1(conv)) + s2_shift) * s2_scale), W2 a 1x1 conv over the MR*32
        // stage-1 channels of this pixel tile.  The stage-1 tile goes through LDS 32 channels at a time
        // ([32][4 waves][NR*32 px]), so the hidden tensor never touches HBM.
        constexpr int MR2 = 2, MW2 = MR2 * 32, NPX = NR * 32;
        float* sH = smem;                                  // [32][4*NPX]
        float* sW2 = smem + 32 * 4 * NPX;                  // [MR*32][MW2]
        __syncthreads();                                   // all waves finished reading the last K-chunk
        {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(p.w2);
            float4* dst = reinterpret_cast<float4*>(sW2);
            for (int i = tid; i < MR * 32 * MW2 / 4; i += 256) dst[i] = src[i];
        }
        const __amdgpu_buffer_rsrc_t rs_pre = tensor_rsrc(p.pre_add, p.pre_add_bs);
        unsigned pixoff[NR];
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            const int gy = y0 + wave * NR + n;
            pixoff[n] = (col_ok && gy < H) ? (unsigned)(gy * W + gx) * 4u : OOB;
        }
        f32x16 acc2[MR2][NR];
#pragma unroll
        for (int m = 0; m < MR2; ++m)
#pragma unroll
            for (int n = 0; n < NR; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[m][n][r] = 0.f;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m > 0) __syncthreads();                    // previous 32-channel slab fully consumed
            float pre[NR][16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const unsigned cbase = (unsigned)co * (unsigned)(HW * 4);
#pragma unroll
                for (int n = 0; n < NR; ++n)
                    pre[n][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                              rs_pre, co < p.Cout ? pixoff[n] + cbase : OOB, 0, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int co = m * 32 + row;
                float4 q0; float q1;
                chan_params(epi, co < p.Cout ? co : 0, q0, q1);
#pragma unroll
                for (int n = 0; n < NR; ++n) {
                    float v = acc[m][n][r];
                    v += q0.x; v += pre[n][r]; v += q0.y; v *= q0.z; v += q0.w;
                    v = v > 0.f ? v : v * slope;
                    if (co >= p.Cout) v = 0.f;
                    sH[row * (4 * NPX) + wave * NPX + n * 32 + l31] = v;
                }
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int c = 2 * kk + lhi;
                float a2[MR2], b2[NR];
#pragma unroll
                for (int m2 = 0; m2 < MR2; ++m2) a2[m2] = sW2[(m * 32 + c) * MW2 + m2 * 32 + l31];
#pragma unroll
                for (int n = 0; n < NR; ++n) b2[n] = sH[c * (4 * NPX) + wave * NPX + n * 32 + l31];
#pragma unroll
                for (int m2 = 0; m2 < MR2; ++m2)
#pragma unroll
                    for (int n = 0; n < NR; ++n)
                        acc2[m2][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[m2], b2[n], acc2[m2][n], 0, 0, 0);
            }
        }
        if (!col_ok) return;
        const float slope2 = p.act2 == BFSR_ACT_NONE ? 1.f : (p.act2 == BFSR_ACT_RELU ? 0.f : p.slope);
        const float4* __restrict__ epi2 = reinterpret_cast<const float4*>(p.epi2);
#pragma unroll
        for (int m2 = 0; m2 < MR2; ++m2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (co >= p.C2) continue;
                float4 q0; float q1;
                chan_params(epi2, co, q0, q1);
#pragma unroll
                for (int n = 0; n < NR; ++n) {
                    const int gy = y0 + wave * NR + n;
                    if (gy >= H) continue;
                    float v = acc2[m2][n][r];
                    v += q0.x; v += q0.y; v *= q0.z; v += q0.w;
                    v = v > 0.f ? v : v * slope2;
                    v *= q1;
                    p.y[(long long)b * p.y_bs + (long long)co * HW + (long long)gy * W + gx] = v;
                }
            }
        return;
    }

    // ---- epilogue: lane holds pixel column l31 of rows (wave*NR+n) and 16 couts per M tile
    if (!col_ok) return;
    const int Cout = p.Cout;
    const bool tensors = p.pre_add || p.res1 || p.res2;
    auto run_epilogue = [&](auto with_tensors) {
        constexpr bool T = decltype(with_tensors)::value;
        const __amdgpu_buffer_rsrc_t rs_pre = tensor_rsrc(p.pre_add, p.pre_add_bs);
        const __amdgpu_buffer_rsrc_t rs_r1 = tensor_rsrc(p.res1, p.res1_bs);
        const __amdgpu_buffer_rsrc_t rs_r2 = tensor_rsrc(p.res2, p.res2_bs);
        const float a1 = p.res1 ? p.alpha1 : 1.f, a2 = p.res2 ? p.alpha2 : 1.f;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (cg * MR + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (co >= Cout) continue;
                float4 q0; float q1;
                chan_params(epi, co, q0, q1);
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
        }
    };
    if (tensors) run_epilogue(std::true_type{});
    else run_epilogue(std::false_type{});
}

template <int KS, int MR, int NR, int CK, bool FUSE2 = false>
int launch_conv(const BfsrConvArgs& a, hipStream_t st)
{
    constexpr int TH = 4 * NR, TW = 32, HALO = KS - 1;
    constexpr int LDS_MAIN = (CK * KS * KS * MR * 32 + CK * (TH + HALO) * (TW + HALO)) * 4;
    constexpr int LDS_FUSE = FUSE2 ? (32 * 4 * NR * 32 + MR * 32 * 64) * 4 : 0;
    constexpr int LDS = LDS_MAIN > LDS_FUSE ? LDS_MAIN : LDS_FUSE;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int groups = ((a.Cout + 31) / 32 + MR - 1) / MR;
    const long long nblk = (long long)tiles_x * tiles_y * groups * a.B;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return -1;
    static std::atomic<unsigned long long> lds_done{0};
    if (LDS > 48 * 1024 && bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_mfma_kernel<KS, MR, NR, CK, FUSE2>), LDS, lds_done) != 0)
        return -1;
    hipLaunchKernelGGL((conv_mfma_kernel<KS, MR, NR, CK, FUSE2>), dim3((unsigned)nblk), dim3(256), LDS, st, a, tiles_x,
                       tiles_x * tiles_y, groups);
    return (int)hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------
// 3x3 'same' conv over a nearest-x2-upsampled input WITHOUT materialising the upsample and at 4/9 of the MACs:
// for output pixel (2y+a, 2x+b) the nine taps of U = nearest_up2(T) land on only 2x2 source pixels
//   rows  a=0: {y-1 <- w[-1], y <- w[0]+w[+1]}      a=1: {y <- w[-1]+w[0], y+1 <- w[+1]}      (same in x)
// so the conv is four 2x2 convs on T (one per output parity) with pre-summed weights (16 [Cout x Cin] matrices,
// packed as 16 "taps" t = (a*2+b)*4 + i*2+j by the host).  GEMM view per parity: M = cout, N = 32 consecutive SOURCE
// columns (= every other output column), K = cin*4.  Workgroup = 4 waves = 4 source rows x 32 source columns ->
// 8 x 64 output pixels; each wave keeps acc[parity][MR].  Used for the hoisted level-1 coupling convs of SRFlow, whose
// 256 stacked-RRDB conditioning channels are the LR-resolution block taps upsampled x2 (SRFlowNet_arch.py:137).
template <int MR>
__global__ __launch_bounds__(256, 2) void conv_up2_kernel(BfsrConvArgs p, int tiles_x, int tiles_xy, int groups)
{
    constexpr int CK = 8, TH = 4, TW = 32, IH = TH + 2, PW = TW + 2, NPOS = IH * PW, TAPS = 16, MW = MR * 32;
    constexpr int WCHUNK = CK * TAPS * MW;
    static_assert(NPOS <= 256, "one staged position per thread");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW = smem;                                // [CK][16][MW]
    float* sIn = smem + WCHUNK;                      // [CK][IH][PW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    int bid = blockIdx.x;
    const int cg = bid % groups; bid /= groups;
    const int tile = bid % tiles_xy; const int b = bid / tiles_xy;
    const int x0 = (tile % tiles_x) * TW, y0 = (tile / tiles_x) * TH;     // source coordinates
    const int H = p.H, W = p.W, Hs = H >> 1, Ws = W >> 1;
    const long long cs_in = (long long)Hs * Ws;
    const float* __restrict__ xin = p.x + (long long)b * p.x_bs;
    const int Cin = p.Cin;
    const int cin_pad = (Cin + CIN_ALIGN - 1) / CIN_ALIGN * CIN_ALIGN;
    const int cin_loop = (Cin + CK - 1) / CK * CK;
    const float* __restrict__ wg = p.w + (long long)cg * cin_pad * TAPS * MW;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0,
                                                                           (unsigned)((long long)Cin * cs_in * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wg), 0,
                                                                          (unsigned)(cin_pad * TAPS * MW * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned voff;
    {
        const int r = tid / PW, c = tid - r * PW;
        const int gy = y0 + r - 1, gx = x0 + c - 1;
        const bool ok = tid < NPOS && gy >= 0 && gy < Hs && gx >= 0 && gx < Ws;
        voff = ok ? (unsigned)(gy * Ws + gx) * 4u : OOB;
    }
    const unsigned cs_bytes = (unsigned)(cs_in * 4);
    f32x16 acc[4][MR];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][m][r] = 0.f;

    constexpr int WV = (WCHUNK / 4 + 255) / 256;
    float vin[CK];
    float4 vw[WV];
    auto load_chunk = [&](int c0) {
        const unsigned sbase = (unsigned)c0 * cs_bytes;
#pragma unroll
        for (int c = 0; c < CK; ++c)
            vin[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, voff, sbase + (unsigned)c * cs_bytes, 0));
        const unsigned wbase = (unsigned)c0 * (TAPS * MW * 4);
#pragma unroll
        for (int i = 0; i < WV; ++i)
            vw[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(tid + i * 256) * 16u, wbase, 0));
    };
    load_chunk(0);
    for (int c0 = 0; c0 < cin_loop; c0 += CK) {
        __syncthreads();
        if (tid < NPOS) {
#pragma unroll
            for (int c = 0; c < CK; ++c) sIn[c * NPOS + tid] = vin[c];
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int idx = tid + i * 256;
            if (i < WV - 1 || idx < WCHUNK / 4) reinterpret_cast<float4*>(sW)[idx] = vw[i];
        }
        __syncthreads();
        if (c0 + CK < cin_loop) load_chunk(c0 + CK);
#pragma unroll
        for (int kk = 0; kk < CK / 2; ++kk) {
            const int c = 2 * kk + lhi;
            const float* inC = sIn + c * NPOS + wave * PW + l31;       // source row (y0+wave) - 1 .. + 1, col x - 1 .. + 1
            const float* wC = sW + c * TAPS * MW + l31;
            float bf[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int d = 0; d < 3; ++d) bf[r][d] = inC[r * PW + d];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int t = (a * 2 + bb) * 4 + i * 2 + j;
#pragma unroll
                            for (int m = 0; m < MR; ++m)
                                acc[a * 2 + bb][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wC[t * MW + m * 32], bf[a + i][bb + j],
                                                                                       acc[a * 2 + bb][m], 0, 0, 0);
                        }
        }
    }
    // ---- optional second K loop: `p.x2` = channels that live at OUTPUT resolution (the level's own 64 key
    // channels), ordinary 3x3 taps with weights `p.w_x2` ([cin_pad2][9][MW]); they accumulate into the same
    // parity accumulators, so the whole conv over cat[key, nearest_up2(taps)] is one kernel and one store.
    if (p.x2) {
        constexpr int KIH = 2 * TH + 2, KPW = 2 * TW + 2, KNPOS = KIH * KPW, KPPT = (KNPOS + 255) / 256;
        constexpr int KW = CK * 9 * MW;                           // floats of key weights per chunk
        float* kW = smem;                                         // [CK][9][MW]
        float* kIn = smem + KW;                                   // [CK][KIH][KPW]
        const int C2 = p.Cin2;
        const int cin2_pad = (C2 + CIN_ALIGN - 1) / CIN_ALIGN * CIN_ALIGN, cin2_loop = (C2 + CK - 1) / CK * CK;
        const long long cs2 = (long long)H * W;
        const float* __restrict__ x2 = p.x2 + (long long)b * p.x2_bs;
        const float* __restrict__ wk = p.w_x2 + (long long)cg * cin2_pad * 9 * MW;
        const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x2), 0,
                                                                              (unsigned)((long long)C2 * cs2 * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_wk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wk), 0,
                                                                               (unsigned)(cin2_pad * 9 * MW * 4), 0x00020000);
        unsigned kvoff[KPPT];
#pragma unroll
        for (int i = 0; i < KPPT; ++i) {
            const int pos = tid + i * 256;
            const int r = pos / KPW, c = pos - r * KPW;
            const int gy = 2 * y0 + r - 1, gx = 2 * x0 + c - 1;
            const bool ok = pos < KNPOS && gy >= 0 && gy < H && gx >= 0 && gx < W;
            kvoff[i] = ok ? (unsigned)(gy * W + gx) * 4u : OOB;
        }
        const unsigned cs2_bytes = (unsigned)(cs2 * 4);
        constexpr int KWV = (KW / 4 + 255) / 256;
        float kin[KPPT][CK];
        float4 kw[KWV];
        auto kload = [&](int c0) {
#pragma unroll
            for (int c = 0; c < CK; ++c)
#pragma unroll
                for (int i = 0; i < KPPT; ++i)
                    kin[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_k, kvoff[i], (unsigned)(c0 + c) * cs2_bytes, 0));
#pragma unroll
            for (int i = 0; i < KWV; ++i)
                kw[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_wk, (unsigned)(tid + i * 256) * 16u,
                                                                                        (unsigned)c0 * (9 * MW * 4), 0));
        };
        kload(0);
        for (int c0 = 0; c0 < cin2_loop; c0 += CK) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < KPPT; ++i) {
                const int pos = tid + i * 256;
                if (i < KPPT - 1 || pos < KNPOS) {
#pragma unroll
                    for (int c = 0; c < CK; ++c) kIn[c * KNPOS + pos] = kin[i][c];
                }
            }
#pragma unroll
            for (int i = 0; i < KWV; ++i) {
                const int idx = tid + i * 256;
                if (i < KWV - 1 || idx < KW / 4) reinterpret_cast<float4*>(kW)[idx] = kw[i];
            }
            __syncthreads();
            if (c0 + CK < cin2_loop) kload(c0 + CK);
#pragma unroll
            for (int kk = 0; kk < CK / 2; ++kk) {
                const int c = 2 * kk + lhi;
                // this wave's source row `wave` covers output rows 2*wave, 2*wave+1: tile rows 2*wave .. 2*wave+3
                const float* inC = kIn + c * KNPOS + (2 * wave) * KPW + 2 * l31;
                const float* wC = kW + c * 9 * MW + l31;
                float bk[4][4];                                   // [tile row][column offset], stride-2 lanes
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int d = 0; d < 4; ++d) bk[r][d] = inC[r * KPW + d];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        float aw[MR];
#pragma unroll
                        for (int m = 0; m < MR; ++m) aw[m] = wC[(dy * 3 + dx) * MW + m * 32];
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                                for (int m = 0; m < MR; ++m)
                                    acc[a * 2 + bb][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[m], bk[a + dy][bb + dx],
                                                                                           acc[a * 2 + bb][m], 0, 0, 0);
                    }
            }
        }
    }

    // ---- epilogue (same stage order as conv_mfma_kernel)
    const int sx = x0 + l31, sy = y0 + wave;
    if (sx >= Ws || sy >= Hs) return;
    const long long HW = (long long)H * W;
    const float slope = p.act == BFSR_ACT_NONE ? 1.f : (p.act == BFSR_ACT_RELU ? 0.f : p.slope);
    const float4* __restrict__ epi = reinterpret_cast<const float4*>(p.epi);
    const unsigned out_bytes = (unsigned)((long long)p.Cout * HW * 4);
    auto tensor_rsrc = [&](const float* t, long long bs) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t ? t + (long long)b * bs : p.y), 0, t ? out_bytes : 0u, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rs_pre = tensor_rsrc(p.pre_add, p.pre_add_bs);
    const __amdgpu_buffer_rsrc_t rs_r1 = tensor_rsrc(p.res1, p.res1_bs);
    const __amdgpu_buffer_rsrc_t rs_r2 = tensor_rsrc(p.res2, p.res2_bs);
    const bool tensors = p.pre_add || p.res1 || p.res2;
    const float a1 = p.res1 ? p.alpha1 : 1.f, a2 = p.res2 ? p.alpha2 : 1.f;
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (cg * MR + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (co >= p.Cout) continue;
            float4 q0 = make_float4(0.f, 0.f, 1.f, 0.f); float q1 = 1.f;
            if (epi) { q0 = epi[co * 2]; q1 = epi[co * 2 + 1].x; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long long o = (long long)co * HW + (long long)(2 * sy + (q >> 1)) * W + 2 * sx + (q & 1);
                float v = acc[q][m][r];
                v += q0.x;
                if (tensors) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_pre, (unsigned)o * 4u, 0, 0));
                v += q0.y; v *= q0.z; v += q0.w;
                v = v > 0.f ? v : v * slope;
                v *= q1;
                if (tensors) {
                    v = a1 * v + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r1, (unsigned)o * 4u, 0, 0));
                    v = a2 * v + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r2, (unsigned)o * 4u, 0, 0));
                }
                p.y[(long long)b * p.y_bs + o] = v;
            }
        }
}

template <int MR>
int launch_conv_up2(const BfsrConvArgs& a, hipStream_t st)
{
    constexpr int LDS_T = (8 * 16 * MR * 32 + 8 * 6 * 34) * 4, LDS_K = (8 * 9 * MR * 32 + 8 * 10 * 66) * 4;
    constexpr int LDS = LDS_T > LDS_K ? LDS_T : LDS_K;
    const int Hs = a.H / 2, Ws = a.W / 2;
    const int tiles_x = (Ws + 31) / 32, tiles_y = (Hs + 3) / 4;
    const int groups = ((a.Cout + 31) / 32 + MR - 1) / MR;
    const long long nblk = (long long)tiles_x * tiles_y * groups * a.B;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return -1;
    static std::atomic<unsigned long long> lds_done{0};
    if (bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_up2_kernel<MR>), LDS, lds_done) != 0) return -1;
    hipLaunchKernelGGL((conv_up2_kernel<MR>), dim3((unsigned)nblk), dim3(256), LDS, st, a, tiles_x, tiles_x * tiles_y, groups);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" long long bfsr_conv_packed_size(int Cout, int Cin, int KS, int mtile)
{
    const int cin_pad = (Cin + CIN_ALIGN - 1) / CIN_ALIGN * CIN_ALIGN;
    const int groups = ((Cout + 31) / 32 + mtile - 1) / mtile;
    return (long long)groups * cin_pad * KS * KS * mtile * 32;
}

extern "C" int bfsr_pack_conv_weight_taps(const float* w, int Cout, int Cin, int T, int mtile, float* packed)
{
    // w [Cout][Cin][T] -> [cout_group][cin_pad][T][mtile*32], zero padded
    if (T < 1 || mtile < 1) return -1;
    const int cin_pad = (Cin + CIN_ALIGN - 1) / CIN_ALIGN * CIN_ALIGN;
    const int MW = mtile * 32;
    const int groups = ((Cout + 31) / 32 + mtile - 1) / mtile;
    const long long n = (long long)groups * cin_pad * T * MW;
    for (long long i = 0; i < n; ++i) packed[i] = 0.f;
    for (int co = 0; co < Cout; ++co) {
        const int g = co / MW, m = co % MW;
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < T; ++t)
                packed[(((long long)g * cin_pad + ci) * T + t) * MW + m] = w[((long long)co * Cin + ci) * T + t];
    }
    return 0;
}

extern "C" long long bfsr_conv_packed_size_taps(int Cout, int Cin, int T, int mtile)
{
    const int cin_pad = (Cin + CIN_ALIGN - 1) / CIN_ALIGN * CIN_ALIGN;
    const int groups = ((Cout + 31) / 32 + mtile - 1) / mtile;
    return (long long)groups * cin_pad * T * mtile * 32;
}

extern "C" int bfsr_conv2d_up2(const BfsrConvArgs* a, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || !a->x || !a->w || !a->y || a->w2) return -1;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || (a->H & 1) || (a->W & 1) || a->Cin <= 0 || a->Cout <= 0) return -1;
    // 32-bit buffer offsets: every view read through a buffer descriptor must be < 2 GiB per batch element (the output
    // itself is written with 64-bit addressing and may be larger)
    if ((long long)a->Cin * (a->H / 2) * (a->W / 2) * 4 >= (1LL << 31)) return -1;
    if (a->x2 && (long long)a->Cin2 * a->H * a->W * 4 >= (1LL << 31)) return -1;
    if ((a->pre_add || a->res1 || a->res2) && (long long)a->Cout * a->H * a->W * 4 >= (1LL << 31)) return -1;
    if (a->mtile == 2) return launch_conv_up2<2>(*a, st);
    if (a->mtile == 1) return launch_conv_up2<1>(*a, st);
    return -1;
}

extern "C" int bfsr_pack_conv_weight(const float* w, int Cout, int Cin, int KS, int mtile, float* packed)
{
    if ((KS != 1 && KS != 3) || mtile < 1) return -1;
    const int cin_pad = (Cin + CIN_ALIGN - 1) / CIN_ALIGN * CIN_ALIGN;
    const int MW = mtile * 32, T = KS * KS;
    const int groups = ((Cout + 31) / 32 + mtile - 1) / mtile;
    const long long n = (long long)groups * cin_pad * T * MW;
    for (long long i = 0; i < n; ++i) packed[i] = 0.f;
    for (int co = 0; co < Cout; ++co) {
        const int g = co / MW, m = co % MW;
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < T; ++t)
                packed[(((long long)g * cin_pad + ci) * T + t) * MW + m] = w[((long long)co * Cin + ci) * T + t];
    }
    return 0;
}

extern "C" int bfsr_conv2d(const BfsrConvArgs* a, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || !a->x || !a->w || !a->y) return -1;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->Cin <= 0 || a->Cout <= 0) return -1;
    if (a->in_shift < 0 || a->in_shift > 4) return -1;
    // 32-bit buffer offsets: views read through a buffer descriptor must be < 2 GiB per batch element
    if ((long long)a->Cin * (a->H >> a->in_shift) * (a->W >> a->in_shift) * 4 >= (1LL << 31)) return -1;
    if ((a->pre_add || a->res1 || a->res2) && (long long)a->Cout * a->H * a->W * 4 >= (1LL << 31)) return -1;
    if (a->in_shift && (((a->H >> a->in_shift) << a->in_shift) != a->H || ((a->W >> a->in_shift) << a->in_shift) != a->W))
        return -1;
    // variant = (NR rows per wave, CK channels per LDS stage).  auto: NR=4 for big grids, NR=2 otherwise;
    // a->tune = NR*100 + CK overrides (used by tools/conv_bench.py to pick the table below).
    // auto tile height from the grid size (measured on MI355X, tools/conv_bench.py): 16x32 pixel tiles (NR=4) only when
    // the launch still fills the chip for >= 4 rounds of 2 workgroups/CU, 8x32 (NR=2) by default, 4x32 (NR=1) when even
    // that leaves fewer than ~5 workgroups per CU (the RRDB / level-3 convs at bench size).
    const long long groups_ = ((a->Cout + 31) / 32 + a->mtile - 1) / a->mtile;
    const long long tiles4 = (long long)((a->W + 31) / 32) * ((a->H + 15) / 16) * a->B;
    const long long tiles2 = (long long)((a->W + 31) / 32) * ((a->H + 7) / 8) * a->B;
    int NR = 2, CK = a->KS == 3 ? 8 : 16;
    if (tiles4 * groups_ >= 4 * 512 && a->mtile <= 2) NR = 4;
    else if (a->KS == 3 && a->mtile <= 2 && tiles2 * groups_ < 1280) NR = 1;
    if (a->tune > 0) { NR = a->tune / 100; CK = a->tune % 100; }
    if (a->w2) {
        // fused 3x3 -> 1x1 (coupling nets: flow.Conv2d 3x3 + ReLU -> flow.Conv2d 1x1 + ReLU); one cout group only
        if (a->KS != 3 || a->mtile != 2 || a->Cout > 64 || a->C2 > 64 || a->C2 <= 0 || a->res1 || a->res2) return -1;
        return launch_conv<3, 2, 2, 8, true>(*a, st);
    }
    const int key = a->KS * 10000 + a->mtile * 1000 + NR * 100 + CK;
    switch (key) {
#define V(KS_, MR_, NR_, CK_) case KS_ * 10000 + MR_ * 1000 + NR_ * 100 + CK_: return launch_conv<KS_, MR_, NR_, CK_>(*a, st);
        V(3, 1, 1, 8) V(3, 1, 2, 8) V(3, 1, 4, 8)
        V(3, 2, 1, 8) V(3, 2, 2, 8) V(3, 2, 4, 8)
        V(3, 3, 2, 8)
        V(1, 1, 2, 16) V(1, 1, 4, 16) V(1, 2, 2, 16) V(1, 2, 4, 16) V(1, 3, 2, 16)
#undef V
        default: break;
    }
    if (a->mtile == 3 && a->tune == 0) {     // only NR=2 variants exist for 3 M-tiles
        if (a->KS == 3) return launch_conv<3, 3, 2, 8>(*a, st);
        if (a->KS == 1) return launch_conv<1, 3, 2, 16>(*a, st);
    }
    return -1;
}
