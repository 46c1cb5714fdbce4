// coupling_wide.hip -- the sequential part of a conditional-affine FlowStep at the WIDE level (C = 96 flow channels: level 3 of both
// SRFlow-LP models; FlowAffineCouplingsAblation.py:57-97, FlowStep.py:88-129) as TWO streaming kernels (round 6):
//
//   bfsr_coupling_wide_head: hid = relu(AN2(W2 . relu(AN0(conv3x3(z1; W0z) + pre_aff))))      z1 (48 ch), pre_aff (64 ch), hid (64 ch): h2 tensors
//   bfsr_coupling_wide_tail: h_aff = (conv3x3(hid; W4) + b4) * exp(3 logs4)  (Conv2dZeros 64 -> 96, flow.py:68-83), then -- as the conv's
//       epilogue -- the step's whole pointwise chain with h_aff taken from the accumulators (the semantics of bfsr_flow_pointwise), the C x C
//       invertible 1x1 on v_mfma_f32_32x32x2_f32; optionally the first 48 output channels are ALSO written as an h2 tensor: the next step's z1.
//
// Why not the pair of levels 1 / 2 (coupling.hip / coupling_tail.hip): their weights are resident in LDS; here fAffine.0 on 48 z1 channels is
// 110 KB and fAffine.4 221 KB as fp16 pairs.  So both kernels STREAM their weights with the input, 16-channel chunk by chunk, exactly as
// conv3x3_h2x_kernel does (conv_h2s.hip: four LDS-DMA loader waves, two LDS stages, ONE barrier per chunk, a pipeline step = one tap) -- same
// arithmetic (two-term fp16 split, lo*hi + hi*lo + hi*hi per operand pair, same chunk -> tap -> product order per output pixel, so the conv
// results are bit-identical to that kernel's) -- but a workgroup owns ALL output channels of an 8-row x 32-pixel tile (64 = 2 M tiles in the
// head, 96 = 3 in the tail; one row per compute wave), which is what lets the 1x1 be chained in registers (head) and the pointwise chain see
// every h_aff channel of a pixel (tail).  8-row tiles: the level is small (config 2: 8 x 80^2 = 240 tiles on 256 CUs).
//
// Before this file a step at this level was five launches (z1 pack, conv_h2x 48 -> 64, the 1x1-only coupling_head, conv_h2x 64 -> 96,
// flow_pointwise_mfma<96>): ~132 us per step at 8 x 80^2, each filling < 1/4 of the chip.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "../../include/bfsr_hip.h"
#include "launch_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

namespace {

constexpr int NW = 8, NLW = 4;                  // compute waves (one tile row each), loader waves
constexpr int TH = 8, PW = 34, NPOS = (TH + 2) * PW, NG = 6, NPOSP = NG * 64;
constexpr int SUB = NPOSP * 16;                 // bytes of one (plane, k half) sub-image: 8 channels of every tile position (padded to 6 x 64)
constexpr int X_IN = 4 * SUB;                   // 24 576
constexpr unsigned OOB = 0x80000000u;
static_assert(NPOS <= NPOSP, "position groups");

// XM = 32-channel M tiles of the conv's output per workgroup.  Weights of one 16-channel chunk: [plane hi,lo][tap = dx*3 + dy][m tile][k half][32][8] fp16
template <int XM> struct WGeo {
    static constexpr int WPL = 9 * 1024 * XM;
    static constexpr int W = 2 * WPL;           // 36 864 | 55 296
    static constexpr int STAGE = X_IN + W;      // 61 440 | 79 872
    static constexpr int RING = 2 * STAGE;      // 122 880 | 159 744
};
constexpr int HEAD_W2B = 4 * 2 * 2 * 64 * 16;   // the 1x1 of the head, resident: [chunk][plane][k half][64 rows][8] fp16 = 16 384
constexpr int HEAD_PB = 2 * 64 * 8;             // epi0, epi2: [64] {shift, scale}
constexpr int HEAD_LDS = WGeo<2>::RING + HEAD_W2B + HEAD_PB;       // 140 288
constexpr int CF = 96, CFN = 48;                // flow channels of the level, z1 channels
constexpr int TAIL_NPAR = 4 * CF;               // bias, post_scale, an_bias, an_escale
constexpr int TAIL_LDS = WGeo<3>::RING + TAIL_NPAR * 4;            // 161 280
static_assert(HEAD_LDS <= 160 * 1024 && TAIL_LDS <= 160 * 1024, "LDS budget");

struct WItem { int b, x0, y0; };
__device__ __forceinline__ WItem wdecode(int it, int tiles_x, int tiles_y)
{
    WItem r;
    int t = it;
    const int ty = t % tiles_y; t /= tiles_y;
    r.x0 = (t % tiles_x) * 32; r.y0 = ty * TH; r.b = t / tiles_x;
    return r;
}

__device__ __forceinline__ void split2(float v, _Float16& h, _Float16& l)
{
    const float hf = bfsr::pin_f16(v);
    h = (_Float16)hf;
    l = (_Float16)(v - hf);
}

// ---- loader waves: LDS-DMA only (conv3x3_h2x_kernel's protocol).  Loader `ld` owns sub-image `ld` (plane ld>>1, k half ld&1) of every
// position group and every fourth 1-KiB piece of the chunk's weights; chunk k+1 (across item boundaries) is staged while chunk k is in
// the matrix pipe; `s_waitcnt vmcnt(0)` + the barrier hand a stage over.
template <int XM>
__device__ __forceinline__ void wide_loader_wave(const unsigned short* __restrict__ x, long long x_bs, int Cin, const unsigned short* __restrict__ w,
                                                 unsigned char* smem, int ld, int lane, int slot, int G, int nitems, int tiles_x, int tiles_y, int H, int W)
{
    constexpr int X_W = WGeo<XM>::W, X_STAGE = WGeo<XM>::STAGE;
    const unsigned HW16 = (unsigned)(H * W) * 16u;
    const int nchunk = Cin >> 4;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(w), 0, (unsigned)(nchunk * X_W), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_in;
    unsigned vg[NG];
    auto lsetup = [&](const WItem& it) {
        rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(x + (long long)it.b * x_bs), 0, (unsigned)(Cin >> 3) * 2u * HW16, 0x00020000);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int pos = g * 64 + lane;
            const int r = pos / PW, c = pos - r * PW;
            const int gy = it.y0 + r - 1, gx = it.x0 + c - 1;
            const bool ok = pos < NPOS && gy >= 0 && gy < H && gx >= 0 && gx < W;
            vg[g] = ok ? (unsigned)(gy * W + gx) * 16u : OOB;            // out of range -> the DMA writes zeros (= the padding)
        }
    };
    auto lstage = [&](int k, int buf) {
        unsigned char* base = smem + buf * X_STAGE;
        const unsigned soff = (unsigned)((2 * k + (ld & 1)) * 2 + (ld >> 1)) * HW16;          // octet 2k + k half, plane ld>>1
#pragma unroll
        for (int g = 0; g < NG; ++g)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(base + ld * SUB + g * 1024), 16, vg[g], soff, 0, 0);
        const unsigned wsoff = (unsigned)k * (unsigned)X_W;
#pragma unroll
        for (int j = 0; j < (X_W / 1024 + NLW - 1) / NLW; ++j) {
            const int piece = ld + j * NLW;
            if (piece < X_W / 1024)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(base + X_IN + piece * 1024), 16,
                                                         (unsigned)lane * 16u + (unsigned)piece * 1024u, wsoff, 0, 0);
        }
    };
    int it = slot;
    lsetup(wdecode(it, tiles_x, tiles_y));
    lstage(0, 0);
    int buf_ = 0;
    while (true) {
        const int nxt = it + G;
        const bool has_next = nxt < nitems;
        for (int k = 0; k < nchunk; ++k) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (k + 1 < nchunk) lstage(k + 1, buf_ ^ 1);
            else if (has_next) { lsetup(wdecode(nxt, tiles_x, tiles_y)); lstage(0, buf_ ^ 1); }
            buf_ ^= 1;
        }
        if (!has_next) return;
        it = nxt;
    }
}

// ---- compute waves: the K loop of one item.  Wave w owns tile row w; a step = one tap: the row's two input planes + the tap's weight
// planes of XM tiles -> 3 * XM MFMAs; the fragments of tap t+1 are read while the MFMAs of tap t run.  One barrier per chunk, passed
// EARLY (before the chunk's last tap, whose fragments are already in registers).  `buf` = LDS stage of the item's first chunk (in / out).
template <int XM>
__device__ __forceinline__ void wide_kloop(const unsigned char* smem, int& buf, f32x16 (&acc)[XM], int wave, int lane, int nchunk)
{
    constexpr int X_WPL = WGeo<XM>::WPL, X_STAGE = WGeo<XM>::STAGE;
    const int l31 = lane & 31, lhi = lane >> 5;
    half8 bq[2][2], aq[2][2][XM];                                        // [buffer][plane] | [buffer][plane][m tile]
    auto load_step = [&](auto buf_, int st, int t) {
        constexpr int BUF = decltype(buf_)::value;
        const int dx = t / 3, dy = t - 3 * dx;
        const unsigned char* sIn = smem + st * X_STAGE;
        const unsigned char* inB = sIn + (lhi * NPOSP + (wave + dy) * PW + l31 + dx) * 16;
        const unsigned char* wA = sIn + X_IN + lane * 16 + t * (1024 * XM);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            bq[BUF][pl] = *reinterpret_cast<const half8*>(inB + pl * 2 * SUB);
#pragma unroll
            for (int m = 0; m < XM; ++m) aq[BUF][pl][m] = *reinterpret_cast<const half8*>(wA + pl * X_WPL + m * 1024);
        }
    };
    auto mfma_step = [&](auto buf_) {
        constexpr int BUF = decltype(buf_)::value;
#pragma unroll
        for (int m = 0; m < XM; ++m) {                                   // smallest terms first: w_lo*x_hi, w_hi*x_lo, w_hi*x_hi
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[BUF][1][m], bq[BUF][0], acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[BUF][0][m], bq[BUF][1], acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[BUF][0][m], bq[BUF][0], acc[m], 0, 0, 0);
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
#pragma unroll
    for (int m = 0; m < XM; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    auto chunk_body = [&](auto p_, auto q_, bool last) {                  // p_: buffer holding tap 0's fragments (already loaded)
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
            load_step(q_, buf, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(p_);
            __builtin_amdgcn_sched_barrier(0);
            load_step(p_, buf, t + 2);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(q_);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!last) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // tap 8's fragments have left stage `buf`
            __builtin_amdgcn_s_barrier();                                // chunk k+1 has landed in stage buf^1; stage buf is free again
            load_step(q_, buf ^ 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(p_);
        __builtin_amdgcn_sched_barrier(0);
        buf ^= 1;
    };
    __builtin_amdgcn_s_barrier();                                        // the item's first chunk has landed in stage `buf`
    load_step(I0(), buf, 0);
    int k = 0;
    for (; k + 2 <= nchunk; k += 2) {                                    // pairs of chunks: the fragment-buffer parity is static inside a pair
        chunk_body(I0(), I1(), false);
        chunk_body(I1(), I0(), k + 2 == nchunk);
    }
    if (k < nchunk) chunk_body(I0(), I1(), true);
}

// =====================================================================================================================
// head: conv3x3 48 -> 64 over the h2 tensor z1 (+ the hoisted partial pre_aff, an h2 view read as hi + lo), ActNorm + ReLU, the 1x1
// chained in registers (coupling_head_kernel's scheme: the accumulator layout of stage 1 IS a valid B operand of the next GEMM when
// W2's K axis is packed in that order), ActNorm + ReLU, hid as an h2 tensor.
__global__ __launch_bounds__((NW + NLW) * 64, 1) void coupling_wide_head_kernel(BfsrWideHeadArgs p, int tiles_x, int tiles_y, int nitems)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sW2 = smem + WGeo<2>::RING;
    const float2* sP = reinterpret_cast<const float2*>(sW2 + HEAD_W2B);  // [0..63] epi0, [64..127] epi2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int G = gridDim.x;
    const int slot = (int)bfsr::xcd_order(blockIdx.x, (unsigned)G);
    if (slot >= nitems) return;
    const int H = p.H, W = p.W;
    const long long HW = (long long)H * W;
    {
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(p.w2);
        uint4* dst = reinterpret_cast<uint4*>(sW2);
        for (int i = tid; i < HEAD_W2B / 16; i += (NW + NLW) * 64) dst[i] = src[i];
        float2* dp = reinterpret_cast<float2*>(sW2 + HEAD_W2B);
        for (int i = tid; i < 128; i += (NW + NLW) * 64) {
            const float4 q = reinterpret_cast<const float4*>(i < 64 ? p.epi0 : p.epi2)[i & 63];
            dp[i] = make_float2(q.x, q.y);
        }
        __syncthreads();
    }
    if (wave >= NW) {
        wide_loader_wave<2>(p.z1, p.z1_bs, p.Cz, p.w0, smem, wave - NW, lane, slot, G, nitems, tiles_x, tiles_y, H, W);
        return;
    }
    const int nchunk = p.Cz >> 4;
    const float s0 = p.acc_scale0, s2 = p.acc_scale2;
    int buf = 0;
    unsigned bad = 0u;
    for (int it = slot; it < nitems; it += G) {
        const WItem cur = wdecode(it, tiles_x, tiles_y);
        const int gy = cur.y0 + wave, gx = cur.x0 + l31;
        const bool con = gy < H && gx < W;
        // this lane's 32 pre_aff values in accumulator order ([m][r]: channel m*32 + (r&3) + 8(r>>2) + 4*lhi = elements 4*lhi..4*lhi+3 of octet 4m + g):
        // one 8-byte load per (octet, plane), issued here, landing under the K loop
        half4 ph[2][4], pl[2][4];
        {
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.pre + (long long)cur.b * p.pre_bs), 0,
                                                                                (unsigned)(8 * 2 * HW * 16), 0x00020000);
            const unsigned vo = con ? (unsigned)(((long long)gy * W + gx) * 16 + lhi * 8) : OOB;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned so = (unsigned)((4 * m + g) * 2) * (unsigned)(HW * 16);
                    ph[m][g] = __builtin_bit_cast(half4, __builtin_amdgcn_raw_buffer_load_b64(rp, vo, so, 0));
                    pl[m][g] = __builtin_bit_cast(half4, __builtin_amdgcn_raw_buffer_load_b64(rp, vo, so + (unsigned)(HW * 16), 0));
                }
        }
        f32x16 acc[2];
        wide_kloop<2>(smem, buf, acc, wave, lane, nchunk);

        // ---- epilogue 1 in registers: weight scale, + pre_aff, ActNorm, ReLU; the result IS the B operand of the 1x1 (K order = accumulator order)
        half8 b2[4][2];                                     // chunk c = (m, half): registers 8*half .. 8*half+7 of tile m; [plane]
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 pa = *reinterpret_cast<const float4*>(&sP[m * 32 + 8 * g + 4 * lhi]);          // {shift, scale} x 2
                const float4 pb = *reinterpret_cast<const float4*>(&sP[m * 32 + 8 * g + 4 * lhi + 2]);
                const float sh[4] = {pa.x, pa.z, pb.x, pb.z}, sc[4] = {pa.y, pa.w, pb.y, pb.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float pre = (float)ph[m][g][e] + (float)pl[m][g][e];
                    float v = ((acc[m][r] * s0 + pre) + sh[e]) * sc[e];
                    v = v > 0.f ? v : 0.f;
                    bad |= (unsigned)!(v < 32768.f);
                    _Float16 h, l;
                    split2(v, h, l);
                    b2[m * 2 + (g >> 1)][0][(g & 1) * 4 + e] = h;
                    b2[m * 2 + (g >> 1)][1][(g & 1) * 4 + e] = l;
                }
            }
        f32x16 acc2[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[m][r] = 0.f;
        {
            half8 fa[2][2][2];                              // [buffer][m][plane]
            auto load_a2 = [&](int c, half8 (&af)[2][2]) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int pq = 0; pq < 2; ++pq)
                        af[m][pq] = *reinterpret_cast<const half8*>(sW2 + (((c * 2 + pq) * 2 + lhi) * 64 + m * 32 + l31) * 16);
            };
            load_a2(0, fa[0]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c + 1 < 4) load_a2(c + 1, fa[(c + 1) & 1]);
#pragma unroll
                for (int m = 0; m < 2; ++m) {               // smallest terms first: w_lo*x_hi, w_hi*x_lo, w_hi*x_hi
                    acc2[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[c & 1][m][1], b2[c][0], acc2[m], 0, 0, 0);
                    acc2[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[c & 1][m][0], b2[c][1], acc2[m], 0, 0, 0);
                    acc2[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[c & 1][m][0], b2[c][0], acc2[m], 0, 0, 0);
                }
            }
        }
        // ---- epilogue 2: weight scale, ActNorm, ReLU; v_permlane32_swap pairs the half-waves so that every lane holds two complete channel
        // octets of its pixel per M tile; split and store both planes of the h2 tensor (16 bytes each)
        {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.hid + (long long)cur.b * p.hid_bs, 0, (unsigned)(8 * 2 * HW * 16), 0x00020000);
            const unsigned vo = con ? (unsigned)(((long long)lhi * 2 * HW + (long long)gy * W + gx) * 16) : OOB;     // out-of-image lanes: dropped
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float u[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 pa = *reinterpret_cast<const float4*>(&sP[64 + m * 32 + 8 * g + 4 * lhi]);
                    const float4 pb = *reinterpret_cast<const float4*>(&sP[64 + m * 32 + 8 * g + 4 * lhi + 2]);
                    const float sh[4] = {pa.x, pa.z, pb.x, pb.z}, sc[4] = {pa.y, pa.w, pb.y, pb.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = (acc2[m][4 * g + e] * s2 + sh[e]) * sc[e];
                        u[4 * g + e] = v > 0.f ? v : 0.f;
                    }
                }
#pragma unroll
                for (int qd = 0; qd < 2; ++qd) {
                    float o[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float lo = u[8 * qd + i], hi = u[8 * qd + 4 + i];
                        // u[] is VALU output (compiler-visible: the MFMA results are complete); the swap needs 2 wait states behind a VALU
                        // write of its operands, which hipcc cannot see inside the string
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
                        o[i] = lo; o[4 + i] = hi;
                    }
                    half8 h8, l8;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        _Float16 h, l;
                        split2(o[i], h, l);
                        h8[i] = h; l8[i] = l;
                        bad |= (unsigned)!(o[i] < 32768.f);
                    }
                    const unsigned so = (unsigned)((m * 4 + qd * 2) * 2) * (unsigned)(HW * 16);      // octet m*4 + qd*2 (+ lhi through the VGPR offset)
                    bfsr::store_b128(rs, __builtin_bit_cast(u32x4, h8), vo, so);
                    bfsr::store_b128(rs, __builtin_bit_cast(u32x4, l8), vo, so + (unsigned)(HW * 16));
                }
            }
        }
    }
    if (p.flag && __any((int)bad)) { if (lane == 0) atomicOr(p.flag, 1u); }
}

// =====================================================================================================================
// tail: Conv2dZeros 64 -> 96 over the h2 tensor hid, then the step's pointwise chain.  Channel bookkeeping of the epilogue:
//   * conv accumulators: lane (l31, lhi) holds h_aff channels 32m + 8g + 4*lhi + e (register 4g + e of tile m) of pixel (row = wave, column = l31):
//     the (shift, scale) pairs of z2 channels j = 16m + 4g + 2*lhi + p (p = e >> 1) are complete in one lane -- no exchange;
//   * the flow state enters the matvec as the B operand of v_mfma_f32_32x32x2_f32 (lane = pixel, K index = lhi): step kk = 0..47 contracts
//     channel c(kk, lhi) = 4*(kk >> 1) + 2*lhi + (kk & 1), which makes lane (l31, lhi) own exactly the z2 channels whose (shift, scale) it holds and
//     whole (shift, scale) QUADS of h_ft; `wperm` is W with its K axis in that order: wperm[(2*kk + lhi)*96 + i] = W[i][c(kk, lhi)];
//   * the result comes out in accumulator order again (channel 32mo + 8g + 4*lhi + e).
template <int REV>
__global__ __launch_bounds__((NW + NLW) * 64, 1) void coupling_wide_tail_kernel(BfsrWideTailArgs q, int tiles_x, int tiles_y, int nitems)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sPar = reinterpret_cast<float*>(smem + WGeo<3>::RING);        // [0] bias, [96] post_scale, [192] an_bias, [288] an_escale
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int G = gridDim.x;
    const int slot = (int)bfsr::xcd_order(blockIdx.x, (unsigned)G);
    if (slot >= nitems) return;
    const int H = q.H, W = q.W;
    const long long HW = (long long)H * W;
    {
        for (int i = tid; i < TAIL_NPAR; i += (NW + NLW) * 64) {
            float v;
            if (i < CF) v = q.bias[i];
            else if (i < 2 * CF) v = q.post_scale[i - CF];
            else if (i < 3 * CF) v = q.an_bias ? q.an_bias[i - 2 * CF] : 0.f;
            else v = q.an_bias ? q.an_escale[i - 3 * CF] : 1.f;
            sPar[i] = v;
        }
        __syncthreads();
    }
    if (wave >= NW) {
        wide_loader_wave<3>(q.hid, q.hid_bs, 64, q.w, smem, wave - NW, lane, slot, G, nitems, tiles_x, tiles_y, H, W);
        return;
    }
    const float eps = q.eps, as = q.acc_scale;
    // IEEE division and expf are ~10 instructions each and all eight compute waves are in this epilogue at once: quotients are formed as
    // v_rcp_f32 + one Newton step on the quotient (<= 1 ulp), exp as v_exp_f32 of x * log2(e) (coupling_tail_kernel's choices)
    auto fdiv = [](float a, float b) {
        const float r = __builtin_amdgcn_rcpf(b);
        const float qt = a * r;
        return fmaf(fmaf(-b, qt, a), r, qt);
    };
    const bool hf = q.h_ft != nullptr, has_w = q.wperm != nullptr;
    const float* __restrict__ wp = q.wperm;
    int buf = 0;
    unsigned bad = 0u, big = 0u;
    for (int it = slot; it < nitems; it += G) {
        const WItem cur = wdecode(it, tiles_x, tiles_y);
        f32x16 acc[3];
        wide_kloop<3>(smem, buf, acc, wave, lane, 4);

        const int gy = cur.y0 + wave, gx = cur.x0 + l31;
        const bool con = gy < H && gx < W;
        const unsigned pix = (unsigned)(gy * W + gx);
        int po = 0;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(po));                                     // per-item opaque offset: the parameter reads must not be hoisted into SGPRs
#endif
        const float* sp = sPar + po;
        // plane strides as per-item opaque scalars: as loop invariants hipcc hoists the ~250 distinct `channel * HW * 4` scalar offsets of the
        // epilogue out of the item loop and spills them (157 SGPRs -> VGPR lanes)
        unsigned hw4 = (unsigned)(HW * 4), hw16 = (unsigned)(HW * 16);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+s"(hw4), "+s"(hw16));
#endif
        // ---- the flow state of this lane's pixel in B-operand order
        float xs[CFN];
        {
            const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.z_in + (long long)cur.b * q.z_in_bs), 0,
                                                                                (unsigned)(CF * HW * 4), 0x00020000);
            const unsigned vz = con ? (unsigned)(((long long)pix + 2LL * lhi * HW) * 4) : OOB;
#pragma unroll
            for (int kk = 0; kk < CFN; ++kk)
                xs[kk] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, vz, (unsigned)(4 * (kk >> 1) + (kk & 1)) * hw4, 0));
        }
        // ---- this step's self-conditional affine on z2 (h_aff from the accumulators).  The fences bound what hipcc's scheduler may hoist: without
        // them every LDS parameter read and every scalar offset of the epilogue moved to its top and the z loads were spilled one by one
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = 32 * m + 8 * g + 4 * lhi;                  // (lhi is per-lane: the LDS address is a VGPR expression)
                const float4 bb = *reinterpret_cast<const float4*>(&sp[ch]), pp = *reinterpret_cast<const float4*>(&sp[CF + ch]);
                const float b4[4] = {bb.x, bb.y, bb.z, bb.w}, p4[4] = {pp.x, pp.y, pp.z, pp.w};
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const float sh = (acc[m][4 * g + 2 * pr] * as + b4[2 * pr]) * p4[2 * pr];
                    const float raw = (acc[m][4 * g + 2 * pr + 1] * as + b4[2 * pr + 1]) * p4[2 * pr + 1];
                    const float t = 1.f + __expf(-(raw + 2.f));
                    const int kk = 24 + 2 * (4 * m + g) + pr;
                    if constexpr (REV) xs[kk] = xs[kk] * fdiv(t, fmaf(eps, t, 1.f)) - sh;     // z2 / scale - shift, 1/scale = t / (1 + eps t)
                    else xs[kk] = (xs[kk] + sh) * (fdiv(1.f, t) + eps);
                }
            }
#if defined(__HIP_DEVICE_COMPILE__)
            // pin the phase's results HERE (machine-sinking moved the arithmetic of all three phases below the parameter reads of all three:
            // 96 live parameter registers next to 48 accumulators and 48 state values)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(xs[24 + 8 * m + j]));
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hf ? q.h_ft + (long long)cur.b * q.h_ft_bs : q.z_in), 0,
                                                                            hf ? (unsigned)(2 * CF * HW * 4) : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rzo = __builtin_amdgcn_make_buffer_rsrc(q.z_out + (long long)cur.b * q.z_out_bs, 0, (unsigned)(CF * HW * 4), 0x00020000);
        if constexpr (REV) {
            if (hf) {                                                     // z = z / scaleFt - shiftFt on all 96 channels: quad 2t + lhi = {shift, raw scale} of channels c(2t), c(2t+1)
#pragma unroll
                for (int tb = 0; tb < 24; tb += 8) {
                    float4 fq[8];
                    if (q.h_ft_fmt == 1) {
                        const unsigned vq = con ? (unsigned)(((long long)pix + (long long)lhi * HW) * 16) : OOB;
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            fq[t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rf, vq, (unsigned)(2 * (tb + t)) * hw16, 0));
                    } else {
                        const unsigned vq = con ? (unsigned)(((long long)pix + 4LL * lhi * HW) * 4) : OOB;
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            fq[t].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, vq, (unsigned)(8 * (tb + t) + 0) * hw4, 0));
                            fq[t].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, vq, (unsigned)(8 * (tb + t) + 1) * hw4, 0));
                            fq[t].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, vq, (unsigned)(8 * (tb + t) + 2) * hw4, 0));
                            fq[t].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, vq, (unsigned)(8 * (tb + t) + 3) * hw4, 0));
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const float ta = 1.f + __expf(-(fq[t].y + 2.f)), tb2 = 1.f + __expf(-(fq[t].w + 2.f));
                        xs[2 * (tb + t)] = xs[2 * (tb + t)] * fdiv(ta, fmaf(eps, ta, 1.f)) - fq[t].x;
                        xs[2 * (tb + t) + 1] = xs[2 * (tb + t) + 1] * fdiv(tb2, fmaf(eps, tb2, 1.f)) - fq[t].z;
                    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                    for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(xs[2 * tb + j]));
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if (has_w) {                                               // the NEXT step's ActNorm (identity if absent)
#pragma unroll
            for (int t = 0; t < 24; ++t) {
                const float2 ab = *reinterpret_cast<const float2*>(&sp[2 * CF + 4 * t + 2 * lhi]), ae = *reinterpret_cast<const float2*>(&sp[3 * CF + 4 * t + 2 * lhi]);
                xs[2 * t] = (xs[2 * t] + ab.x) * ae.x;
                xs[2 * t + 1] = (xs[2 * t + 1] + ab.y) * ae.y;
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : "+v"(xs[2 * t]), "+v"(xs[2 * t + 1]));
#endif
                if ((t & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!has_w) {
            // no matvec (forward, last step of the level): the state leaves in B-operand order, channel c(kk, lhi)
            const unsigned vz = con ? (unsigned)(((long long)pix + 2LL * lhi * HW) * 4) : OOB;
#pragma unroll
            for (int kk = 0; kk < CFN; ++kk) {
                bad |= (unsigned)!(fabsf(xs[kk]) < 3.0e38f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, xs[kk]), rzo, vz, (unsigned)(4 * (kk >> 1) + (kk & 1)) * hw4, 0);
            }
            continue;
        }
        // ---- the invertible 1x1 on the fp32 matrix pipe: [96 x 96] x [96 x 32 pixels].  The A operands (W, L1 / L2 resident) are fetched four
        // K steps ahead in two register sets; the fences keep hipcc from hoisting all 144 loads to the top (310 spilled registers)
        f32x16 y[3];
#pragma unroll
        for (int mo = 0; mo < 3; ++mo)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[mo][r] = 0.f;
        {
            // (buffer loads: one per-lane offset + literal offsets; as `wp[...]` hipcc kept 144 64-bit per-lane addresses alive across the item loop)
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, (unsigned)(CF * CF * 4), 0x00020000);
            const unsigned vw = (unsigned)((lhi * CF + l31) * 4);
            float wa[2][12];
            auto wload = [&](int gk, float (&wv)[12]) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo)
                        wv[3 * j + mo] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, vw, (unsigned)((2 * (4 * gk + j) * CF + 32 * mo) * 4), 0));
            };
            wload(0, wa[0]);
#pragma unroll
            for (int gk = 0; gk < CFN / 4; ++gk) {
                if (gk + 1 < CFN / 4) wload(gk + 1, wa[(gk + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mo = 0; mo < 3; ++mo) y[mo] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[gk & 1][3 * j + mo], xs[4 * gk + j], y[mo], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- behind the matvec, in accumulator order, one M tile at a time: reverse: ActNorm inverse; forward: the next step's feature-conditional
        // affine; the store; and (z1h) channels 0..47 once more as an h2 tensor (the next step's z1): octets 4mo + 2qd + lhi after one half-wave
        // swap per register pair
        const unsigned vzo = con ? (unsigned)(((long long)pix + 4LL * lhi * HW) * 4) : OOB;       // out-of-image lanes: dropped by the range check, no branch
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(q.z1h ? q.z1h + (long long)cur.b * q.z1h_bs : nullptr, 0,
                                                                            q.z1h ? (unsigned)(6 * 2 * HW * 16) : 0u, 0x00020000);
        const unsigned vh = (con && q.z1h) ? (unsigned)(((long long)lhi * 2 * HW + pix) * 16) : OOB;
#pragma unroll
        for (int mo = 0; mo < 3; ++mo) {
            float o[16];
            if constexpr (REV) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = 32 * mo + 8 * g + 4 * lhi;
                    const float4 ab = *reinterpret_cast<const float4*>(&sp[2 * CF + ch]), ae = *reinterpret_cast<const float4*>(&sp[3 * CF + ch]);
                    const float a4[4] = {ab.x, ab.y, ab.z, ab.w}, e4[4] = {ae.x, ae.y, ae.z, ae.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[4 * g + e] = y[mo][4 * g + e] * e4[e] - a4[e];
                }
            } else if (hf) {                                              // quad 16mo + 4g + 2*lhi + pr = {shift, raw scale} of channels 32mo + 8g + 4*lhi + 2pr, + 1
                float4 fq[8];
                if (q.h_ft_fmt == 1) {
                    const unsigned vq = con ? (unsigned)(((long long)pix + 2LL * lhi * HW) * 16) : OOB;
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        fq[t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rf, vq, (unsigned)(16 * mo + 4 * (t >> 1) + (t & 1)) * hw16, 0));
                } else {
                    const unsigned vq = con ? (unsigned)(((long long)pix + 8LL * lhi * HW) * 4) : OOB;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const unsigned c0 = (unsigned)(64 * mo + 16 * (t >> 1) + 4 * (t & 1));
                        fq[t].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, vq, (c0 + 0) * hw4, 0));
                        fq[t].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, vq, (c0 + 1) * hw4, 0));
                        fq[t].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, vq, (c0 + 2) * hw4, 0));
                        fq[t].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, vq, (c0 + 3) * hw4, 0));
                    }
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) {                             // t = 2g + pr -> registers 4g + 2pr, + 1
                    const float ta = 1.f + __expf(-(fq[t].y + 2.f)), tb2 = 1.f + __expf(-(fq[t].w + 2.f));
                    o[2 * t] = (y[mo][2 * t] + fq[t].x) * (fdiv(1.f, ta) + eps);
                    o[2 * t + 1] = (y[mo][2 * t + 1] + fq[t].z) * (fdiv(1.f, tb2) + eps);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = y[mo][r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                bad |= (unsigned)!(fabsf(o[r]) < 3.0e38f);                 // NaN / inf guard of the flow state
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[r]), rzo, vzo, (unsigned)(32 * mo + (r & 3) + 8 * (r >> 2)) * hw4, 0);
            }
            if (q.z1h && mo < 2) {
#pragma unroll
                for (int qd = 0; qd < 2 - mo; ++qd) {                     // (mo, qd) = (0, 0), (0, 1), (1, 0): channels 0..47
                    float v8[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float lo = o[8 * qd + i], hi = o[8 * qd + 4 + i];
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
                        v8[i] = lo; v8[4 + i] = hi;
                    }
                    half8 h8, l8;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        _Float16 h, l;
                        split2(v8[i], h, l);
                        h8[i] = h; l8[i] = l;
                        big |= (unsigned)!(fabsf(v8[i]) < 65504.f);
                    }
                    const unsigned so = (unsigned)((4 * mo + 2 * qd) * 2) * hw16;
                    bfsr::store_b128(rh, __builtin_bit_cast(u32x4, h8), vh, so);
                    bfsr::store_b128(rh, __builtin_bit_cast(u32x4, l8), vh, so + hw16);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (q.flag) {
        if (__any((int)bad)) { if (lane == 0) atomicOr(q.flag, 2u); }
        if (__any((int)big)) { if (lane == 0) atomicOr(q.flag, 1u); }
    }
}

inline unsigned short f16_bits(float v)
{
    const _Float16 h = (_Float16)v;
    unsigned short s;
    __builtin_memcpy(&s, &h, 2);
    return s;
}

template <class K, class A>
int launch_wide(K kernel, const A& a, int lds, std::atomic<unsigned long long>& done, hipStream_t st)
{
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + TH - 1) / TH;
    const long long nitems = (long long)tiles_x * tiles_y * a.B;
    if (nitems <= 0 || nitems > 0x7fffffffLL) return -1;
    int cus = bfsr::cu_count();
    if (cus <= 0) return -1;
    const long long grid = nitems < cus ? nitems : cus;                  // one persistent workgroup per CU
    if (bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), lds, done) != 0) return -1;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3((NW + NLW) * 64), lds, st, a, tiles_x, tiles_y, (int)nitems);
    return (int)hipGetLastError();
}

}  // namespace

// ---- host-side packing -----------------------------------------------------------------------------------------------
// 3x3 weights [Cout][Cin][3][3] (Cout = 32 * XM, one output-channel group) as conv3x3_h2x_kernel stages them:
// fp16 [16-channel chunk][plane hi,lo][tap = dx*3 + dy][m tile][k half][32][8] of w * scale.
extern "C" long long bfsr_coupling_wide_conv_packed_size(int Cout, int Cin)
{
    if (Cout <= 0 || (Cout & 31) || Cout > 96 || Cin <= 0 || (Cin & 15)) return -1;
    return (long long)(Cin / 16) * 2 * 9 * (Cout / 32) * 2 * 32 * 8;     // fp16 elements
}

extern "C" int bfsr_pack_coupling_wide_conv(const float* w, int Cout, int Cin, float scale, unsigned short* packed)
{
    const long long n = bfsr_coupling_wide_conv_packed_size(Cout, Cin);
    if (!w || !packed || n <= 0 || !(scale > 0.f)) return -1;
    const int xm = Cout / 32;
    for (long long i = 0; i < n; ++i) packed[i] = 0;
    const long long plane = 9LL * xm * 2 * 32 * 8;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx) {
                    const float v = w[((long long)co * Cin + ci) * 9 + dy * 3 + dx] * scale;
                    const _Float16 h = (_Float16)v;
                    const _Float16 l = (_Float16)(v - (float)h);
                    const long long base = (long long)(ci / 16) * 2;
                    const long long in = ((((long long)(dx * 3 + dy) * xm + co / 32) * 2 + (ci % 16) / 8) * 32 + co % 32) * 8 + ci % 8;
                    packed[(base + 0) * plane + in] = f16_bits((float)h);
                    packed[(base + 1) * plane + in] = f16_bits((float)l);
                }
    return 0;
}

// the head's 1x1 [64][64] (fAffine.2) in the K order of the 3x3's accumulators: the last four chunks of bfsr_pack_coupling_head's image
extern "C" long long bfsr_coupling_wide_w2_packed_size(void) { return HEAD_W2B / 2; }

extern "C" int bfsr_pack_coupling_wide_w2(const float* w2, float scale2, unsigned short* packed)
{
    if (!w2 || !packed || !(scale2 > 0.f)) return -1;
    for (int i = 0; i < HEAD_W2B / 2; ++i) packed[i] = 0;
    for (int c = 0; c < 4; ++c) {
        const int m = c >> 1, hf = c & 1;
        for (int half = 0; half < 2; ++half)
            for (int row = 0; row < 64; ++row)
                for (int e = 0; e < 8; ++e) {
                    const int r = hf * 8 + e;
                    const int ch = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float v = w2[(long long)row * 64 + ch] * scale2;
                    const _Float16 h = (_Float16)v;
                    const float lo = v - (float)h;
                    packed[((((long long)c * 2 + 0) * 2 + half) * 64 + row) * 8 + e] = f16_bits((float)h);
                    packed[((((long long)c * 2 + 1) * 2 + half) * 64 + row) * 8 + e] = f16_bits(lo);
                }
    }
    return 0;
}

// W [96][96] row-major -> wperm[(2*kk + half)*96 + i] = W[i][4*(kk >> 1) + 2*half + (kk & 1)]   (the tail's matvec order)
extern "C" int bfsr_pack_coupling_wide_wmat(const float* w, float* wperm)
{
    if (!w || !wperm) return -1;
    for (int kk = 0; kk < CFN; ++kk)
        for (int half = 0; half < 2; ++half) {
            const int c = 4 * (kk >> 1) + 2 * half + (kk & 1);
            for (int i = 0; i < CF; ++i) wperm[(2 * kk + half) * CF + i] = w[i * CF + c];
        }
    return 0;
}

extern "C" int bfsr_coupling_wide_head(const BfsrWideHeadArgs* a, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || !a->z1 || !a->w0 || !a->pre || !a->w2 || !a->epi0 || !a->epi2 || !a->hid) return -1;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->Cz <= 0 || (a->Cz & 15) || a->Cz > 64) return -1;
    if (!(a->acc_scale0 > 0.f) || !(a->acc_scale2 > 0.f)) return -1;
    if ((reinterpret_cast<unsigned long long>(a->z1) & 15) || (a->z1_bs & 7) || (reinterpret_cast<unsigned long long>(a->pre) & 15) || (a->pre_bs & 7) ||
        (reinterpret_cast<unsigned long long>(a->hid) & 15) || (a->hid_bs & 7) || (reinterpret_cast<unsigned long long>(a->w2) & 15)) return -1;
    if ((long long)8 * 2 * a->H * a->W * 16 >= (1LL << 31)) return -1;   // 32-bit byte offsets inside one batch item
    static std::atomic<unsigned long long> lds_done{0};
    return launch_wide(&coupling_wide_head_kernel, *a, HEAD_LDS, lds_done, st);
}

extern "C" int bfsr_coupling_wide_tail(const BfsrWideTailArgs* a, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || !a->hid || !a->w || !a->bias || !a->post_scale || !a->z_in || !a->z_out) return -1;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->C != CF) return -1;
    if (a->an_bias && !a->an_escale) return -1;
    if (!(a->acc_scale > 0.f)) return -1;
    if (a->h_ft_fmt != 0 && a->h_ft_fmt != 1) return -1;
    if (a->reverse && !a->wperm) return -1;                               // the inverse always ends in W^-1 and the ActNorm inverse
    if ((reinterpret_cast<unsigned long long>(a->hid) & 15) || (a->hid_bs & 7)) return -1;
    if (a->h_ft && a->h_ft_fmt == 1 && ((reinterpret_cast<unsigned long long>(a->h_ft) & 15) || (a->h_ft_bs & 3))) return -1;
    if (a->z1h && ((reinterpret_cast<unsigned long long>(a->z1h) & 15) || (a->z1h_bs & 7))) return -1;
    if ((long long)2 * CF * a->H * a->W * 4 >= (1LL << 31)) return -1;
    static std::atomic<unsigned long long> lds_rev{0}, lds_fwd{0};
    return a->reverse ? launch_wide(&coupling_wide_tail_kernel<1>, *a, TAIL_LDS, lds_rev, st) : launch_wide(&coupling_wide_tail_kernel<0>, *a, TAIL_LDS, lds_fwd, st);
}
