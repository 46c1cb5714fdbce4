// coupling.hip -- the head of the sequential part of a conditional-affine FlowStep (FlowAffineCouplingsAblation.py:57-135,
// FlowStep.py:88-129).  A coupled step at levels with C in {12, 24} flow channels runs as TWO kernels:
//
//   bfsr_coupling_head (this file): hid = relu(AN2(W2 . relu(AN0(conv3x3(z1; W0z) + pre_aff))))        -> hid, an h2 tensor [B][8][2][H][W][8] fp16
//   bfsr_coupling_tail (conv_h2s.hip, conv3x3_h2x_kernel with the coupling epilogue): h_aff = Conv2dZeros(hid), then the step's
//       pointwise chain with h_aff taken from registers.
//
// Arithmetic (round 4): the TWO-TERM fp16 split of conv_h2s.hip / conv_bf16x3.hip -- x = hi + lo with hi = fp16(x), lo = fp16(x - hi)
// (22 significant bits), product = lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation, weights pre-multiplied by a
// power of two and the accumulators by its inverse first thing in the epilogue -- three matrix instructions per operand pair instead
// of the six of round 3's 3xBF16 head.  Values that are split (z1, the hidden activations) must stay below 2^15 in magnitude: the kernel
// raises `flag` otherwise (the host turns that into an error, bfsr_amd/ops.py).
//
// Structure.  NWV waves = NWV tile rows x 32 pixels per workgroup, M = 64 = 2 row tiles of output channels.  The 3x3 has K = 9 taps x
// ceil(Cz/8) channel octets: a k-chunk of 16 = two (tap, octet) units, lanes 0-31 read the B operand of the first unit, lanes 32-63 of
// the second, straight from the split z1 tile in LDS.  The 1x1 is CHAINED IN REGISTERS: the accumulator layout of stage 1 (lane = pixel;
// half-wave h holds channels (r&3) + 8(r>>2) + 4h) is a valid B operand of the next GEMM if W2's K axis is packed in that order.
// PERSISTENT workgroups walk their tiles in an XCD-aware order; z1 and the hoisted partial of tile t+1 are loaded into registers while
// tile t is in the matrix pipe.
//
// What changed against round 3's kernel (tools/exp/kernels/coupling_r3.hip; DESIGN.md section 5 "the head fault"):
//   * the z1 tile is DOUBLE-BUFFERED in LDS and there is ONE barrier per tile: tile t+1 is written into the other buffer, so no wave can
//     overwrite positions another wave of the workgroup is still reading, whatever their relative timing -- the write-after-read reuse
//     of a single tile across one barrier was where round 3's intermittent wrong-half-row fault entered (study: tools/exp/head_fault.py);
//   * ActNorm parameters come from LDS (staged once per workgroup), not from 64 global loads per lane and tile;
//   * pre_aff may be handed over quad-major (pre_fmt 1: [B][16][H][W][4], 8 x 16-byte loads per lane instead of 32 x 4-byte ones);
//   * hid leaves as an h2 tensor, the half-wave's octet offset travels in the VGPR offset of the stores (round 3's scalar-offset form
//     made hipcc wrap every store in a two-pass waterfall loop).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "../../include/bfsr_hip.h"
#include "launch_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TW = 32, PW = TW + 2;
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ void split2(float v, _Float16& h, _Float16& l)
{
    const float hf = bfsr::pin_f16(v);
    h = (_Float16)hf;
    l = (_Float16)(v - hf);
}

// NO = ceil(Cz/8) z1 octets, NWV = waves (= tile rows) per workgroup.  NO = 0: no 3x3 stage at all -- hid = relu(AN2(W2 . relu(AN0(pre_aff)))),
// the form the hoisted fFeatures nets use (fFeatures.0's ActNorm + ReLU on the hoisted conv result, then fFeatures.2; FlowAffineCouplingsAblation.py:127-135)
template <int NO, int NWV>
struct HeadGeo {
    static constexpr int TH = NWV, NT = NWV * 64, NPOS = (TH + 2) * PW;
    static constexpr int NU = 9 * NO, NC1 = (NU + 1) / 2;       // (tap, octet) units and 16-wide k-chunks of the 3x3
    static constexpr int ZT = NO * 2 * NPOS * 16;               // bytes of one split z1 tile: [octet][plane][pos][8] fp16
    static constexpr int W0B = NC1 * 2 * 2 * 64 * 16;           // [chunk][plane][k half][64 rows][8] fp16
    static constexpr int W2B = 4 * 2 * 2 * 64 * 16;
    static constexpr int PB = 2 * 64 * 8;                       // epi0, epi2: [64] {shift, scale}
    static constexpr int LDS = 2 * ZT + W0B + W2B + PB;
    static constexpr int ZU = NO ? (NO * NPOS + NT - 1) / NT : 1; // staged (octet, position) units per thread
};

// PF = pre_fmt (compile-time: the two load forms need very different numbers of scalar offsets)
template <int NO, int NWV, int PF>
__global__ __launch_bounds__(NWV * 64, 2) void coupling_head_kernel(BfsrCouplingHeadArgs p, int tiles_x, int tiles_xy, int ntiles)
{
    typedef HeadGeo<NO, NWV> Geo;
    constexpr int TH = Geo::TH, NT = Geo::NT, NPOS = Geo::NPOS, NU = Geo::NU, NC1 = Geo::NC1, ZT = Geo::ZT, W0B = Geo::W0B, W2B = Geo::W2B, ZU = Geo::ZU;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sW0 = smem + 2 * ZT;
    unsigned char* sW2 = sW0 + W0B;
    const float2* sP = reinterpret_cast<const float2*>(sW2 + W2B);       // [0..63] epi0, [64..127] epi2

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int G = gridDim.x;
    const int slot = (int)bfsr::xcd_order(blockIdx.x, (unsigned)G);
    if (slot >= ntiles) return;
    const int H = p.H, W = p.W, Cz = p.Cz;
    const long long HW = (long long)H * W;
    {
        // all loads of the packed weights first, then the LDS writes: one memory round trip per workgroup instead of one per 4 KiB
        constexpr int NWQ = (W0B + W2B) / 16, WQ = (NWQ + NT - 1) / NT;
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(p.w);
        uint4* dst = reinterpret_cast<uint4*>(sW0);
        uint4 wq[WQ];
#pragma unroll
        for (int k = 0; k < WQ; ++k) { const int i = tid + k * NT; wq[k] = i < NWQ ? src[i] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
        for (int k = 0; k < WQ; ++k) { const int i = tid + k * NT; if (i < NWQ) dst[i] = wq[k]; }
        float2* dp = reinterpret_cast<float2*>(sW2 + W2B);
        for (int i = tid; i < 128; i += NT) {
            const float4 q = reinterpret_cast<const float4*>(i < 64 ? p.epi0 : p.epi2)[i & 63];
            dp[i] = make_float2(q.x, q.y);
        }
    }
    if constexpr (NO == 0) __syncthreads();                               // (with a 3x3 stage the first tile's barrier covers the staging)
    unsigned bad = 0u;                                                    // range guard of the fp16 split (see the header)

    // ---- per-tile register prefetch: this thread's z1 units (8 channels of one staged position) and its 32 pre_aff values
    // TWO register sets (A, B) with static roles: the tile loop is unrolled by two, tile t computes from set A while the loads of tile
    // t+G land in set B and vice versa.  (One set that is reloaded inside the loop made hipcc copy the loaded registers at the loop
    // header behind an `s_waitcnt vmcnt(0)` -- which also drained the z1 loads just issued: every tile started with a full memory
    // round trip.)
    float zrA[ZU][8], zrB[ZU][8];
    float preA[2][16], preB[2][16];                                       // [m][r]: channel m*32 + (r&3) + 8(r>>2) + 4*lhi of this lane's pixel (= accumulator order)
    auto prefetch_z = [&](int t, float (&zr)[ZU][8]) {
        const int tile = t % tiles_xy, b = t / tiles_xy;
        const int x0 = (tile % tiles_x) * TW, y0 = (tile / tiles_x) * TH;
        // unconditional buffer loads (a branch per load made hipcc wait for every load before issuing the next): halo positions outside
        // the image get an out-of-range VGPR offset, channels >= Cz fall beyond the descriptor's Cz * H * W * 4 bytes -- both return 0
        const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.z + (long long)b * p.z_bs), 0, (unsigned)(Cz * HW * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < ZU; ++i) {
            const int u = tid + i * NT;
            const int o = u / NPOS, pos = u - o * NPOS;
            const int r = pos / PW, c = pos - r * PW;
            const int gy = y0 + r - 1, gx = x0 + c - 1;
            const bool ok = u < NO * NPOS && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const unsigned vo = ok ? (unsigned)((8LL * o * HW + (long long)gy * W + gx) * 4) : OOB;
#pragma unroll
            for (int e = 0; e < 8; ++e) zr[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, vo, (unsigned)(e * HW * 4), 0));
        }
    };
    auto prefetch_pre = [&](int t, float (&pre)[2][16]) {
        const int tile = t % tiles_xy, b = t / tiles_xy;
        const int x0 = (tile % tiles_x) * TW, y0 = (tile / tiles_x) * TH;
        const int gy = y0 + wave, gx = x0 + l31;
        const bool pok = gy < H && gx < W;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pre_aff + (long long)b * p.pre_aff_bs), 0,
                                                                            (unsigned)(64 * HW * 4), 0x00020000);
        if constexpr (PF == 1) {
            // quad-major [16][H][W][4]: quad m*8 + 2g + lhi of this lane's pixel = one 16-byte load
            const unsigned vo = pok ? (unsigned)(((long long)lhi * HW + (long long)gy * W + gx) * 16) : OOB;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (unsigned)((m * 8 + 2 * g) * HW * 16), 0));
                    pre[m][4 * g] = v.x; pre[m][4 * g + 1] = v.y; pre[m][4 * g + 2] = v.z; pre[m][4 * g + 3] = v.w;
                }
        } else {
            const unsigned vo = pok ? (unsigned)(((long long)gy * W + gx + 4LL * lhi * HW) * 4) : OOB;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    pre[m][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, (unsigned)((m * 32 + (r & 3) + 8 * (r >> 2)) * HW * 4), 0));
        }
    };
    if constexpr (NO > 0) prefetch_z(slot, zrA);
    prefetch_pre(slot, preA);

    const float s0 = p.acc_scale0, s2 = p.acc_scale2;
    auto tile_body = [&](int t, unsigned char* sZ, float (&zr)[ZU][8], float (&pre)[2][16], float (&zr_n)[ZU][8], float (&pre_n)[2][16]) {
        const int tile = t % tiles_xy, b = t / tiles_xy;
        const int x0 = (tile % tiles_x) * TW, y0 = (tile / tiles_x) * TH;
        f32x16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        half8 fb[2][2], fa[2][2][2];                        // [buffer][plane] | [buffer][m][plane]
        if constexpr (NO > 0) {
        // ---- registers -> split z1 tile in LDS
#pragma unroll
        for (int i = 0; i < ZU; ++i) {
            const int u = tid + i * NT;
            if (u < NO * NPOS) {
                const int o = u / NPOS, pos = u - o * NPOS;
                half8 h8, l8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    _Float16 h, l;
                    split2(zr[i][e], h, l);
                    h8[e] = h; l8[e] = l;
                    bad |= (unsigned)!(fabsf(zr[i][e]) < 32768.f);
                }
                *reinterpret_cast<half8*>(sZ + ((o * 2 + 0) * NPOS + pos) * 16) = h8;
                *reinterpret_cast<half8*>(sZ + ((o * 2 + 1) * NPOS + pos) * 16) = l8;
            }
        }
        __syncthreads();                                    // the ONLY barrier of a tile: tile t is complete in buffer `par^1`, and every wave has
                                                            // finished its reads of tile t-1 (the other buffer), which tile t+1 will overwrite
        if (t + G < ntiles) prefetch_z(t + G, zr_n);        // next tile's z1 loads fly under this tile's MFMAs and stores
        }
        // smallest terms first: w_lo*x_hi, w_hi*x_lo, w_hi*x_hi
#define BFSR_THREE(ACC_, A_, B_)                                                                               \
    ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[1], B_[0], ACC_, 0, 0, 0);                                 \
    ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[0], B_[1], ACC_, 0, 0, 0);                                 \
    ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[0], B_[0], ACC_, 0, 0, 0);
        // ---- 3x3: chunk j = units (2j, 2j+1); unit u = (tap u / NO, octet u % NO); lanes 0-31 take unit 2j, lanes 32-63 unit 2j+1
        if constexpr (NO > 0) {
            auto frags = [&](int j, half8 (&bf)[2], half8 (&af)[2][2]) {
                const int u0 = 2 * j, u1 = (2 * j + 1 < NU) ? 2 * j + 1 : 2 * j;  // a missing second unit re-reads the first (its weights are 0)
                const int t0 = u0 / NO, o0 = u0 % NO, t1 = u1 / NO, o1 = u1 % NO;
                const int a0 = (o0 * 2 * NPOS + (t0 / 3) * PW + (t0 % 3)) * 16, a1 = (o1 * 2 * NPOS + (t1 / 3) * PW + (t1 % 3)) * 16;
                const unsigned char* bp = sZ + (lhi ? a1 : a0) + (wave * PW + l31) * 16;
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) bf[pl] = *reinterpret_cast<const half8*>(bp + pl * NPOS * 16);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        af[m][pl] = *reinterpret_cast<const half8*>(sW0 + (((j * 2 + pl) * 2 + lhi) * 64 + m * 32 + l31) * 16);
            };
            frags(0, fb[0], fa[0]);
#pragma unroll
            for (int j = 0; j < NC1; ++j) {
                if (j + 1 < NC1) frags(j + 1, fb[(j + 1) & 1], fa[(j + 1) & 1]);
#pragma unroll
                for (int m = 0; m < 2; ++m) { BFSR_THREE(acc[m], fa[j & 1][m], fb[j & 1]) }
            }
        }
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
                    float v = ((acc[m][r] * s0 + pre[m][r]) + sh[e]) * sc[e];
                    v = v > 0.f ? v : 0.f;
                    bad |= (unsigned)!(v < 32768.f);
                    _Float16 h, l;
                    split2(v, h, l);
                    b2[m * 2 + (g >> 1)][0][(g & 1) * 4 + e] = h;
                    b2[m * 2 + (g >> 1)][1][(g & 1) * 4 + e] = l;
                }
            }
        // Keep the next tile's loads BEHIND epilogue 1: hoisted above it (or epilogue 1 sunk below them -- hipcc does that across the
        // branch) they sit between pre(t) and the z1 loads in the in-order vmcnt queue, and epilogue 1's wait for pre(t) becomes a wait for
        // the z1 loads just issued.  The empty asm statements make the B operands of the 1x1 "used" here, the fence pins the order.
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int c = 0; c < 4; ++c) { asm volatile("" :: "v"(b2[c][0])); asm volatile("" :: "v"(b2[c][1])); }
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (t + G < ntiles) prefetch_pre(t + G, pre_n);     // ... and its hoisted partial under the 1x1 and the stores
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc2[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[m][r] = 0.f;
        {
            auto load_a2 = [&](int c, half8 (&af)[2][2]) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        af[m][pl] = *reinterpret_cast<const half8*>(sW2 + (((c * 2 + pl) * 2 + lhi) * 64 + m * 32 + l31) * 16);
            };
            load_a2(0, fa[0]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c + 1 < 4) load_a2(c + 1, fa[(c + 1) & 1]);
#pragma unroll
                for (int m = 0; m < 2; ++m) { BFSR_THREE(acc2[m], fa[c & 1][m], b2[c]) }
            }
        }
#undef BFSR_THREE
        // ---- epilogue 2: weight scale, ActNorm, ReLU; v_permlane32_swap pairs the half-waves so that every lane holds two complete
        // channel octets of its pixel per row tile; split and store both planes of the h2 tensor (16 bytes each)
        {
            const int gy = y0 + wave, gx = x0 + l31;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.hid + (long long)b * p.hid_bs, 0, (unsigned)(8 * 2 * HW * 16), 0x00020000);
            const unsigned vo = (gy < H && gx < W) ? (unsigned)(((long long)lhi * 2 * HW + (long long)gy * W + gx) * 16) : OOB;     // out-of-image lanes: dropped
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
    };
    for (int t = slot; t < ntiles; t += 2 * G) {
        tile_body(t, smem, zrA, preA, zrB, preB);
        if (t + G < ntiles) tile_body(t + G, smem + ZT, zrB, preB, zrA, preA);
    }
    if (p.flag && __any((int)bad)) {
        if (lane == 0) atomicOr(p.flag, 1u);
    }
}

template <int NO, int NWV, int PF>
int launch_head(const BfsrCouplingHeadArgs& a, hipStream_t st)
{
    typedef HeadGeo<NO, NWV> Geo;
    constexpr int LDS = Geo::LDS;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static std::atomic<unsigned long long> lds_done{0};
    if (LDS > 65536 && bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&coupling_head_kernel<NO, NWV, PF>), LDS, lds_done) != 0) return -1;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + Geo::TH - 1) / Geo::TH;
    const long long ntiles = (long long)tiles_x * tiles_y * a.B;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) return -1;
    int cus = bfsr::cu_count();                             // cached per device; no silent default
    if (cus <= 0) return -1;
    const long long slots = (long long)cus * (8 / NWV);     // two waves per SIMD
    const long long grid = ntiles < slots ? ntiles : slots;
    hipLaunchKernelGGL((coupling_head_kernel<NO, NWV, PF>), dim3((unsigned)grid), dim3(NWV * 64), LDS, st, a, tiles_x, tiles_x * tiles_y, (int)ntiles);
    return (int)hipGetLastError();
}

inline unsigned short f16_bits(float v)
{
    const _Float16 h = (_Float16)v;
    unsigned short s;
    __builtin_memcpy(&s, &h, 2);
    return s;
}

}  // namespace

// ---- host-side packing -----------------------------------------------------------------------------------------------
extern "C" long long bfsr_coupling_head_packed_size(int Cz)
{
    if (Cz < 0 || Cz > 16) return -1;                         // Cz = 0: no 3x3 stage (the 1x1 only)
    const int NO = (Cz + 7) / 8, NC1 = (9 * NO + 1) / 2;
    return (long long)(NC1 + 4) * 2 * 2 * 64 * 8;             // fp16 elements
}

// w0 [64][Cz][3][3] (fAffine.0 rows restricted to z1), w2 [64][64] (fAffine.2, 1x1), each multiplied by its power-of-two scale and split
// into fp16 planes hi, lo -> the LDS image of coupling_head_kernel: [chunk][plane][k half][64 rows][8]; 3x3 chunks: k half h of chunk j =
// unit u = 2j+h = (tap u / NO, octet u % NO), element e = channel 8*octet + e (zero beyond Cz / beyond the last unit); 1x1 chunks:
// chunk c = (m, half): k half h, element e = input channel m*32 + (r&3) + 8*(r>>2) + 4*h with r = 8*half + e  (the accumulator order
// of the 3x3's output, see the kernel).
extern "C" int bfsr_pack_coupling_head(const float* w0, const float* w2, int Cz, float scale0, float scale2, unsigned short* packed)
{
    if ((!w0 && Cz > 0) || !w2 || !packed || Cz < 0 || Cz > 16 || !(scale0 > 0.f) || !(scale2 > 0.f)) return -1;
    const int NO = (Cz + 7) / 8, NU = 9 * NO, NC1 = (NU + 1) / 2;
    const long long n = bfsr_coupling_head_packed_size(Cz);
    for (long long i = 0; i < n; ++i) packed[i] = 0;
    auto put = [&](long long chunk, int half, int row, int e, float v) {
        const _Float16 h = (_Float16)v;
        const float lo = v - (float)h;
        packed[((((chunk * 2 + 0) * 2 + half) * 64 + row) * 8) + e] = f16_bits((float)h);
        packed[((((chunk * 2 + 1) * 2 + half) * 64 + row) * 8) + e] = f16_bits(lo);
    };
    for (int j = 0; j < NC1; ++j)
        for (int half = 0; half < 2; ++half) {
            const int u = 2 * j + half;
            if (u >= NU) continue;
            const int tap = u / NO, o = u % NO;
            for (int row = 0; row < 64; ++row)
                for (int e = 0; e < 8; ++e) {
                    const int ch = o * 8 + e;
                    if (ch < Cz) put(j, half, row, e, w0[((long long)row * Cz + ch) * 9 + tap] * scale0);
                }
        }
    for (int c = 0; c < 4; ++c) {
        const int m = c >> 1, hf = c & 1;
        for (int half = 0; half < 2; ++half)
            for (int row = 0; row < 64; ++row)
                for (int e = 0; e < 8; ++e) {
                    const int r = hf * 8 + e;
                    const int ch = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    put(NC1 + c, half, row, e, w2[(long long)row * 64 + ch] * scale2);
                }
    }
    return 0;
}

extern "C" int bfsr_coupling_head(const BfsrCouplingHeadArgs* a, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || (!a->z && a->Cz > 0) || !a->pre_aff || !a->w || !a->epi0 || !a->epi2 || !a->hid) return -1;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->Cz < 0 || a->Cz > 16) return -1;
    if (a->pre_fmt != 0 && a->pre_fmt != 1) return -1;
    if (!(a->acc_scale0 > 0.f) || !(a->acc_scale2 > 0.f)) return -1;
    if ((reinterpret_cast<unsigned long long>(a->hid) & 15) || (a->hid_bs & 7)) return -1;
    if (a->pre_fmt == 1 && ((reinterpret_cast<unsigned long long>(a->pre_aff) & 15) || (a->pre_aff_bs & 3))) return -1;
    if ((long long)64 * a->H * a->W * 4 >= (1LL << 31)) return -1;
    // FOUR waves per workgroup, two independent workgroups per CU (their barriers are private, so the VALU / memory phases of one
    // overlap the MFMA phases of the other on every SIMD)
    if (a->Cz == 0) return a->pre_fmt == 1 ? launch_head<0, 4, 1>(*a, st) : launch_head<0, 4, 0>(*a, st);
    if (a->pre_fmt == 1) return a->Cz <= 8 ? launch_head<1, 4, 1>(*a, st) : launch_head<2, 4, 1>(*a, st);
    return a->Cz <= 8 ? launch_head<1, 4, 0>(*a, st) : launch_head<2, 4, 0>(*a, st);
}
