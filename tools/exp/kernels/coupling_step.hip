// coupling_step.hip -- ONE kernel per conditional-affine FlowStep (FlowAffineCouplingsAblation.py:57-135 + FlowStep.py:88-129):
//
//   t1   = relu(AN0(conv3x3(z1; W0z) + pre_aff))          S1   3xBF16 on v_mfma_f32_32x32x16_bf16, z1 tile in LDS (x3)
//   hid  = relu(AN2(W2 . t1))                             S2   1x1 chained in REGISTERS (accumulator layout = next B operand)
//   h    = (conv3x3(hid; W4) + b4) * exp(3 logs4)         S3   Conv2dZeros on v_mfma_f32_16x16x32_bf16, hid tile in LDS (x3)
//   z    = pointwise chain of the step with that h        PW   (same semantics as bfsr_flow_pointwise / bfsr_coupling_tail)
//
// replacing coupling_head + coupling_tail (coupling.hip), which wrote the 64-channel `hid` to HBM and read it back (2 x 210 MB per
// level-1 step at BASELINE config 2) and were each bound by their own exposed load/store phases (190 + 160 us per step against
// ~45 + ~45 us of matrix time).  Here `hid` never leaves the CU.
//
// Tile geometry.  A workgroup (8 waves, persistent, one per CU, XCD-aware tile order) produces 6 x 30 output pixels per tile.
// The Conv2dZeros needs `hid` on the 8 x 32 halo of that tile = 256 positions = EIGHT 32-position MFMA column tiles: wave w owns
// hid row w for S1/S2 (both 32-channel row tiles: t1 stays in its registers between the two GEMMs), so the 1-pixel halo of hid
// is recomputed (256 / 180 = 1.42x of S1 + S2) instead of exchanged -- z1 is staged on the 10 x 34 halo.  For S3 the 6 output
// rows x two 16-pixel halves are 12 column tiles: waves 0-3 take two, waves 4-7 one (three per SIMD).
//
// LDS (one workgroup per CU): hid as an x3 tile [3 planes][8 octets][256 positions][8] bf16 = 96 KiB (the z1 tile aliases its
// start: it is dead once S1 is done), W2 (24 KiB) and -- when z1 has <= 8 channels -- W0 (30 KiB) resident for the lifetime of the
// workgroup, the per-channel ActNorm vectors, and h [2*(C-C/2)][180] fp32 for the pointwise chain.  W4 (41 KiB as x3) does not
// fit beside them: its MFMA A fragments are streamed from global memory (L2-resident, 16 B per lane, coalesced) through a register
// ring that runs ahead of the MFMAs, like linf_mlp.hip's weights; so is W0 at C = 24 (z1 = 12 channels = 2 octets).
//
// Overlap.  The hoisted partial `pre_aff` (256 B per position, the dominant HBM stream) and the z1 tile of tile t+1 are loaded
// into registers while tile t is in the matrix pipe; the pointwise operands of tile t are loaded under its own S3.  z_in and
// z_out must NOT alias: a tile reads the z1 halo that its neighbours' pointwise chains overwrite (the host ping-pongs).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/bfsr_hip.h"
#include "launch_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int OH = 6, OW = 30;                 // output pixels per tile
constexpr int HR = 8, HC = 32, NPH = HR * HC;  // hid halo tile: 256 positions
constexpr int ZR = 10, ZC = 34, NPZ = ZR * ZC; // z1 halo tile: 340 positions
constexpr int NPX = OH * OW;                   // 180
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ float sigmoid_scale(float raw, float eps) { return 1.f / (1.f + expf(-(raw + 2.f))) + eps; }

__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l)
{
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

#ifndef BFSR_STEP_V
#define BFSR_STEP_V 0          // timing experiments of the trace build (tools/exp/build_step_trace.sh): never set in the product
#endif
#ifdef BFSR_STEP_TRACE
// diagnostics build only (tools/exp/build_step_trace.sh): s_memtime stamps of workgroup 0's waves at the phase boundaries
__device__ unsigned long long* g_step_trace = nullptr;
#define BFSR_TRACE(K_)                                                                                   \
    if (blockIdx.x == 0 && lane == 0 && g_step_trace && iter < 8) g_step_trace[(iter * 8 + wave) * 16 + (K_)] = __builtin_amdgcn_s_memtime();
#else
#define BFSR_TRACE(K_)
#endif

template <int NO, int C>
struct Geo {
    static constexpr int CN = C / 2, CC = C - CN, CO2 = 2 * CC, MT = (CO2 + 15) / 16, MW = MT * 16;
    static constexpr int NU = 9 * NO, NC1 = (NU + 1) / 2;
    static constexpr bool W0_RES = NO == 1 && !(BFSR_STEP_V & 64);
    static constexpr int HID_B = 3 * 8 * NPH * 16 + 64;        // + 2 positions of slack behind the last slab (waste lanes of S3)
    static constexpr int Z_B = NO * 3 * NPZ * 16;
    static constexpr int CHUNK_B = 3 * 2 * 64 * 16;            // one 16-wide k chunk of a 64-row GEMM: [plane][k half][64][8]
    static constexpr int W2_B = 4 * CHUNK_B, W0_B = NC1 * CHUNK_B;
    static constexpr int EPI_B = 2 * 64 * 8;                   // {shift, scale} of the two ActNorms
    static constexpr int SH_B = CO2 * NPX * 4;
    static constexpr int OFF_W2 = HID_B, OFF_W0 = OFF_W2 + W2_B, OFF_EPI = OFF_W0 + (W0_RES ? W0_B : 0), OFF_SH = OFF_EPI + EPI_B;
    static constexpr int LDS = OFF_SH + SH_B + (((BFSR_STEP_V & 64) && NO == 1) ? Z_B : 0);
    static constexpr int W4_CHUNK = 3 * 4 * MW * 8;            // bf16 elements of one (tap, 32-channel half) chunk
    static_assert(Z_B <= HID_B, "the z1 tile aliases the hid tile");
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <int NO, int C>
__global__ __launch_bounds__(512, 2) void coupling_step_kernel(BfsrCouplingStepArgs p, int tiles_x, int tiles_xy, int ntiles)
{
    using G = Geo<NO, C>;
    constexpr int CN = G::CN, CC = G::CC, CO2 = G::CO2, MT = G::MT, MW = G::MW, NU = G::NU, NC1 = G::NC1;
    constexpr int ZU = (NO * NPZ + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sHid = smem;
    unsigned char* sZ = ((BFSR_STEP_V & 64) && NO == 1) ? smem + G::OFF_SH + G::SH_B : smem;    // experiment 64: z1 tile NOT aliased
    unsigned char* sW2 = smem + G::OFF_W2;
    unsigned char* sW0 = smem + G::OFF_W0;
    float2* sE0 = reinterpret_cast<float2*>(smem + G::OFF_EPI);
    float2* sE2 = sE0 + 64;
    float* sH = reinterpret_cast<float*>(smem + G::OFF_SH);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5, l15 = lane & 15, lq = lane >> 4;
    const int GD = gridDim.x;
    const int slot = (int)bfsr::xcd_order(blockIdx.x, (unsigned)GD);
    if (slot >= ntiles) return;
    const int H = p.H, W = p.W;
    const long long HW = (long long)H * W;

    // ---- resident weights and ActNorm vectors
    {
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(p.w_head);
        if constexpr (G::W0_RES) {
            uint4* dst = reinterpret_cast<uint4*>(sW0);
            for (int i = tid; i < G::W0_B / 16; i += 512) dst[i] = src[i];
        }
        uint4* dst2 = reinterpret_cast<uint4*>(sW2);
        for (int i = tid; i < G::W2_B / 16; i += 512) dst2[i] = src[G::W0_B / 16 + i];
        if (tid < 64) {
            sE0[tid] = make_float2(p.epi0[4 * tid], p.epi0[4 * tid + 1]);
            sE2[tid] = make_float2(p.epi2[4 * tid], p.epi2[4 * tid + 1]);
        }
    }

    auto tile_origin = [&](int t, int& b, int& x0, int& y0) {
        const int tile = t % tiles_xy;
        b = t / tiles_xy;
        x0 = (tile % tiles_x) * OW;
        y0 = (tile / tiles_x) * OH;
    };

    // ---- register prefetch of the next tile: this thread's z1 units (8 channels of one staged position) and its 32 pre_aff values
    float zr[ZU][8], pre[2][16];
    auto prefetch_z = [&](int t) {
        int b, x0, y0;
        tile_origin(t, b, x0, y0);
        const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.z_in + (long long)b * p.z_in_bs), 0, (unsigned)(CN * HW * 4), 0x00020000);
        int tz = tid;
        asm volatile("" : "+v"(tz));                            // keep the per-thread tile coordinates out of the loop-invariant set
#pragma unroll
        for (int i = 0; i < ZU; ++i) {
            const int u = tz + i * 512;
            const int o = u / NPZ, pos = u - o * NPZ;
            const int r = pos / ZC, c = pos - r * ZC;
            const int gy = y0 + r - 2, gx = x0 + c - 2;
            const bool ok = u < NO * NPZ && gy >= 0 && gy < H && gx >= 0 && gx < W;
            // channels beyond CN fall outside the descriptor's range and read 0
            const unsigned vo = ok ? (unsigned)(((long long)o * 8 * HW + (long long)gy * W + gx) * 4) : OOB;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                zr[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, vo, (unsigned)(e * HW * 4), 0));
        }
    };
    auto prefetch_pre = [&](int t) {
        int b, x0, y0;
        tile_origin(t, b, x0, y0);
        const int gy = y0 - 1 + wave, gx = x0 - 1 + l31;
        const bool pok = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pre_aff + (long long)b * p.pre_aff_bs), 0,
                                                                            (unsigned)(64 * HW * 4), 0x00020000);
        const unsigned vo = pok ? (unsigned)(((long long)gy * W + gx + 4LL * lhi * HW) * 4) : OOB;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                pre[m][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, (unsigned)((m * 32 + (r & 3) + 8 * (r >> 2)) * HW * 4), 0));
    };
    prefetch_z(slot);
    prefetch_pre(slot);

    // this lane's A fragments of the two 64-row GEMMs: [chunk][plane][k half][64 rows][8]
    const unsigned a64_off = (unsigned)((lhi * 64 + l31) * 16);
    // weights that are not LDS-resident stream from global memory (L2) by buffer loads: ONE offset VGPR per lane, the fragment's
    // position as a scalar offset (flat 64-bit addresses per fragment cost 2 VGPRs each and spilled)
    const __amdgpu_buffer_rsrc_t rs_w0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w_head), 0, (unsigned)(G::W0_B + G::W2_B), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w4 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w_tail), 0, (unsigned)(18 * G::W4_CHUNK * 2), 0x00020000);
    const unsigned a4_off = (unsigned)((lq * MW + l15) * 16);

#define BFSR_SIX32(ACC_, A_, B_)                                                                               \
    ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[2], B_[0], ACC_, 0, 0, 0);                                \
    ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[0], B_[2], ACC_, 0, 0, 0);                                \
    ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[1], B_[1], ACC_, 0, 0, 0);                                \
    ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[1], B_[0], ACC_, 0, 0, 0);                                \
    ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[0], B_[1], ACC_, 0, 0, 0);                                \
    ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[0], B_[0], ACC_, 0, 0, 0);

    int iter = -1;
    for (int t = slot; t < ntiles; t += GD) {
        int b, x0, y0;
        tile_origin(t, b, x0, y0);
        ++iter;
        BFSR_TRACE(0)
        // ---- z1 registers -> x3 tile in LDS.  (Everybody is past the previous tile's S3: the region is free.)
#pragma unroll
        for (int i = 0; i < ZU; ++i) {
            const int u = tid + i * 512;
            if (u < NO * NPZ) {
                const int o = u / NPZ, pos = u - o * NPZ;
                bf16x8 h8, m8, l8;
#pragma unroll
                for (int e = 0; e < 8; ++e) { __bf16 h, m, l; split3(zr[i][e], h, m, l); h8[e] = h; m8[e] = m; l8[e] = l; }
                *reinterpret_cast<bf16x8*>(sZ + ((o * 3 + 0) * NPZ + pos) * 16) = h8;
                *reinterpret_cast<bf16x8*>(sZ + ((o * 3 + 1) * NPZ + pos) * 16) = m8;
                *reinterpret_cast<bf16x8*>(sZ + ((o * 3 + 2) * NPZ + pos) * 16) = l8;
            }
        }
        BFSR_TRACE(1)
        __syncthreads();                                        // barrier 1: z1 tile (and, first time, the weights) visible
        BFSR_TRACE(2)
#ifdef BFSR_STEP_TRACE
        const bool dump = (BFSR_STEP_V & 2048) && g_step_trace && t == 0;
        unsigned char* dbase = reinterpret_cast<unsigned char*>(g_step_trace) + 4096;
        if (dump) for (int i = tid; i < G::Z_B / 16; i += 512) reinterpret_cast<uint4*>(dbase)[i] = reinterpret_cast<const uint4*>(sZ)[i];
#endif
        if (t + GD < ntiles) prefetch_z(t + GD);

        // ================= S1: 3x3 on z1.  chunk j = units (2j, 2j+1); unit u = (tap u / NO, octet u % NO); k half = lhi
        f32x16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        // MFMA operand discipline of this kernel (S1, S2, S3): a register that an MFMA reads as SrcA / SrcB stays ALLOCATED until the
        // wave has issued a whole further chunk of MFMAs (KEEP = an empty asm that uses it, pinned behind a sched_barrier), and only
        // then is it reloaded.  hipcc recycles a dead fragment register at once -- as the destination of the next ds_read or as a
        // VALU temporary two instructions behind the MFMA -- and with two waves sharing the SIMD's matrix pipe a queued
        // v_mfma_f32_32x32x16_bf16 was observed to pick up the NEW contents for columns 16-31 (tools/exp/step_hid_dump.py: z1, S1,
        // pre_aff and t1 exact, the S2 accumulators wrong in 1 of 3 runs; exact again as soon as stores separated E1 from S2).
#define BFSR_KEEP(X_) asm volatile("" :: "v"(X_))
        bf16x8 fb[3][3], fa[3][2][3];                           // three (B, A) fragment sets of the 64-row GEMMs
        {
            // chunk j: B fragments of units (2j, 2j+1) from the z1 tile; A from LDS (resident W0) or streamed from global memory
            auto b_frags = [&](int j, bf16x8 (&bf)[3]) {
                const int u0 = 2 * j, u1 = (2 * j + 1 < NU) ? 2 * j + 1 : 2 * j;   // a missing second unit re-reads the first (weights 0)
                const int t0 = u0 / NO, o0 = u0 % NO, t1 = u1 / NO, o1 = u1 % NO;
                const int a0 = (o0 * 3 * NPZ + (t0 / 3) * ZC + (t0 % 3)) * 16, a1 = (o1 * 3 * NPZ + (t1 / 3) * ZC + (t1 % 3)) * 16;
                const unsigned char* bp = sZ + (lhi ? a1 : a0) + (wave * ZC + l31) * 16;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bf[pl] = *reinterpret_cast<const bf16x8*>(bp + pl * NPZ * 16);
            };
            auto a_frags = [&](int j, bf16x8 (&af)[2][3]) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        if constexpr (G::W0_RES)
                            af[m][pl] = *reinterpret_cast<const bf16x8*>(sW0 + j * G::CHUNK_B + pl * 2048 + m * 512 + a64_off);
                        else
                            af[m][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w0, a64_off, (unsigned)(j * G::CHUNK_B + pl * 2048 + m * 512), 0));
                    }
            };
            auto keep_set = [&](int sl) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) { BFSR_KEEP(fb[sl][pl]); BFSR_KEEP(fa[sl][0][pl]); BFSR_KEEP(fa[sl][1][pl]); }
            };
            b_frags(0, fb[0]); a_frags(0, fa[0]);
            if (NC1 > 1) { b_frags(1, fb[1]); a_frags(1, fa[1]); }
#pragma unroll
            for (int j = 0; j < NC1; ++j) {
#pragma unroll
                for (int m = 0; m < 2; ++m) { BFSR_SIX32(acc[m], fa[j % 3][m], fb[j % 3]) }
                __builtin_amdgcn_sched_barrier(0);
                if (j >= 1) keep_set((j - 1) % 3);              // chunk j-1's operands may be recycled from here on ...
                if (j + 2 < NC1) { b_frags(j + 2, fb[(j + 2) % 3]); a_frags(j + 2, fa[(j + 2) % 3]); }   // ... by chunk j+2's
            }
        }
        BFSR_TRACE(3)
#ifdef BFSR_STEP_TRACE
        if (dump) {
            float* da = reinterpret_cast<float*>(dbase + 32768);
            float* dp = reinterpret_cast<float*>(dbase + 32768 + 65536);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) { da[((wave * 2 + m) * 16 + r) * 64 + lane] = acc[m][r]; dp[((wave * 2 + m) * 16 + r) * 64 + lane] = pre[m][r]; }
        }
#endif
        // ---- E1 in registers: + pre_aff, ActNorm, ReLU; the result IS the B operand of the 1x1 (K order = accumulator order)
        bf16x8 b2[4][3];                                        // chunk c = (m, half): registers 8*half .. 8*half+7 of row tile m
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = hf * 8 + e;
                    const int ch = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    const float2 q = sE0[ch];
                    float v = ((acc[m][r] + pre[m][r]) + q.x) * q.y;
                    v = v > 0.f ? v : 0.f;
                    __bf16 h, mm, l;
                    split3(v, h, mm, l);
                    b2[m * 2 + hf][0][e] = h; b2[m * 2 + hf][1][e] = mm; b2[m * 2 + hf][2][e] = l;
                }
        __builtin_amdgcn_sched_barrier(0);                      // E1 has read both accumulators: S1's last MFMAs are complete
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { BFSR_KEEP(fb[(NC1 - 1) % 3][pl]); BFSR_KEEP(fa[(NC1 - 1) % 3][0][pl]); BFSR_KEEP(fa[(NC1 - 1) % 3][1][pl]); }
        BFSR_TRACE(4)
#ifdef BFSR_STEP_TRACE
        if (dump) {                                             // t1 as the S2 B operand holds it: sum of the three planes
            float* db = reinterpret_cast<float*>(dbase + 32768 + 3 * 65536 + 98304);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    db[((wave * 4 + c) * 8 + e) * 64 + lane] = ((float)b2[c][0][e] + (float)b2[c][1][e]) + (float)b2[c][2][e];
        }
#endif
        if (!(BFSR_STEP_V & 2) && t + GD < ntiles) prefetch_pre(t + GD);   // the next tile's hoisted partial flies under S2 ... S3
        // ================= S2: 1x1, K = 64 = 4 chunks in accumulator order
        f32x16 acc2[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[m][r] = 0.f;
        {
            // A fragments (W2, LDS) two chunks ahead in the three register sets of S1 (its fragments are dead), same discipline
            auto load_a2 = [&](int c, bf16x8 (&dst)[2][3]) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        dst[m][pl] = *reinterpret_cast<const bf16x8*>(sW2 + c * G::CHUNK_B + pl * 2048 + m * 512 + a64_off);
            };
            load_a2(0, fa[0]);
            load_a2(1, fa[1]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int m = 0; m < 2; ++m) { BFSR_SIX32(acc2[m], fa[c % 3][m], b2[c]) }
                __builtin_amdgcn_sched_barrier(0);
                if (c >= 1) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) { BFSR_KEEP(fa[(c - 1) % 3][0][pl]); BFSR_KEEP(fa[(c - 1) % 3][1][pl]); BFSR_KEEP(b2[c - 1][pl]); }
                }
                if (c + 2 < 4) load_a2(c + 2, fa[(c + 2) % 3]);
            }
        }
        if ((BFSR_STEP_V & 2) && t + GD < ntiles) prefetch_pre(t + GD);
#ifdef BFSR_STEP_TRACE
        if (dump) {
            float* d2 = reinterpret_cast<float*>(dbase + 32768 + 2 * 65536);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) d2[((wave * 2 + m) * 16 + r) * 64 + lane] = acc2[m][r];
        }
#endif
        BFSR_TRACE(5)
        __syncthreads();                                        // barrier 2: nobody reads the z1 tile any more -> hid may be written
        BFSR_TRACE(6)
        // ---- E2: ActNorm, ReLU, zero outside the image (Conv2dZeros pads hid with zeros), channel-octet transposition, x3 -> LDS
        {
            const int gy = y0 - 1 + wave, gx = x0 - 1 + l31;
            const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
            asm volatile("s_nop 11" ::: "memory");              // MFMA result -> VALU read inside the asm below
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float v[2][8];
#pragma unroll
                for (int qd = 0; qd < 2; ++qd)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float lo = acc2[m][8 * qd + i], hi = acc2[m][8 * qd + 4 + i];
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
                        v[qd][i] = lo; v[qd][4 + i] = hi;
                    }
#pragma unroll
                for (int qd = 0; qd < 2; ++qd) {
                    const int oct = m * 4 + qd * 2 + lhi;       // channel octet of hid held by this lane
                    bf16x8 h8, m8, l8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float2 q = sE2[oct * 8 + e];
                        float u = (v[qd][e] + q.x) * q.y;
                        u = (u > 0.f && inside) ? u : 0.f;
                        __bf16 h, mm, l;
                        split3(u, h, mm, l);
                        h8[e] = h; m8[e] = mm; l8[e] = l;
                    }
                    unsigned char* dst = sHid + (oct * NPH + wave * HC + l31) * 16;
                    *reinterpret_cast<bf16x8*>(dst) = h8;
                    *reinterpret_cast<bf16x8*>(dst + 8 * NPH * 16) = m8;
                    *reinterpret_cast<bf16x8*>(dst + 16 * NPH * 16) = l8;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);                      // E2 has read both accumulators: S2's last MFMAs are complete
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { BFSR_KEEP(fa[0][0][pl]); BFSR_KEEP(fa[0][1][pl]); BFSR_KEEP(b2[3][pl]); }
        BFSR_TRACE(7)
        __syncthreads();                                        // barrier 3: hid tile complete
        BFSR_TRACE(8)
#ifdef BFSR_STEP_TRACE
        if (dump) {                                                 // experiment 2048: dump tile 0's hid tile (raw x3 planes)
            for (int i = tid; i < 3 * 8 * NPH; i += 512) reinterpret_cast<uint4*>(dbase + 32768 + 3 * 65536)[i] = reinterpret_cast<const uint4*>(sHid)[i];
        }
#endif

        // ---- operands of phase A of this tile's pointwise chain: work item i = tid + 512 k = (channel i / 180, pixel i % 180),
        // loaded here, consumed after S3
        constexpr int NIA = (C * NPX + 511) / 512;
        float pz[NIA], psh[NIA], psr[NIA];
        // `tv` = tid, re-defined opaquely per tile: the item index arithmetic below (divisions by 180 / 30 per item) is loop-invariant
        // per thread, and hipcc otherwise hoists ~80 values out of the tile loop and SPILLS them (scratch reloads with vmcnt(0)
        // waits in the middle of S3, measured in the ISA)
        int tv = tid;
        asm volatile("" : "+v"(tv));
        {
            const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.z_in + (long long)b * p.z_in_bs), 0, (unsigned)(C * HW * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.h_ft ? p.h_ft + (long long)b * p.h_ft_bs : p.z_in), 0,
                                                                                p.h_ft ? (unsigned)(2 * C * HW * 4) : 0u, 0x00020000);
#pragma unroll
            for (int k = 0; k < NIA; ++k) {
                const int i = tv + 512 * k;
                const int c = i / NPX, q = i - c * NPX;
                const int qy = y0 + q / OW, qx = x0 + q % OW;
                const bool on = i < C * NPX && qy < H && qx < W;
                const long long pixq = (long long)qy * W + qx;
                pz[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, on ? (unsigned)((c * HW + pixq) * 4) : OOB, 0, 0));
                const unsigned vf = on ? (unsigned)((2LL * c * HW + pixq) * 4) : OOB;
                psh[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, vf, 0, 0));
                psr[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, vf, (unsigned)(HW * 4), 0));
            }
        }

        // ================= S3: Conv2dZeros 64 -> CO2 on 16x16x32 tiles.  chunk = (tap, 32-channel half): k group lq = octet half*4+lq
        // column tile n = (output row, 16-pixel half): wave w takes n = w and, for w < 4, n = w + 8.  Two accumulator chains are
        // always in flight: the two column tiles of waves 0-3; for waves 4-7 the six cross products of their single tile alternate
        // between two partial accumulators (six dependent 16x16x32 MFMAs on one accumulator were the critical path: 11.4k cycles
        // for 216 MFMAs).  The B fragments of chunk ck+1 are read from LDS before the MFMAs of chunk ck are issued.
        {
            const bool two = wave < 4;
            f32x4 acc4[2][MT];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc4[n][m][r] = 0.f;
            // per-lane LDS bases of the B fragments, from the opaque `tv` (not hoisted out of the tile loop): column tile n, k group
            // lq; planes 0-1 and plane 2 get separate bases so that every fragment is base + a 16-bit immediate offset
            const int v15 = tv & 15, vq = (tv >> 4) & 3;
            const unsigned char* hb0[2];
            const unsigned char* hb2[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int nt = two ? wave + 8 * n : wave;
                hb0[n] = sHid + vq * NPH * 16 + (((nt >> 1) * HC) + (nt & 1) * 16 + v15) * 16;
                hb2[n] = hb0[n] + 2 * 8 * NPH * 16;
            }
            // A ring (W4 from global memory) and B sets (hid from LDS) under the operand discipline of S1: chunk ck's operands are
            // released behind the MFMAs of chunk ck+1 and reloaded right there, RA-1 / NB-1 chunks ahead of their use
            constexpr int RA = MT == 1 ? 4 : 3, NB = MT == 1 ? 3 : 2;
            bf16x8 ar[RA][3][MT];
            auto load_a4 = [&](int ck, bf16x8 (&dst)[3][MT]) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        dst[pl][m] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w4, a4_off, (unsigned)((ck * G::W4_CHUNK + pl * 4 * MW * 8 + m * 16 * 8) * 2), 0));
            };
            bf16x8 bfr[NB][2][3];                               // [set][column tile][plane]
            auto load_b = [&](int ck, bf16x8 (&dst)[2][3]) {
                const int tap = ck >> 1, hf = ck & 1;
                const int toff = ((tap / 3) * HC + (tap % 3)) * 16 + hf * 4 * NPH * 16;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (n == 0 || two) {
                        dst[n][0] = *reinterpret_cast<const bf16x8*>(hb0[n] + toff);
                        dst[n][1] = *reinterpret_cast<const bf16x8*>(hb0[n] + 8 * NPH * 16 + toff);
                        dst[n][2] = *reinterpret_cast<const bf16x8*>(hb2[n] + toff);
                    }
                }
            };
            auto keep3 = [&](int ck) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) BFSR_KEEP(ar[ck % RA][pl][m]);
                    BFSR_KEEP(bfr[ck % NB][0][pl]);
                    if (two) BFSR_KEEP(bfr[ck % NB][1][pl]);
                }
            };
#pragma unroll
            for (int i = 0; i < RA - 1; ++i) load_a4(i, ar[i]);
#pragma unroll
            for (int i = 0; i < NB - 1; ++i) load_b(i, bfr[i]);
#pragma unroll
            for (int ck = 0; ck < 18; ++ck) {
#define BFSR_T(N_, BN_, PA_, PB_) acc4[N_][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[ck % RA][PA_][m], bfr[ck % NB][BN_][PB_], acc4[N_][m], 0, 0, 0);
                if (two) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        BFSR_T(0, 0, 2, 0) BFSR_T(1, 1, 2, 0) BFSR_T(0, 0, 0, 2) BFSR_T(1, 1, 0, 2) BFSR_T(0, 0, 1, 1) BFSR_T(1, 1, 1, 1)
                        BFSR_T(0, 0, 1, 0) BFSR_T(1, 1, 1, 0) BFSR_T(0, 0, 0, 1) BFSR_T(1, 1, 0, 1) BFSR_T(0, 0, 0, 0) BFSR_T(1, 1, 0, 0)
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        BFSR_T(0, 0, 2, 0) BFSR_T(1, 0, 0, 2) BFSR_T(0, 0, 1, 1) BFSR_T(1, 0, 1, 0) BFSR_T(0, 0, 0, 1) BFSR_T(1, 0, 0, 0)
                    }
                }
#undef BFSR_T
                __builtin_amdgcn_sched_barrier(0);
                if (ck >= 1) keep3(ck - 1);
                if (ck + RA - 1 < 18) load_a4(ck + RA - 1, ar[(ck + RA - 1) % RA]);
                if (ck + NB - 1 < 18) load_b(ck + NB - 1, bfr[(ck + NB - 1) % NB]);
            }
            BFSR_TRACE(9)
            // ---- E3: accumulator layout of 16x16: lane (l15, lq) holds rows 4*lq + i of column l15 -> h tile in LDS
            if (!two) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc4[0][m][i] += acc4[1][m][i];
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                if (n == 0 || two) {
                    const int nt = wave + 8 * n;
                    const int col = (nt & 1) * 16 + v15;
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int co = m * 16 + vq * 4 + i;
                            if (co < CO2 && col < OW) sH[co * NPX + (nt >> 1) * OW + col] = (acc4[n][m][i] + p.bias[co]) * p.post_scale[co];
                        }
                }
            }
            __builtin_amdgcn_sched_barrier(0);                  // E3 has read the accumulators: the last MFMAs are complete
            keep3(17);
        }
        BFSR_TRACE(10)
        __syncthreads();                                        // barrier 4: h complete; every wave is done reading hid
        BFSR_TRACE(11)

        // ---- pointwise chain of the step on ALL threads, in two phases through LDS (one thread per pixel kept waves 0-2 busy for
        // 10.8k cycles per tile while five waves waited).  Same arithmetic per element as flow_pointwise_kernel.
        // Phase A, item (channel c, pixel q): reverse: z2 = z2/scale - shift, then z = z/scaleFt - shiftFt;
        //                                     forward: z2 = (z2 + shift)*scale, then the NEXT step's ActNorm.   -> sX[c][q]
        float* sX = reinterpret_cast<float*>(smem);             // aliases the hid tile: dead behind barrier 4
        {
            const float eps = p.eps;
            const bool hf = p.h_ft != nullptr;
#pragma unroll
            for (int k = 0; k < NIA; ++k) {
                const int i = tv + 512 * k;
                if (i < C * NPX) {
                    const int c = i / NPX, q = i - c * NPX;
                    float v = pz[k];
                    if (c >= CN) {
                        const float sh = sH[(2 * (c - CN)) * NPX + q], sr = sH[(2 * (c - CN) + 1) * NPX + q];
                        v = p.reverse ? v / sigmoid_scale(sr, eps) - sh : (v + sh) * sigmoid_scale(sr, eps);
                    }
                    if (p.reverse) {
                        if (hf) v = v / sigmoid_scale(psr[k], eps) - psh[k];
                    } else if (p.an_bias) {
                        v = (v + p.an_bias[c]) * p.an_escale[c];
                    }
                    sX[i] = v;
                }
            }
        }
        // Phase B, item (pixel q, group g of 6 output channels), groups padded to 192 items so that g is wave-uniform (W through
        // scalar loads): y = W x; reverse: ActNorm inverse; forward: the next step's feature-conditional affine.
        constexpr int NGRP = C / 6, NIB = (NGRP * 192 + 511) / 512;
        float bsh[NIB][6], bsr[NIB][6];
        const bool fwd_ft = !p.reverse && p.h_ft != nullptr && p.wmat != nullptr;
#pragma unroll
        for (int k = 0; k < NIB; ++k) {
            const int g = (wave + 8 * k) / 3, q = tv + 512 * k - g * 192;
            const int qy = y0 + q / OW, qx = x0 + q % OW;
            const bool on = g < NGRP && q < NPX && qy < H && qx < W;
            if (fwd_ft || (!p.reverse && p.h_ft != nullptr)) {
                const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.h_ft + (long long)b * p.h_ft_bs), 0, (unsigned)(2 * C * HW * 4), 0x00020000);
                const unsigned vo = on ? (unsigned)(((long long)qy * W + qx) * 4) : OOB;
#pragma unroll
                for (int e = 0; e < 6; ++e) {
                    bsh[k][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, vo, (unsigned)((2 * (6 * g + e)) * HW * 4), 0));
                    bsr[k][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, vo, (unsigned)((2 * (6 * g + e) + 1) * HW * 4), 0));
                }
            }
        }
        BFSR_TRACE(12)
        __syncthreads();                                        // barrier 5: sX complete
        BFSR_TRACE(13)
#pragma unroll
        for (int k = 0; k < NIB; ++k) {
            const int g = (wave + 8 * k) / 3, q = tv + 512 * k - g * 192;      // g is wave-uniform: 192 items = 3 waves per group
            const int qy = y0 + q / OW, qx = x0 + q % OW;
            if (g < NGRP && q < NPX && qy < H && qx < W) {
                const float eps = p.eps;
                float* zo = p.z_out + (long long)b * p.z_out_bs + (long long)qy * W + qx;
                float y[6];
                if (p.wmat) {
                    float xv[C];
#pragma unroll
                    for (int j = 0; j < C; ++j) xv[j] = sX[j * NPX + q];
                    const float* __restrict__ w = p.wmat + (6 * g) * C;
#pragma unroll
                    for (int e = 0; e < 6; ++e) {
                        float a = 0.f;
#pragma unroll
                        for (int j = 0; j < C; ++j) a = fmaf(w[e * C + j], xv[j], a);
                        y[e] = a;
                        __builtin_amdgcn_sched_barrier(0);      // one row of W in SGPRs at a time (72 scalars at once spilled)
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 6; ++e) y[e] = sX[(6 * g + e) * NPX + q];
                }
#pragma unroll
                for (int e = 0; e < 6; ++e) {
                    const int ci = 6 * g + e;
                    float v = y[e];
                    if (p.reverse) {
                        if (p.wmat && p.an_bias) v = v * p.an_escale[ci] - p.an_bias[ci];
                        else if (!p.wmat && p.an_bias) v = v * p.an_escale[ci] - p.an_bias[ci];
                    } else if (p.h_ft) {
                        v = (v + bsh[k][e]) * sigmoid_scale(bsr[k][e], eps);
                    }
                    zo[(long long)ci * HW] = v;
                }
            }
        }
        BFSR_TRACE(14)
        __syncthreads();                                        // barrier 6: sX (= the hid / z1 region) is free for the next tile's staging
    }
#undef BFSR_SIX32
}

template <int NO, int C>
int launch_step(const BfsrCouplingStepArgs& a, hipStream_t st)
{
    using G = Geo<NO, C>;
    static std::atomic<unsigned long long> lds_done{0};
    if (bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&coupling_step_kernel<NO, C>), G::LDS, lds_done) != 0) return -1;
    const int tiles_x = (a.W + OW - 1) / OW, tiles_y = (a.H + OH - 1) / OH;
    const long long ntiles = (long long)tiles_x * tiles_y * a.B;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) return -1;
    const int cus = bfsr::cu_count();
    if (cus <= 0) return -1;
    const long long grid = ntiles < cus ? ntiles : cus;         // one persistent workgroup per CU
    hipLaunchKernelGGL((coupling_step_kernel<NO, C>), dim3((unsigned)grid), dim3(512), G::LDS, st, a, tiles_x, tiles_x * tiles_y, (int)ntiles);
    return (int)hipGetLastError();
}

}  // namespace

#ifdef BFSR_STEP_TRACE
extern "C" int bfsr_debug_step_trace(void* buf)
{
    unsigned long long* q = reinterpret_cast<unsigned long long*>(buf);
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_step_trace), &q, sizeof(q));
}
#endif

extern "C" long long bfsr_coupling_step_tail_packed_size(int Cout)
{
    if (Cout <= 0 || Cout > 32) return -1;
    return 18LL * 3 * 4 * ((Cout + 15) / 16 * 16) * 8;            // bf16 elements
}

// w [Cout][64][3][3] (fAffine.4 = Conv2dZeros weight) -> exact 3-term bf16 split in the fragment order of coupling_step_kernel's S3:
// [chunk = tap*2 + half][plane][k group lq][MW rows][8]; element j of k group lq = input channel half*32 + lq*8 + j; rows padded to MW
extern "C" int bfsr_pack_coupling_step_tail(const float* w, int Cout, unsigned short* packed)
{
    if (!w || !packed || Cout <= 0 || Cout > 32) return -1;
    const int MW = (Cout + 15) / 16 * 16;
    const long long n = bfsr_coupling_step_tail_packed_size(Cout);
    for (long long i = 0; i < n; ++i) packed[i] = 0;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < 64; ++ci)
            for (int t = 0; t < 9; ++t) {
                float r = w[((long long)co * 64 + ci) * 9 + t];
                const int chunk = t * 2 + ci / 32, lq = (ci % 32) / 8, j = ci % 8;
                for (int pl = 0; pl < 3; ++pl) {
                    const __bf16 h = (__bf16)r;
                    unsigned short s;
                    __builtin_memcpy(&s, &h, 2);
                    packed[((((long long)chunk * 3 + pl) * 4 + lq) * MW + co) * 8 + j] = s;
                    r -= (float)h;
                }
            }
    return 0;
}

extern "C" int bfsr_coupling_step(const BfsrCouplingStepArgs* a, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || !a->z_in || !a->z_out || !a->pre_aff || !a->w_head || !a->w_tail || !a->epi0 || !a->epi2 || !a->bias || !a->post_scale) return -1;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0) return -1;
    if (a->an_bias && !a->an_escale) return -1;
    if ((long long)64 * a->H * a->W * 4 >= (1LL << 31)) return -1;
    {   // a tile reads the z1 halo its neighbours rewrite: in-place operation is a race, not an option
        const long long span_i = ((long long)(a->B - 1) * a->z_in_bs + (long long)a->C * a->H * a->W) * 4;
        const long long span_o = ((long long)(a->B - 1) * a->z_out_bs + (long long)a->C * a->H * a->W) * 4;
        const char* i0 = reinterpret_cast<const char*>(a->z_in);
        const char* o0 = reinterpret_cast<const char*>(a->z_out);
        if (i0 < o0 + span_o && o0 < i0 + span_i) return -1;
    }
    switch (a->C) {
        case 12: return launch_step<1, 12>(*a, st);
        case 24: return launch_step<2, 24>(*a, st);
        default: return -1;
    }
}
