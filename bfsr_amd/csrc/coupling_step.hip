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

template <int NO, int C>
struct Geo {
    static constexpr int CN = C / 2, CC = C - CN, CO2 = 2 * CC, MT = (CO2 + 15) / 16, MW = MT * 16;
    static constexpr int NU = 9 * NO, NC1 = (NU + 1) / 2;
    static constexpr bool W0_RES = NO == 1;
    static constexpr int HID_B = 3 * 8 * NPH * 16 + 64;        // + 2 positions of slack behind the last slab (waste lanes of S3)
    static constexpr int Z_B = NO * 3 * NPZ * 16;
    static constexpr int CHUNK_B = 3 * 2 * 64 * 16;            // one 16-wide k chunk of a 64-row GEMM: [plane][k half][64][8]
    static constexpr int W2_B = 4 * CHUNK_B, W0_B = NC1 * CHUNK_B;
    static constexpr int EPI_B = 2 * 64 * 8;                   // {shift, scale} of the two ActNorms
    static constexpr int SH_B = CO2 * NPX * 4;
    static constexpr int OFF_W2 = HID_B, OFF_W0 = OFF_W2 + W2_B, OFF_EPI = OFF_W0 + (W0_RES ? W0_B : 0), OFF_SH = OFF_EPI + EPI_B;
    static constexpr int LDS = OFF_SH + SH_B;
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
    unsigned char* sZ = smem;
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
        const float* __restrict__ zb = p.z_in + (long long)b * p.z_in_bs;
#pragma unroll
        for (int i = 0; i < ZU; ++i) {
            const int u = tid + i * 512;
            const int o = u / NPZ, pos = u - o * NPZ;
            const int r = pos / ZC, c = pos - r * ZC;
            const int gy = y0 + r - 2, gx = x0 + c - 2;
            const bool ok = u < NO * NPZ && gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = o * 8 + e;
                zr[i][e] = (ok && ch < CN) ? zb[(long long)ch * HW + (long long)gy * W + gx] : 0.f;
            }
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

    for (int t = slot; t < ntiles; t += GD) {
        int b, x0, y0;
        tile_origin(t, b, x0, y0);
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
        __syncthreads();                                        // barrier 1: z1 tile (and, first time, the weights) visible
        if (t + GD < ntiles) prefetch_z(t + GD);

        // ================= S1: 3x3 on z1.  chunk j = units (2j, 2j+1); unit u = (tap u / NO, octet u % NO); k half = lhi
        f32x16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        {
            bf16x8 ar[2][2][3];                                 // global A ring (W0 not resident): [slot][m][plane]
            auto load_a0 = [&](int j, bf16x8 (&dst)[2][3]) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        dst[m][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w0, a64_off, (unsigned)(j * G::CHUNK_B + pl * 2048 + m * 512), 0));
            };
            if constexpr (!G::W0_RES) load_a0(0, ar[0]);
#pragma unroll 1
            for (int j = 0; j < NC1; ++j) {
                const int u0 = 2 * j, u1 = (2 * j + 1 < NU) ? 2 * j + 1 : 2 * j;   // a missing second unit re-reads the first (weights 0)
                const int t0 = u0 / NO, o0 = u0 % NO, t1 = u1 / NO, o1 = u1 % NO;
                const int a0 = (o0 * 3 * NPZ + (t0 / 3) * ZC + (t0 % 3)) * 16, a1 = (o1 * 3 * NPZ + (t1 / 3) * ZC + (t1 % 3)) * 16;
                const unsigned char* bp = sZ + (lhi ? a1 : a0) + (wave * ZC + l31) * 16;
                bf16x8 bf[3], af[2][3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bf[pl] = *reinterpret_cast<const bf16x8*>(bp + pl * NPZ * 16);
                if constexpr (G::W0_RES) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            af[m][pl] = *reinterpret_cast<const bf16x8*>(sW0 + j * G::CHUNK_B + pl * 2048 + m * 512 + a64_off);
#pragma unroll
                    for (int m = 0; m < 2; ++m) { BFSR_SIX32(acc[m], af[m], bf) }
                } else {
                    if (j + 1 < NC1) load_a0(j + 1, ar[(j + 1) & 1]);
                    if (j & 1) {
#pragma unroll
                        for (int m = 0; m < 2; ++m) { BFSR_SIX32(acc[m], ar[1][m], bf) }
                    } else {
#pragma unroll
                        for (int m = 0; m < 2; ++m) { BFSR_SIX32(acc[m], ar[0][m], bf) }
                    }
                }
            }
        }
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
        if (t + GD < ntiles) prefetch_pre(t + GD);              // the next tile's hoisted partial flies under S2 ... S3
        // ================= S2: 1x1, K = 64 = 4 chunks in accumulator order
        f32x16 acc2[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[m][r] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x8 af[2][3];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    af[m][pl] = *reinterpret_cast<const bf16x8*>(sW2 + c * G::CHUNK_B + pl * 2048 + m * 512 + a64_off);
#pragma unroll
            for (int m = 0; m < 2; ++m) { BFSR_SIX32(acc2[m], af[m], b2[c]) }
            __builtin_amdgcn_sched_barrier(0);                  // keep the fragment reads of chunk c+1 behind this chunk's MFMAs
        }
        __syncthreads();                                        // barrier 2: nobody reads the z1 tile any more -> hid may be written
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
        __syncthreads();                                        // barrier 3: hid tile complete

        // ---- operands of this tile's pointwise chain (one thread per output pixel): issued here, consumed after S3
        const int py = y0 + tid / OW, px = x0 + tid % OW;
        const bool pw_on = tid < NPX && py < H && px < W;
        const long long pix = (long long)py * W + px;
        float x[C], fsh[C], fsr[C];
        {
            const unsigned vo = pw_on ? (unsigned)(pix * 4) : OOB;
            const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.z_in + (long long)b * p.z_in_bs), 0, (unsigned)(C * HW * 4), 0x00020000);
#pragma unroll
            for (int c = 0; c < C; ++c) x[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, vo, (unsigned)(c * HW * 4), 0));
            if (p.h_ft) {
                const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.h_ft + (long long)b * p.h_ft_bs), 0, (unsigned)(2 * C * HW * 4), 0x00020000);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    fsh[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, vo, (unsigned)((2 * c) * HW * 4), 0));
                    fsr[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, vo, (unsigned)((2 * c + 1) * HW * 4), 0));
                }
            }
        }

        // ================= S3: Conv2dZeros 64 -> CO2 on 16x16x32 tiles.  chunk = (tap, 32-channel half): k group lq = octet half*4+lq
        // column tile n = (output row, 16-pixel half): wave w takes n = w and, for w < 4, n = w + 8
        {
            const int n_mine = wave < 4 ? 2 : 1;
            f32x4 acc4[2][MT];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc4[n][m][r] = 0.f;
            int nb[2];                                          // byte offset of (row, half*16 + l15) inside an (octet, plane) slab
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int nt = wave + 8 * n;
                nb[n] = (((nt >> 1) * HC) + (nt & 1) * 16 + l15) * 16;
            }
            constexpr int RING = MT == 1 ? 3 : 2;
            bf16x8 ar[RING][3][MT];
            auto load_a4 = [&](int ck, bf16x8 (&dst)[3][MT]) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        dst[pl][m] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w4, a4_off, (unsigned)((ck * G::W4_CHUNK + pl * 4 * MW * 8 + m * 16 * 8) * 2), 0));
            };
#pragma unroll
            for (int i = 0; i < RING - 1; ++i) load_a4(i, ar[i]);
#pragma unroll
            for (int ck = 0; ck < 18; ++ck) {
                if (ck + RING - 1 < 18) load_a4(ck + RING - 1, ar[(ck + RING - 1) % RING]);
                const int tap = ck >> 1, hf = ck & 1;
                const int toff = ((tap / 3) * HC + (tap % 3)) * 16 + (hf * 4 + lq) * NPH * 16;
                bf16x8 bf[2][3];
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        bf[n][pl] = *reinterpret_cast<const bf16x8*>(sHid + pl * 8 * NPH * 16 + toff + nb[n < n_mine ? n : 0]);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (n < n_mine) {
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
#define BFSR_T(PA_, PB_) acc4[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[ck % RING][PA_][m], bf[n][PB_], acc4[n][m], 0, 0, 0);
                            BFSR_T(2, 0) BFSR_T(0, 2) BFSR_T(1, 1) BFSR_T(1, 0) BFSR_T(0, 1) BFSR_T(0, 0)
#undef BFSR_T
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- E3: accumulator layout of 16x16: lane (l15, lq) holds rows 4*lq + i of column l15 -> h tile in LDS
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                if (n < n_mine) {
                    const int nt = wave + 8 * n;
                    const int col = (nt & 1) * 16 + l15;
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int co = m * 16 + lq * 4 + i;
                            if (co < CO2 && col < OW) sH[co * NPX + (nt >> 1) * OW + col] = (acc4[n][m][i] + p.bias[co]) * p.post_scale[co];
                        }
                }
            }
        }
        __syncthreads();                                        // barrier 4: h complete; every wave is done reading hid

        // ---- pointwise chain, one thread per pixel: identical arithmetic to flow_pointwise_kernel / coupling_tail_kernel
        if (pw_on) {
            const float eps = p.eps;
            const bool hf = p.h_ft != nullptr;
            const float* ha = sH + tid;
            float* zo = p.z_out + (long long)b * p.z_out_bs + pix;
            if (p.reverse) {
#pragma unroll
                for (int j = 0; j < CC; ++j) x[CN + j] = x[CN + j] / sigmoid_scale(ha[(2 * j + 1) * NPX], eps) - ha[(2 * j) * NPX];
                if (hf) {
#pragma unroll
                    for (int c = 0; c < C; ++c) x[c] = x[c] / sigmoid_scale(fsr[c], eps) - fsh[c];
                }
            } else {
#pragma unroll
                for (int j = 0; j < CC; ++j) x[CN + j] = (x[CN + j] + ha[(2 * j) * NPX]) * sigmoid_scale(ha[(2 * j + 1) * NPX], eps);
                if (p.an_bias) {
#pragma unroll
                    for (int c = 0; c < C; ++c) x[c] = (x[c] + p.an_bias[c]) * p.an_escale[c];
                }
            }
            if (p.wmat) {
                const float* __restrict__ w = p.wmat;
#pragma unroll
                for (int i = 0; i < C; ++i) {
                    float y = 0.f;
#pragma unroll
                    for (int j = 0; j < C; ++j) y = fmaf(w[i * C + j], x[j], y);
                    if (p.reverse) {
                        if (p.an_bias) y = y * p.an_escale[i] - p.an_bias[i];
                    } else if (hf) {
                        y = (y + fsh[i]) * sigmoid_scale(fsr[i], eps);
                    }
                    zo[(long long)i * HW] = y;
                }
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float y = x[c];
                    if (p.reverse) {
                        if (p.an_bias) y = y * p.an_escale[c] - p.an_bias[c];
                    } else if (hf) {
                        y = (y + fsh[c]) * sigmoid_scale(fsr[c], eps);
                    }
                    zo[(long long)c * HW] = y;
                }
            }
        }
        // the next iteration's z1 staging overwrites the hid region: every wave passed barrier 4 after its last hid read, and sH is
        // next written behind three more barriers
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
        const long long span = ((long long)(a->B - 1) * (a->z_in_bs > a->z_out_bs ? a->z_in_bs : a->z_out_bs) + (long long)a->C * a->H * a->W) * 4;
        const char* i0 = reinterpret_cast<const char*>(a->z_in);
        const char* o0 = reinterpret_cast<const char*>(a->z_out);
        if (i0 < o0 + span && o0 < i0 + span) return -1;
    }
    switch (a->C) {
        case 12: return launch_step<1, 12>(*a, st);
        case 24: return launch_step<2, 24>(*a, st);
        default: return -1;
    }
}
