// conv_bf16x3_ps.hip -- producer/consumer ("warp specialised"), persistent variant of the fp32-accurate 3xBF16 3x3 conv for
// 32-output-channel workgroup tiles (the RDB convs of the RRDB encoder, RRDBNet_arch.py:39-45; same arithmetic, same packed
// weights and same fused epilogue as conv_bf16x3_kernel<3,1,1,..>).
//
// Why: PMC on conv_bf16x3_kernel at the RDB shapes (profiles/r02_a_pmc_rdb.txt) shows the matrix pipe 57-64 % busy; an ablation of
// an LDS-DMA variant (tools/exp) shows the bare fragment-read + MFMA loop of this tile shape reaches 200-250 TFLOP/s, and that
// what costs the rest is staging work issued by the SAME waves that feed the matrix pipe (global loads, the fp32 -> 3 x bf16 split,
// LDS writes, two barriers per 16-channel chunk, a cold prologue per tile).  Here the roles are split:
//   * NC = 8 consumer waves (wave w owns output row w of an 8 x 32 pixel tile, M = 32 couts): per chunk ONE barrier, then only
//     ds_read_b128 + v_mfma_f32_32x32x16_bf16 (software-pipelined one tap ahead);
//   * NP producer waves: global loads of chunk s+1 (kept in registers across the barrier), split into the three bf16 planes,
//     ds_write into the other LDS stage, weights by LDS-DMA (they are bf16 already) -- all under the consumers' MFMAs of chunk s;
//   * the workgroup is persistent (one per CU) and walks its tiles in an XCD-aware order; the producers run ahead across tile
//     boundaries, so a tile's first chunk is staged under the previous tile's last chunk and epilogue.
// LDS: 2 stages x (input [3][2][340][8] bf16 = 32 640 B + weights [3][9][2][32][8] bf16 = 27 648 B) = 120 576 B.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/bfsr_hip.h"
#include "launch_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

namespace {

constexpr int CK = 16, NC = 8, TH = 8, PW = 34, NPOS = (TH + 2) * PW;      // 340 staged positions per chunk
constexpr int SUB = NPOS * 16;                  // bytes of one (plane, k half) sub-image
constexpr int IN_BYTES = 6 * SUB;               // 32 640
constexpr int WPL = 9 * 2 * 32 * 16;            // bytes of one weight plane per chunk
constexpr int W_BYTES = 3 * WPL;                // 27 648
constexpr int STAGE = IN_BYTES + W_BYTES;       // 60 288
constexpr int LDS_TOTAL = 2 * STAGE;            // 120 576
constexpr int W_PIECES = W_BYTES / 1024;        // 27 LDS-DMA wave-instructions per weight slab
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l)
{
    h = (__bf16)v;
    const float r1 = v - (float)h;        // exact
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);          // exact residual, <= 8 significant bits
}

struct Item { int cg, b, x0, y0; };

template <int NP>
__global__ __launch_bounds__((NC + NP) * 64, 1) void conv_bf16x3_ps_kernel(BfsrConvArgs p, int tiles_x, int tiles_y, int groups, int nitems)
{
    constexpr int NPT = NP * 64;                                 // producer threads
    constexpr int PPT = (NPOS + NPT - 1) / NPT;                  // positions per producer thread (1 or 2)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
    const int slot = (int)bfsr::xcd_order(blockIdx.x, (unsigned)G);
    const int H = p.H, W = p.W, sh = p.in_shift, Ws = W >> sh;
    const long long cs_in = (long long)(H >> sh) * Ws;
    const int Cin = p.Cin, nchunk = (Cin + CK - 1) / CK;
    if (slot >= nitems) return;
    const int my_items = (nitems - slot + G - 1) / G;            // items slot, slot+G, ...

    auto decode = [&](int it) {
        Item r;
        r.cg = it % groups; int t = it / groups;
        const int ty = t % tiles_y; t /= tiles_y;
        r.x0 = (t % tiles_x) * 32; r.y0 = ty * TH; r.b = t / tiles_x;
        return r;
    };

    if (wave >= NC) {
        // =============================== producers ==========================================================================
        const int ptid = tid - NC * 64, pwave = wave - NC;
        const unsigned cs_bytes = (unsigned)(cs_in * 4);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0,
                                                                              (unsigned)((long long)groups * nchunk * W_BYTES), 0x00020000);
        unsigned voff[PPT];
        float vin[PPT][CK];
        __amdgpu_buffer_rsrc_t rs_in;
        int cg = 0;
        auto setup = [&](const Item& it) {
            rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (long long)it.b * p.x_bs), 0,
                                                      (unsigned)((long long)Cin * cs_in * 4), 0x00020000);
            cg = it.cg;
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const int pos = ptid + i * NPT;
                const int r = pos / PW, c = pos - r * PW;
                const int gy = it.y0 + r - 1, gx = it.x0 + c - 1;
                const bool ok = pos < NPOS && gy >= 0 && gy < H && gx >= 0 && gx < W;
                voff[i] = ok ? (unsigned)((gy >> sh) * Ws + (gx >> sh)) * 4u : OOB;
            }
        };
        auto load_regs = [&](int k) {                            // R(step): global -> registers (channels past Cin read as 0)
            const unsigned sbase = (unsigned)(k * CK) * cs_bytes;
#pragma unroll
            for (int c = 0; c < CK; ++c)
#pragma unroll
                for (int i = 0; i < PPT; ++i)
                    vin[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, voff[i], sbase + (unsigned)c * cs_bytes, 0));
        };
        auto write_stage = [&](int k, int buf) {                 // W(step): weights by LDS-DMA, registers -> split -> LDS
            unsigned char* base = smem + buf * STAGE;
            const unsigned wsoff = (unsigned)(cg * nchunk + k) * (unsigned)W_BYTES;
#pragma unroll
            for (int j = 0; j < (W_PIECES + NP - 1) / NP; ++j) {
                const int piece = pwave + j * NP;
                if (piece < W_PIECES)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(base + IN_BYTES + piece * 1024), 16,
                                                             (unsigned)lane * 16u + (unsigned)piece * 1024u, wsoff, 0, 0);
            }
            __bf16* sIn = reinterpret_cast<__bf16*>(base);
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const int pos = ptid + i * NPT;
                if (i < PPT - 1 || pos < NPOS) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        bf16x8 h8, m8, l8;
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            __bf16 h, m, l;
                            split3(vin[i][hf * 8 + c], h, m, l);
                            h8[c] = h; m8[c] = m; l8[c] = l;
                        }
                        *reinterpret_cast<bf16x8*>(sIn + ((0 * 2 + hf) * NPOS + pos) * 8) = h8;
                        *reinterpret_cast<bf16x8*>(sIn + ((1 * 2 + hf) * NPOS + pos) * 8) = m8;
                        *reinterpret_cast<bf16x8*>(sIn + ((2 * 2 + hf) * NPOS + pos) * 8) = l8;
                    }
                }
            }
        };
        // steps s = 0 .. S-1 enumerate (item, chunk) of this workgroup's items in order; stage of step s = s & 1
        const int S = my_items * nchunk;
        int it = slot, k = 0;                                    // (item, chunk) of the step whose registers are loaded next
        Item cur = decode(it);
        setup(cur);
        load_regs(0);                                            // R(0)
        for (int s = 0; s < S; ++s) {
            // after barrier s-1 the consumers are done with stage s&1 (they computed step s-2 from it)
            write_stage(k, s & 1);                               // W(s)  (waits for R(s) at first use)
            // next step's (item, chunk)
            if (++k == nchunk) { k = 0; it += G; if (s + 1 < S) { cur = decode(it); setup(cur); } }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // weight DMA landed, ds_writes done
            if (s + 1 < S) load_regs(k);                         // R(s+1): in flight across the barrier
            __builtin_amdgcn_s_barrier();                        // barrier s: stage s&1 is ready
        }
        return;
    }

    // =================================== consumers ==========================================================================
    const int l31 = lane & 31, lhi = lane >> 5;
    int it = slot;
    int step = 0;
    for (int n = 0; n < my_items; ++n, it += G) {
        const Item cur = decode(it);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int k = 0; k < nchunk; ++k, ++step) {
            __builtin_amdgcn_s_barrier();                        // barrier `step`: the producers have filled stage step&1
            const unsigned char* sIn = smem + (step & 1) * STAGE;
            const unsigned char* sW = sIn + IN_BYTES;
            const unsigned char* inB = sIn + (lhi * NPOS + wave * PW + l31) * 16;         // plane stride 2*SUB
            const unsigned char* wA = sW + (lhi * 32 + l31) * 16;                         // [plane][tap][k half][32][8]
            bf16x8 bfr[3][3], afr[2][3];
            auto load_b = [&](int dx) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int r = 0; r < 3; ++r)
                        bfr[pl][r] = *reinterpret_cast<const bf16x8*>(inB + pl * 2 * SUB + (r * PW + dx) * 16);
            };
            auto load_a = [&](int b_, int dx, int dy) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    afr[b_][pl] = *reinterpret_cast<const bf16x8*>(wA + pl * WPL + (dy * 3 + dx) * 1024);
            };
            load_b(0);
            load_a(0, 0, 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dx = t / 3, dy = t % 3, ab = t & 1;
                if (t > 0 && dy == 0) load_b(dx);
                if (t + 1 < 9) load_a(ab ^ 1, (t + 1) / 3, (t + 1) % 3);
                __builtin_amdgcn_sched_barrier(0);
#define BFSR_TERM(PA_, PB_) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ab][PA_], bfr[PB_][dy], acc, 0, 0, 0);
                BFSR_TERM(2, 0) BFSR_TERM(0, 2) BFSR_TERM(1, 1) BFSR_TERM(1, 0) BFSR_TERM(0, 1) BFSR_TERM(0, 0)
#undef BFSR_TERM
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue (fp32, identical stage order to conv_bf16x3_kernel); the producers are already staging the next tile
        const long long HW = (long long)H * W;
        const int gx = cur.x0 + l31, gy = cur.y0 + wave;
        if (gx < W && gy < H) {
            const float slope = p.act == BFSR_ACT_NONE ? 1.f : (p.act == BFSR_ACT_RELU ? 0.f : p.slope);
            const float4* __restrict__ epi = reinterpret_cast<const float4*>(p.epi);
            const unsigned out_bytes = (unsigned)((long long)p.Cout * HW * 4);
            const int b = cur.b;
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
                for (int r = 0; r < 16; ++r) {
                    const int co = cur.cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (co >= p.Cout) continue;
                    float4 q0 = make_float4(0.f, 0.f, 1.f, 0.f); float q1 = 1.f;
                    if (epi) { q0 = epi[co * 2]; q1 = epi[co * 2 + 1].x; }
                    const long long o = (long long)co * HW + (long long)gy * W + gx;
                    float v = acc[r];
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
            };
            if (tensors) run_epilogue(std::true_type{});
            else run_epilogue(std::false_type{});
        }
    }
}

template <int NP>
int launch_ps(const BfsrConvArgs& a, hipStream_t st, int wgs)
{
    static std::atomic<unsigned long long> lds_done{0};
    if (bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_bf16x3_ps_kernel<NP>), LDS_TOTAL, lds_done) != 0) return -1;
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + TH - 1) / TH;
    const int groups = (a.Cout + 31) / 32;
    const long long nitems = (long long)tiles_x * tiles_y * groups * a.B;
    if (nitems <= 0 || nitems > 0x7fffffffLL) return -1;
    const long long grid = nitems < wgs ? nitems : wgs;
    hipLaunchKernelGGL((conv_bf16x3_ps_kernel<NP>), dim3((unsigned)grid), dim3((NC + NP) * 64), LDS_TOTAL, st, a, tiles_x, tiles_y, groups, (int)nitems);
    return (int)hipGetLastError();
}

}  // namespace

// Called by bfsr_conv2d_bf16x3 (conv_bf16x3.hip) for KS = 3, mtile = 1 once the arguments have been validated.
// np = producer waves (4 or 8); wgs = persistent workgroups (0 = one per CU).
int bfsr_conv2d_bf16x3_ps(const BfsrConvArgs* a, hipStream_t st, int np, int wgs)
{
    if ((long long)((a->Cout + 31) / 32) * ((a->Cin + CK - 1) / CK) * W_BYTES >= (1LL << 32)) return -1;
    if (wgs <= 0) {
        int dev = 0;
        wgs = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&wgs, hipDeviceAttributeMultiprocessorCount, dev);
    }
    return np == 8 ? launch_ps<8>(*a, st, wgs) : launch_ps<4>(*a, st, wgs);
}
