// conv_h2s.hip -- 3x3 'same' conv on the fp16 matrix pipe (v_mfma_f32_32x32x16_f16, fp32 accumulation) over activations that are
// STORED in fp16, for the dense blocks of the RRDB encoder on the reduced-precision path (LINF precision='fp16', BASELINE
// config 5; LINF-LP/models/rrdb.py:38-76).
//
// Why: conv_f16_kernel reads fp32 NCHW activations and rounds them to fp16 while staging (global load -> VALU -> ds_write,
// two barriers per 16-channel chunk); on the RDB shapes it is bound by that staging, not by the matrix pipe (one product per
// operand pair instead of the six of the 3xBF16 path, yet only 1.6x faster per pixel than conv3x3_x3s_kernel).  Here the
// producer's epilogue writes fp16 once and the tiles are staged by LDS-DMA (`buffer_load ... lds`), like conv_x3s.hip.
//
// h2 tensor layout: [B][C/8][2 planes hi,lo][H][W][8] fp16 with hi = fp16(x), lo = fp16(x - hi): x ~ hi + lo to 22 significant
// bits.  A conv reads ONLY the hi plane (= exactly the operand conv_f16_kernel forms by rounding at staging time, so the two
// kernels contract identical numbers); the lo plane exists for the tensors that are residual operands (the RDB / RRDB trunk:
// `x5*0.2 + x`), whose sum must not be re-rounded to 11 bits at each of the 69 dense blocks.  Outputs that only ever feed
// convs (x1..x4 of a dense block) are written hi-only (y_fmt 2).
//
// GEMM view per workgroup: M = 32 output channels, N = 16 rows x 32 pixels (compute wave w owns rows 2w, 2w+1: every weight
// fragment feeds two MFMAs and the four input rows of a tap column feed six), K = 16 channels per chunk x 9 taps.
// LDS stage = input [2 k halves][640 positions][8] (18 x 34 tile, padded to 10 x 64 positions) + weights [9 taps][2][32][8]
// = 20 480 + 9 216 B; FIVE stages (148 480 B): a chunk is only 576 MFMA cycles per wave, shorter than an HBM round trip, so the
// four dedicated loader waves run three chunks ahead (across tile boundaries) and park on `s_waitcnt vmcnt(N)` with N = the
// pieces of the later stages.  One barrier per chunk, passed by the compute waves one pipeline step EARLY (before the last tap
// column of the previous chunk) so that, with a register double buffer of fragments, LDS latency never shows at chunk or tile
// boundaries; the eight compute waves issue nothing but ds_read_b128 + MFMA (21 reads per 18 MFMAs: 58 % of the LDS read rate at
// full matrix rate).  Persistent workgroups, XCD-aware order.  Measured (tools/exp/h2s_bench.py, profiles/r02_d_h2s_*.txt):
// K loop alone 1.2-1.45 PFLOP/s, whole kernel 0.6-0.8 PFLOP/s and ~3.2 TB/s algorithmic at 128 x 128x128 -- the epilogue
// (VALU-issue bound, all compute waves in it at once) is the remaining 35 %.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "../../include/bfsr_hip.h"
#include "launch_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

namespace {

constexpr int NW = 8, NLW = 4;                  // compute waves, loader waves
constexpr int TH = 16, PW = 34, NPOS = (TH + 2) * PW, NG = 10, NPOSP = NG * 64;
constexpr int SUB = NPOSP * 16;                 // bytes of one k-half sub-image (8 channels of every tile position)
constexpr int IN_BYTES = 2 * SUB;               // 20 480
constexpr unsigned OOB = 0x80000000u;

// MT = 32-cout M tiles per workgroup.  1: the dense blocks' 32-channel convs.  2 (round 6): 64-channel convs (conv5 of a dense block, trunk convs,
// coef | freq, the priors) -- the input tile is staged ONCE for 64 output channels, four MFMAs per four LDS reads instead of two per three; the ring
// has four stages then (a chunk is twice the matrix work, so two chunks in flight cover the same time as three did).
template <int MT> struct SGeo {
    static constexpr int W_BYTES = 9 * 1024 * MT;               // weights of one chunk: [9 taps][2 k halves][32 MT couts][8] fp16
    static constexpr int STAGE = IN_BYTES + W_BYTES;            // 29 696 | 38 912
    static constexpr int NS = MT == 1 ? 5 : 4;                  // LDS ring: 148 480 | 155 648 B
    static constexpr int LDS_TOTAL = NS * STAGE;
    static constexpr int NPIECE = 20 + W_BYTES / 1024;          // 1-KiB LDS-DMA pieces per stage: 29 | 38
    static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
};

struct Item { int cg, b, x0, y0; };

__device__ __forceinline__ void split2(float v, _Float16& h, _Float16& l)
{
    const float hf = bfsr::pin_f16(v);
    h = (_Float16)hf;
    l = (_Float16)(v - hf);
}

// `s_waitcnt vmcnt(n)` for a wave-uniform run-time n (the instruction takes an immediate); n = stages in flight x pieces per stage
__device__ __forceinline__ void wait_vmcnt(int n)
{
    switch (n) {
#define W_(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
        W_(5) W_(6) W_(7) W_(8) W_(9) W_(10) W_(12) W_(14) W_(15) W_(16) W_(18) W_(20)
#undef W_
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// ---- loader waves (shared by both kernels).  Barrier c = "chunk c is in LDS, and every compute wave is past its reads of chunk c-2"
// (they arrive one pipeline step early, see below), so the stage of chunk c-2 is refilled with chunk c+NS-2: three stages in flight.
// `s_waitcnt vmcnt(N)` stands for "everything but the youngest N/np stages has landed" because a loader issues the same number np
// of pieces for every chunk -- and only ONE KIND of load: LDS-DMA pieces and ordinary register loads of one wave do NOT complete
// in issue order relative to each other (measured: a wave mixing them passed the barrier with input pieces still in flight).
//   Four DMA loaders: piece i of a stage (i < 20: input position group i>>1, k half i&1; else weight piece i-20) -> loader i % 4.
// RES (round 6): the conv's WHOLE weight tensor resident in LDS (convs with one output-channel group and <= 8 chunks: conv1..3 of a dense block, the
// 64 -> 64 convs): staged once per workgroup in front of the ring, whose stages then hold input only (`ns` of them, as many as fit).  As streamed
// pieces the weights were 30-47 % of the L2 -> LDS fill of these convs, and the fill (5.4 of ~6.4 TB/s at conv1) is what they are bound by.
template <int MT, int RES, class Decode>
__device__ __forceinline__ void h2s_loader_wave(const BfsrConvX3Args& p, unsigned char* smem, int wave, int lane, int slot, int G, int groups,
                                                int nchunk, int T, unsigned HW16, int ns, Decode decode)
{
    const int H = p.H, W = p.W;
    const int ld = wave - NW;
    constexpr int W_BYTES = SGeo<MT>::W_BYTES, STAGE = RES ? IN_BYTES : SGeo<MT>::STAGE, NPIECE = RES ? 20 : SGeo<MT>::NPIECE;
    const int NS = RES ? ns : SGeo<MT>::NS;
    unsigned char* ring = smem + (RES ? nchunk * W_BYTES : 0);
    // ---- DMA loaders
    constexpr int ND = NLW;                                          // DMA loaders
    constexpr int NPALL = NPIECE;                                    // pieces they share
    const int np = (NPALL - ld + ND - 1) / ND;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w), 0,
                                                                          (unsigned)((long long)groups * nchunk * W_BYTES), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_in;
    constexpr int NJI = (20 + NLW - 2) / (NLW - 1);                  // input pieces per loader, at most (7; a dependent bound here makes hipcc 7.2 drop the HOST stub silently)
    unsigned vg[NJI];
    int cg_ = 0;
    auto lsetup = [&](const Item& it) {
        const unsigned short* xb = p.x + (long long)it.b * p.x_bs;
        rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xb), 0, (unsigned)(p.Cin >> 3) * 2u * HW16, 0x00020000);
        cg_ = it.cg;
#pragma unroll
        for (int j = 0; j < NJI; ++j) {
            const int i = ld + j * ND;
            const int pos = (i >> 1) * 64 + lane;
            const int r = pos / PW, c = pos - r * PW;
            const int gy = it.y0 + r - 1, gx = it.x0 + c - 1;
            const bool ok = i < 20 && pos < NPOS && gy >= 0 && gy < H && gx >= 0 && gx < W;
            vg[j] = ok ? (unsigned)(gy * W + gx) * 16u : OOB;        // out of range -> the DMA writes zeros (= the padding)
        }
    };
    if (RES) {                                                        // the resident weights: issued first, so every later `vmcnt` wait covers them
        const int nwp = nchunk * (W_BYTES / 1024);
        for (int pi = ld; pi < nwp; pi += ND)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(smem + pi * 1024), 16, (unsigned)lane * 16u + (unsigned)pi * 1024u, 0, 0, 0);
    }
    auto lstage = [&](int k, int buf, int skip) {
        unsigned char* base = ring + buf * STAGE;
        const unsigned wsoff = (unsigned)(cg_ * nchunk + k) * (unsigned)W_BYTES;
#pragma unroll
        for (int j = 0; j < (NPALL + ND - 1) / ND; ++j) {
            const int i = ld + j * ND;
            if (i >= NPALL) continue;
            if (j < NJI && i < 20) {
                if (skip & 1) continue;
                const unsigned soff = (unsigned)(2 * k + (i & 1)) * 2u * HW16;                  // hi plane of channel octet 2k + (i&1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(base + (i & 1) * SUB + (i >> 1) * 1024), 16, vg[j], soff, 0, 0);
            } else {
                if (skip & 2) continue;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(base + IN_BYTES + (i - 20) * 1024), 16,
                                                         (unsigned)lane * 16u + (unsigned)(i - 20) * 1024u, wsoff, 0, 0);
            }
        }
    };
    int issued = 0, iss_it = slot, iss_k = 0, iss_buf = 0;
    auto issue = [&]() {
        if (iss_k == 0) lsetup(decode(iss_it));
        lstage(iss_k, iss_buf, 0);
        ++issued;
        iss_buf = iss_buf + 1 == NS ? 0 : iss_buf + 1;
        if (++iss_k == nchunk) { iss_k = 0; iss_it += G; }
    };
    for (int i = 0; i < NS - 2 && issued < T; ++i) issue();
    for (int c = 0; c < T; ++c) {
        wait_vmcnt((issued - c - 1) * np);                           // all but the stages issued after chunk c's
        __builtin_amdgcn_s_barrier();
        if (issued < T) issue();
    }
}

template <int MT, int RES>
__global__ __launch_bounds__((NW + NLW) * 64, 1) void conv3x3_h2s_kernel(BfsrConvX3Args p, int tiles_x, int tiles_y, int groups, int nitems, int ns)
{
    constexpr int STAGE = RES ? IN_BYTES : SGeo<MT>::STAGE, W_BYTES_ = SGeo<MT>::W_BYTES;
    const int NS = RES ? ns : SGeo<MT>::NS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int G = gridDim.x;
    const int slot = (int)bfsr::xcd_order(blockIdx.x, (unsigned)G);      // see conv_x3s.hip: one XCD walks neighbouring tiles
    if (slot >= nitems) return;
    const int H = p.H, W = p.W;
    const unsigned HW16 = (unsigned)(H * W) * 16u;                       // bytes of one (octet, plane) image
    const int nchunk = p.Cin >> 4;
    const int n_mine = (nitems - slot + G - 1) / G;
    const int T = n_mine * nchunk;                                       // chunks this workgroup consumes = barriers every wave passes

    auto decode = [&](int it) {
        Item r;
        r.cg = it % groups; int t = it / groups;
        const int ty = t % tiles_y; t /= tiles_y;
        r.x0 = (t % tiles_x) * 32; r.y0 = ty * TH; r.b = t / tiles_x;
        return r;
    };

    if (wave >= NW) {
        h2s_loader_wave<MT, RES>(p, smem, wave, lane, slot, G, groups, nchunk, T, HW16, ns, decode);
        return;
    }

    // ---- compute waves: a software pipeline of (chunk, dx) steps.  A step = the 4 input rows of tap column dx (B fragments) and the
    // 3*MT weight fragments of that column -> 6*MT MFMAs; the fragments of step s+1 are read from LDS while the MFMAs of step s run,
    // across chunk and tile boundaries: barrier c+1 is passed BEFORE the last step of chunk c so that the first fragments of chunk
    // c+1 are in flight under it.  Register double buffer: chunk parity P (compile-time, two instantiations of the body).
    half8 bq[2][4], aq[2][3 * MT];
    const unsigned char* ring = smem + (RES ? nchunk * W_BYTES_ : 0);
    auto load_step = [&](auto buf_, int st, int dx, int kw) {            // kw: the chunk's index inside its item (where its resident weights sit)
        constexpr int BUF = decltype(buf_)::value;
        const unsigned char* sIn = ring + st * STAGE;
        const unsigned char* inB = sIn + (lhi * NPOSP + 2 * wave * PW + l31 + dx) * 16;
        const unsigned char* wA = (RES ? smem + kw * W_BYTES_ : sIn + IN_BYTES) + (lhi * MT * 32 + l31) * 16 + dx * 3 * (MT * 1024);   // tap = dx*3 + dy
#pragma unroll
        for (int r = 0; r < 4; ++r) bq[BUF][r] = *reinterpret_cast<const half8*>(inB + r * PW * 16);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int m = 0; m < MT; ++m) aq[BUF][dy * MT + m] = *reinterpret_cast<const half8*>(wA + dy * (MT * 1024) + m * 512);
    };
    f32x16 acc[MT][2];
    auto mfma_step = [&](auto buf_) {
        constexpr int BUF = decltype(buf_)::value;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[BUF][dy * MT + m], bq[BUF][dy], acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[BUF][dy * MT + m], bq[BUF][dy + 1], acc[m][1], 0, 0, 0);
            }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    int c = 0, st = 0, kc = 0;                                           // chunk counter, its LDS stage, its index inside the item
    auto chunk_body = [&](auto p_, auto q_, bool pf) {                   // p_ = buffer of this chunk's first step, q_ = the other; pf: prefetch the next chunk's first step
        const int nst = st + 1 == NS ? 0 : st + 1;
        const int nkc = kc + 1 == nchunk ? 0 : kc + 1;
        load_step(q_, st, 1, kc);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(p_);
        __builtin_amdgcn_sched_barrier(0);
        load_step(p_, st, 2, kc);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(q_);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < T) {
            __builtin_amdgcn_s_barrier();
            if (pf) load_step(q_, nst, 0, nkc);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(p_);
        __builtin_amdgcn_sched_barrier(0);
        st = nst; kc = nkc; ++c;
    };
    const float slope = p.act == BFSR_ACT_NONE ? 1.f : (p.act == BFSR_ACT_RELU ? 0.f : p.slope);
    const float4* __restrict__ epi = reinterpret_cast<const float4*>(p.epi);
    const long long HW = (long long)H * W;
    __builtin_amdgcn_s_barrier();                                        // barrier 0
    load_step(I0(), 0, 0, 0);
    for (int it = slot; it < nitems; it += G) {
        const Item cur = decode(it);
        float4 pm[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int idx = (cur.cg * MT + m) * 64 + lane;
            pm[m] = (lane & 1) ? make_float4(1.f, 0.f, 0.f, 0.f) : make_float4(0.f, 0.f, 1.f, 0.f);
            if (epi && (idx >> 1) < p.Cout) pm[m] = epi[idx];
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[m][0][r] = 0.f; acc[m][1][r] = 0.f; }
        for (int k = 0; k < nchunk; k += 2) {                            // Cin % 32 == 0: the register-buffer parity is static
            chunk_body(I0(), I1(), true);
            chunk_body(I1(), I0(), k + 2 < nchunk);                      // the last chunk of a tile leaves the registers to the epilogue
        }

        // ---- epilogue (the loaders are already staging the next item).  acc[r] = channel (r&3) + 8(r>>2) + 4*lhi of pixel
        // (row, x0 + l31); v_permlane32_swap pairs the half-waves so that each lane holds two complete channel octets.
        // Inline asm for the reason given in conv_x3s.hip (the builtin is folded on MFMA results); the pads cover the
        // MFMA -> VALU-read and VALU-write -> permlane hazards the compiler does not see around asm.
        // Per-channel parameters: the coalesced load issued at the top of the tile, handed to the lanes that need them by
        // ds_bpermute (reading epi[co] directly = 16 broadcast loads of 1 KiB per octet, after the last MFMA).  Residual
        // operands: ALL groups' loads are issued before the first is used -- one HBM round trip per residual and tile, not one
        // per octet (measured: the serialised version cost more than the K loop of the 192-channel conv).
        const bool fast_slope = slope >= 0.f && slope <= 1.f;
        auto fetch = [&](float val, int src_lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane * 4, __float_as_int(val))); };
        int lh = lhi, lx = l31;                                          // opaque copies: keeps the per-lane address arithmetic of the epilogue
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(lh), "+v"(lx));                           // INSIDE the tile loop (hoisted, it is spilled to scratch and reloaded here)
#endif
        const int gx = cur.x0 + lx;
        // every global access of the epilogue is a buffer instruction whose VGPR offset is out of range for pixels outside the image (and whose
        // descriptor ends at Cout channels): no `if (inside)` branch per access (conv3x3_h2x_kernel's epilogue, round 4)
        unsigned vo16[2], vo4[2];                                        // per row j: byte offset of (half-wave octet lh, pixel) in 16-byte / 4-byte units
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gy = cur.y0 + 2 * wave + j;
            const bool ok = gy < H && gx < W;
            vo16[j] = ok ? (unsigned)(((long long)lh * 2 * HW + (long long)gy * W + gx) * 16) : OOB;
            vo4[j] = ok ? (unsigned)(((long long)lh * 8 * HW + (long long)gy * W + gx) * 4) : OOB;
        }
        const unsigned h2_bytes = (unsigned)((long long)(p.Cout >> 3) * 2 * HW * 16);
        auto res_rsrc = [&](const unsigned short* res, long long bs) {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(res + (long long)cur.b * bs), 0, h2_bytes, 0x00020000);
        };
        typedef unsigned u32x4s_ __attribute__((ext_vector_type(4)));
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");       // MFMA result -> VALU read inside the asm below: 20 wait states (>= 19 of a 16-pass XDL op), self-sufficient
        // ONE (M TILE, ROW) AT A TIME (round 6): 16 results per lane, their residual operands and parameters -- with MT = 2 the all-at-once form of
        // round 3 spilled (64 accumulators + 64 registers of residual operands; 1.05 ms against 0.61 ms for conv5 at 128 x 128^2), and a per-M-tile
        // form still made hipcc park the residual loads in scratch one by one (`load, s_waitcnt vmcnt(0), scratch_store`: 0.71 ms)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bool plain = (lane & 1) ? pm[m].x == 1.f : (pm[m].y == 0.f && pm[m].z == 1.f && pm[m].w == 0.f);
            const bool bias_only = __all(plain);
            const bool fast = bias_only && fast_slope;
            const int oct0 = (cur.cg * MT + m) * 4;                      // first of this M tile's four channel octets (+ q*2 + lh)
            // the row's epilogue; `act(q, i, u)` = the per-channel parameter stage (bias / affine / activation / post-scale) of octet q, channel i
            auto do_row = [&](int j, auto&& act) {
                // operands of the first residual: issued FIRST, consumed after the parameter stage (their HBM round trip runs under the swaps,
                // the bpermute exchange and the bias/activation arithmetic instead of in front of the adds)
                half8 r1h[2], r1l[2];
                if (p.res1) {
                    const __amdgpu_buffer_rsrc_t rr = res_rsrc(p.res1, p.res1_bs);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const unsigned so = (unsigned)((oct0 + q * 2) * 2) * (unsigned)(HW * 16);
                        r1h[q] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rr, vo16[j], so, 0));
                        r1l[q] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rr, vo16[j], so + (unsigned)(HW * 16), 0));
                    }
                }
                float o[2][8];                                           // [octet q][channel]
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float lo = acc[m][j][8 * q + i], hi = acc[m][j][8 * q + 4 + i];
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
                        o[q][i] = lo;
                        o[q][4 + i] = hi;
                    }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[q][i] = act(q, i, o[q][i]);
                if (p.res1) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int i = 0; i < 8; ++i) o[q][i] = p.alpha1 * o[q][i] + ((float)r1h[q][i] + (float)r1l[q][i]);
                }
                if (p.res2) {
                    const __amdgpu_buffer_rsrc_t rr = res_rsrc(p.res2, p.res2_bs);
                    half8 rh[2], rl[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const unsigned so = (unsigned)((oct0 + q * 2) * 2) * (unsigned)(HW * 16);
                        rh[q] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rr, vo16[j], so, 0));
                        rl[q] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rr, vo16[j], so + (unsigned)(HW * 16), 0));
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int i = 0; i < 8; ++i) o[q][i] = p.alpha2 * o[q][i] + ((float)rh[q][i] + (float)rl[q][i]);
                }
                if (p.y_fmt != 0) {                                      // h2 output: both planes (1) or the hi plane only (2)
                    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned short*>(p.y) + (long long)cur.b * p.y_bs, 0, h2_bytes, 0x00020000);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const unsigned so = (unsigned)((oct0 + q * 2) * 2) * (unsigned)(HW * 16);
                        half8 h8, l8;
#pragma unroll
                        for (int i = 0; i < 8; ++i) { _Float16 h, l; split2(o[q][i], h, l); h8[i] = h; l8[i] = l; }
                        bfsr::store_b128(ry, __builtin_bit_cast(u32x4s_, h8), vo16[j], so);
                        if (p.y_fmt == 1) bfsr::store_b128(ry, __builtin_bit_cast(u32x4s_, l8), vo16[j], so + (unsigned)(HW * 16));
                    }
                } else {                                                 // fp32 NCHW: channels >= Cout fall beyond the descriptor
                    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(p.y) + (long long)cur.b * p.y_bs, 0,
                                                                                        (unsigned)((long long)p.Cout * HW * 4), 0x00020000);
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[q][i]), ry, vo4[j],
                                                                  (unsigned)(((oct0 + q * 2) * 8 + i) * HW * 4), 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            // two code paths WITHOUT shared default-valued parameter arrays: as `e1[i] = 0.f; ... if (!bias_only) e1[i] = fetch(...)` the defaults of
            // every (M tile, octet, channel) became live registers across the whole epilogue (128 with MT = 2).  The branch is wave-uniform (`fast`
            // comes from __all), so every lane takes part in each ds_bpermute exchange.
            if (fast) {                                                  // bias + (leaky) ReLU only: 3 VALU ops per channel instead of 8
                float e0[2][8];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int i = 0; i < 8; ++i) e0[q][i] = fetch(pm[m].x, ((q * 2 + lh) * 8 + i) * 2);   // lane holding this channel's first float4
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    do_row(j, [&](int q, int i, float v) { const float u = v + e0[q][i]; return fmaxf(u, u * slope); });   // = u > 0 ? u : u*slope for 0 <= slope <= 1
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    do_row(j, [&](int q, int i, float v) {
                        const int src = ((q * 2 + lh) * 8 + i) * 2;
                        const float e0 = fetch(pm[m].x, src), e1 = fetch(pm[m].y, src), e2 = fetch(pm[m].z, src), e3 = fetch(pm[m].w, src), e4 = fetch(pm[m].x, src + 1);
                        float u = v + e0;
                        u = (u + e1) * e2 + e3;
                        u = u > 0.f ? u : u * slope;
                        return u * e4;
                    });
            }
        }
        if (c < T) load_step(I0(), st, 0, 0);                            // first fragments of the next tile (its barrier is already behind us)
    }
}

// =====================================================================================================================
// conv3x3_h2x_kernel -- the same conv at FP32-CLASS accuracy on the fp16 matrix pipe: BOTH planes of the h2 activations and a
// two-term fp16 split of the weights, three products  lo*hi + hi*lo + hi*hi  (each exact in the fp32 accumulator).
//   x = hi + lo carries 22 significant bits (hi = fp16(x), lo = fp16(x - hi)); the dropped lo*lo term is <= 2^-22 relative.  fp16
//   has a narrow exponent: the weights are pre-multiplied by a power of two (their largest magnitude lands in [2^9, 2^10), so the
//   lo terms of all but negligible weights are normal numbers) and the accumulators are multiplied by the inverse power first thing
//   in the epilogue (`acc_scale`, exact); activations need |x| < 65504.  Measured end to end against an fp64 evaluation of the
//   SRFlow-LP pipeline this arithmetic is indistinguishable from fp32 (sr 3.7e-6 vs 3.6e-6 for the CPU's fp32)
//   at HALF the matrix instructions and 2/3 of the operand bytes of the 3xBF16 scheme (six products of three-term bf16 splits).
// Structure: the 16-row tile, two rows per compute wave and the fragment double buffer of conv3x3_h2s_kernel; LDS stage = input
// [2 planes][2 k halves][640 positions][8] + weights [2 planes][9 taps][2][32][8] = 40 960 + 18 432 B, two stages, ONE barrier per
// 16-channel chunk (conv_x3s.hip's protocol: the four loader waves stage chunk k+1 -- across tile boundaries -- while chunk k is
// in the matrix pipe; loader `ld` owns sub-image `ld` (plane, k half) of every position group and its share of the weight pieces).
// Per chunk and wave: 42 ds_read_b128 for 54 MFMAs.  Persistent workgroups, XCD-aware order.  Epilogue = conv3x3_h2s_kernel's.
#ifndef BFSR_H2X_ABL
#define BFSR_H2X_ABL 0                          // ablation builds only (tools/exp/h2x_abl.sh): bit 0 no fragment reads, 1 no MFMAs, 2 no DMA, 3 no epilogue
#endif
constexpr int X_IN = 4 * SUB;                   // 40 960
// XM = 32-cout M tiles per workgroup: 1, or 2 for the 64-channel convs (conv5 of a dense block, trunk convs): the input tile is staged
// once for 64 output channels -- the kernel is bound by the L2 -> LDS fill (59 KB per chunk and 32 couts), so bytes per MFMA count
template <int XM> struct XGeo {
    static constexpr int WPL = 9 * 1024 * XM;   // one weight plane of a chunk: [tap][m tile][k half][32][8]
    static constexpr int W = 2 * WPL;           // 18 432 | 36 864
    static constexpr int STAGE = X_IN + W;      // 59 392 | 77 824
    static constexpr int LDS = 2 * STAGE;       // 118 784 | 155 648
};

// UP4: the instantiation that adds the compact x4 taps result (BfsrConvX3Args.up4; quad-major output, no residuals) -- its own kernel so that the
// default one does not carry its address arithmetic (as a run-time branch it took the default kernel from 137 to 150 registers and 13 to 37 spilled SGPRs)
template <int XM, bool UP4 = false>
__global__ __launch_bounds__((NW + NLW) * 64, 1) void conv3x3_h2x_kernel(BfsrConvX3Args p, int tiles_x, int tiles_y, int groups, int nitems)
{
    constexpr int X_WPL = XGeo<XM>::WPL, X_W = XGeo<XM>::W, X_STAGE = XGeo<XM>::STAGE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int G = gridDim.x;
    const int slot = (int)bfsr::xcd_order(blockIdx.x, (unsigned)G);
    if (slot >= nitems) return;
    const int H = p.H, W = p.W;
    const unsigned HW16 = (unsigned)(H * W) * 16u;                       // bytes of one (octet, plane) image
    const int nchunk = p.Cin >> 4;

    auto decode = [&](int it) {
        Item r;
        r.cg = it % groups; int t = it / groups;
        const int ty = t % tiles_y; t /= tiles_y;
        r.x0 = (t % tiles_x) * 32; r.y0 = ty * TH; r.b = t / tiles_x;
        return r;
    };

    if (wave >= NW) {
        // ---- loader waves: LDS-DMA only (see h2s_loader_wave on why a wave must not mix load kinds)
        const int ld = wave - NW;                                        // = sub-image (plane ld>>1, k half ld&1) of every position group
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w), 0,
                                                                              (unsigned)((long long)groups * nchunk * X_W), 0x00020000);
        __amdgpu_buffer_rsrc_t rs_in;
        unsigned vg[NG];
        int cg_ = 0;
        auto lsetup = [&](const Item& it) {
            const unsigned short* xb = p.x + (long long)it.b * p.x_bs;
            rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xb), 0, (unsigned)(p.Cin >> 3) * 2u * HW16, 0x00020000);
            cg_ = it.cg;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int pos = g * 64 + lane;
                const int r = pos / PW, c = pos - r * PW;
                const int gy = it.y0 + r - 1, gx = it.x0 + c - 1;
                const bool ok = pos < NPOS && gy >= 0 && gy < H && gx >= 0 && gx < W;
                vg[g] = ok ? (unsigned)(gy * W + gx) * 16u : OOB;        // out of range -> the DMA writes zeros (= the padding)
            }
        };
        auto lstage = [&](int k, int buf) {
            if (BFSR_H2X_ABL & 4) return;
            unsigned char* base = smem + buf * X_STAGE;
            const unsigned soff = (unsigned)((2 * k + (ld & 1)) * 2 + (ld >> 1)) * HW16;      // octet 2k + k half, plane ld>>1
#pragma unroll
            for (int g = 0; g < NG; ++g)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(base + ld * SUB + g * 1024), 16, vg[g], soff, 0, 0);
            const unsigned wsoff = (unsigned)(cg_ * nchunk + k) * (unsigned)X_W;
#pragma unroll
            for (int j = 0; j < (X_W / 1024 + NLW - 1) / NLW; ++j) {
                const int piece = ld + j * NLW;
                if (piece < X_W / 1024)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(base + X_IN + piece * 1024), 16,
                                                             (unsigned)lane * 16u + (unsigned)piece * 1024u, wsoff, 0, 0);
            }
        };
        int it = slot;
        lsetup(decode(it));
        lstage(0, 0);
        int buf_ = 0;
        while (true) {
            const int nxt = it + G;
            const bool has_next = nxt < nitems;
            for (int k = 0; k < nchunk; ++k) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (k + 1 < nchunk) lstage(k + 1, buf_ ^ 1);
                else if (has_next) { lsetup(decode(nxt)); lstage(0, buf_ ^ 1); }
                buf_ ^= 1;
            }
            if (!has_next) return;
            it = nxt;
        }
    }

    // ---- compute waves: wave w owns rows 2w, 2w+1.  A step = one tap (dx, dy): 2 input rows x 2 planes + the tap's 2 weight planes
    // -> 6 MFMAs; the fragments of step t+1 are read while the MFMAs of step t run (register double buffer: 48 registers -- a
    // tap-COLUMN step as in conv3x3_h2s_kernel needs 112 and spills beside the 32 accumulators).
    half8 bq[2][2][2], aq[2][2][XM];                                     // [buffer][plane][row] | [buffer][plane][m tile]
    auto load_step = [&](auto buf_, int st, int t) {
        constexpr int BUF = decltype(buf_)::value;
        if (BFSR_H2X_ABL & 1) return;
        const int dx = t / 3, dy = t - 3 * dx;
        const unsigned char* sIn = smem + st * X_STAGE;
        const unsigned char* inB = sIn + (lhi * NPOSP + (2 * wave + dy) * PW + l31 + dx) * 16;
        const unsigned char* wA = sIn + X_IN + lane * 16 + t * (1024 * XM);                              // tap = dx*3 + dy = t
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int r = 0; r < 2; ++r) bq[BUF][pl][r] = *reinterpret_cast<const half8*>(inB + pl * 2 * SUB + r * PW * 16);
#pragma unroll
            for (int m = 0; m < XM; ++m) aq[BUF][pl][m] = *reinterpret_cast<const half8*>(wA + pl * X_WPL + m * 1024);
        }
    };
    f32x16 acc[XM][2];
    if (BFSR_H2X_ABL & 1) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) bq[b][pl][r][i] = (_Float16)(0.001f * (lane + i + r));
#pragma unroll
                for (int m = 0; m < XM; ++m)
#pragma unroll
                    for (int i = 0; i < 8; ++i) aq[b][pl][m][i] = (_Float16)(0.002f * (lane + i + m));
            }
    }
    auto mfma_step = [&](auto buf_) {
        constexpr int BUF = decltype(buf_)::value;
        if (BFSR_H2X_ABL & 2) {                                          // keep the fragment reads alive: one VALU use per fragment
#pragma unroll
            for (int m = 0; m < XM; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[m][j][0] += (float)aq[BUF][0][m][0] + (float)aq[BUF][1][m][0] + (float)bq[BUF][0][j][0] + (float)bq[BUF][1][j][0];
            return;
        }
#pragma unroll
        for (int m = 0; m < XM; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // smallest terms first: w_lo*x_hi, w_hi*x_lo, w_hi*x_hi
                acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[BUF][1][m], bq[BUF][0][j], acc[m][j], 0, 0, 0);
                acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[BUF][0][m], bq[BUF][1][j], acc[m][j], 0, 0, 0);
                acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[BUF][0][m], bq[BUF][0][j], acc[m][j], 0, 0, 0);
            }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    const float slope = p.act == BFSR_ACT_NONE ? 1.f : (p.act == BFSR_ACT_RELU ? 0.f : p.slope);
    const float4* __restrict__ epi = reinterpret_cast<const float4*>(p.epi);
    const long long HW = (long long)H * W;
    int buf = 0;                                                         // LDS stage of the next chunk
    float xamax = 0.f;                                                   // range guard: max |value| this thread hands to the fp16 split (h2 output)
    for (int it = slot; it < nitems; it += G) {
        const Item cur = decode(it);
        float4 pmm[XM];
        {
            int ln = lane;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(ln));                                 // per-lane address arithmetic stays inside the tile loop
#endif
#pragma unroll
            for (int m = 0; m < XM; ++m) {
                const int idx = (cur.cg * XM + m) * 64 + ln;
                pmm[m] = (ln & 1) ? make_float4(1.f, 0.f, 0.f, 0.f) : make_float4(0.f, 0.f, 1.f, 0.f);
                if (epi && (idx >> 1) < p.Cout) pmm[m] = epi[idx];
            }
        }
#pragma unroll
        for (int m = 0; m < XM; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[m][0][r] = 0.f; acc[m][1][r] = 0.f; }
        // One barrier per chunk, passed EARLY: chunk k+1's barrier sits before the last tap of chunk k (whose fragments are already in
        // registers), so the first fragments of chunk k+1 are in flight under that tap's MFMAs and the loaders may refill stage `buf`
        // one tap earlier.  Nine taps per chunk flip the fragment-buffer parity from chunk to chunk: chunks are processed in pairs.
        auto chunk_body = [&](auto p_, auto q_, bool last) {               // p_: buffer holding tap 0's fragments (already loaded)
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
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // tap 8's fragments have left stage `buf`
                __builtin_amdgcn_s_barrier();                            // chunk k+1 has landed in stage buf^1; stage buf is free again
                load_step(q_, buf ^ 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(p_);
            __builtin_amdgcn_sched_barrier(0);
            buf ^= 1;
        };
        __builtin_amdgcn_s_barrier();                                    // the item's first chunk has landed in stage `buf`
        load_step(I0(), buf, 0);
        int k = 0;
        for (; k + 2 <= nchunk; k += 2) {                                // pairs of chunks: the parity is static inside a pair
            chunk_body(I0(), I1(), false);
            chunk_body(I1(), I0(), k + 2 == nchunk);
        }
        if (k < nchunk) chunk_body(I0(), I1(), true);

        // ---- epilogue (the loaders are already staging the next item): as in conv3x3_h2s_kernel, plus the weight scale; one M tile
        // at a time (64 results per lane and the residual operands of both tiles at once would spill)
        if (BFSR_H2X_ABL & 8) {                                          // keep the accumulators alive
            if (acc[0][0][0] + acc[0][1][5] == 1234.5f) reinterpret_cast<float*>(p.y)[lane] = acc[0][0][1];
            continue;
        }
#pragma unroll
        for (int m = 0; m < XM; ++m) {
        const float4 pm = pmm[m];
        const bool plain = (lane & 1) ? pm.x == 1.f : (pm.y == 0.f && pm.z == 1.f && pm.w == 0.f);
        const bool bias_only = __all(plain);
        const bool fast = bias_only && slope >= 0.f && slope <= 1.f;
        auto fetch = [&](float val, int src_lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane * 4, __float_as_int(val))); };
        int lh = lhi, lx = l31;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(lh), "+v"(lx));
#endif
        const int gx = cur.x0 + lx;
        // every global access of the epilogue is a buffer instruction whose VGPR offset is out of range for pixels outside the image (and
        // whose descriptor ends at Cout channels): no `if (inside)` branch per access -- hipcc turned those into ~60 exec-masked blocks per
        // item with 64-bit address arithmetic each, and all eight compute waves sit in this epilogue at once
        const int oct0 = (cur.cg * XM + m) * 4;                          // first of this item's four channel octets (+ q*2 + lh)
        unsigned vo16[2], vo4[2];                                        // per row j: byte offset of (half-wave octet lh, pixel) in 16-byte / 4-byte units
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gy = cur.y0 + 2 * wave + j;
            const bool ok = gy < H && gx < W;
            vo16[j] = ok ? (unsigned)(((long long)lh * 2 * HW + (long long)gy * W + gx) * 16) : OOB;
            vo4[j] = ok ? (unsigned)(((long long)lh * 8 * HW + (long long)gy * W + gx) * 4) : OOB;
        }
        const unsigned h2_bytes = (unsigned)((long long)(p.Cout >> 3) * 2 * HW * 16);
        half8 rh[2][2], rl[2][2];
        auto load_res = [&](const unsigned short* res, long long bs) {
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(res + (long long)cur.b * bs), 0, h2_bytes, 0x00020000);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const unsigned so = (unsigned)((oct0 + q * 2) * 2) * (unsigned)(HW * 16);
                    rh[j][q] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rr, vo16[j], so, 0));
                    rl[j][q] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rr, vo16[j], so + (unsigned)(HW * 16), 0));
                }
        };
        if (p.res1) load_res(p.res1, p.res1_bs);                         // lands under the swaps / parameter exchange / activation
        // the compact x4 taps result (`up4`, added just before the stores below): fetched HERE into the residual registers (the launcher refuses
        // residuals together with up4: the key conv of the x4 level has none), so that it lands under the swaps / activation as well -- fetched at its point of use it cost the
        // 64 -> 1024 conv of config 4 ~4.5 ms per launch (28.5 -> 33 ms, profiles/r06q / r06u)
        auto load_up4 = [&]() {
            const int hs = H >> 2, wsrc = W >> 2;
            const unsigned cq = (unsigned)(hs * wsrc) * 144u;            // bytes of one channel quad's compact image [H/4][9][W/4][4]
            const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.up4) + (long long)cur.b * p.up4_bs, 0,
                                                                                (unsigned)(p.Cout >> 2) * cq, 0x00020000);
            const int pxc = gx & 3, cxc = pxc == 0 ? 0 : (pxc == 3 ? 2 : 1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int gy = cur.y0 + 2 * wave + j;
                const int pyc = gy & 3, cyc = pyc == 0 ? 0 : (pyc == 3 ? 2 : 1);
                const bool ok = gy < H && gx < W;
                const unsigned vu = ok ? (unsigned)lh * 2u * cq + (unsigned)(((gy >> 2) * 9 + cyc * 3 + cxc) * wsrc + (gx >> 2)) * 16u : OOB;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const unsigned so = (unsigned)((oct0 + q * 2) * 2) * cq;
                    rh[j][q] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(ru, vu, so, 0));
                    rl[j][q] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(ru, vu, so + cq, 0));
                }
            }
        };
        if constexpr (UP4) load_up4();
        float o[2][2][8];
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");       // MFMA result -> VALU read inside the asm below: 20 wait states (>= 19 of a 16-pass XDL op), self-sufficient
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float lo = acc[m][j][8 * q + i] * p.acc_scale, hi = acc[m][j][8 * q + 4 + i] * p.acc_scale;      // VALU first: hipcc pads the MFMA -> VALU hazard itself
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
                    o[j][q][i] = lo;
                    o[j][q][4 + i] = hi;
                }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (fast) {
                float e0[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) e0[i] = fetch(pm.x, ((q * 2 + lh) * 8 + i) * 2);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float u = o[j][q][i] + e0[i];
                        o[j][q][i] = fmaxf(u, u * slope);                // = u > 0 ? u : u*slope for 0 <= slope <= 1
                    }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int src = ((q * 2 + lh) * 8 + i) * 2;         // lane holding this channel's first float4
                    const float e0 = fetch(pm.x, src), e1 = fetch(pm.y, src), e2 = fetch(pm.z, src), e3 = fetch(pm.w, src), e4 = fetch(pm.x, src + 1);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float u = o[j][q][i] + e0;
                        u = (u + e1) * e2 + e3;
                        u = u > 0.f ? u : u * slope;
                        o[j][q][i] = u * e4;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (p.res1) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[j][q][i] = p.alpha1 * o[j][q][i] + ((float)rh[j][q][i] + (float)rl[j][q][i]);
        }
        if (p.res2) {
            load_res(p.res2, p.res2_bs);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[j][q][i] = p.alpha2 * o[j][q][i] + ((float)rh[j][q][i] + (float)rl[j][q][i]);
        }
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        if (p.y_fmt == 1) {
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned short*>(p.y) + (long long)cur.b * p.y_bs, 0, h2_bytes, 0x00020000);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    half8 h8, l8;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { _Float16 h, l; split2(o[j][q][i], h, l); h8[i] = h; l8[i] = l; xamax = fmaxf(xamax, fabsf(o[j][q][i])); }
                    const unsigned so = (unsigned)((oct0 + q * 2) * 2) * (unsigned)(HW * 16);
                    bfsr::store_b128(ry, __builtin_bit_cast(u32x4_, h8), vo16[j], so);
                    bfsr::store_b128(ry, __builtin_bit_cast(u32x4_, l8), vo16[j], so + (unsigned)(HW * 16));
                }
        } else {
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(p.y) + (long long)cur.b * p.y_bs, 0,
                                                                                (unsigned)((long long)p.Cout * HW * 4), 0x00020000);
            if (p.y_fmt == 2) {                                          // fp32 quad-major [Cout/4][H][W][4]: the octet = two 16-byte stores
                if constexpr (UP4) {
                    // + the COMPACT result of bfsr_conv2d_up4_h2t (y_fmt 3): per source pixel (y/4, x/4) and channel quad the nine phase-class values
                    // [Cout/4][H/4][9][W/4][4] (ABI 8: class rows, so that a wave's loads are whole cache lines) -- the x4 level's taps share, added here instead of
                    // being read back as a full-resolution pre_add
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const float4 r0 = __builtin_bit_cast(float4, rh[j][q]), r1 = __builtin_bit_cast(float4, rl[j][q]);
                            o[j][q][0] += r0.x; o[j][q][1] += r0.y; o[j][q][2] += r0.z; o[j][q][3] += r0.w;
                            o[j][q][4] += r1.x; o[j][q][5] += r1.y; o[j][q][6] += r1.z; o[j][q][7] += r1.w;
                        }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const unsigned so = (unsigned)((oct0 + q * 2) * 2) * (unsigned)(HW * 16);
                        bfsr::store_b128_stream(ry, __builtin_bit_cast(u32x4_, make_float4(o[j][q][0], o[j][q][1], o[j][q][2], o[j][q][3])), vo16[j], so);
                        bfsr::store_b128_stream(ry, __builtin_bit_cast(u32x4_, make_float4(o[j][q][4], o[j][q][5], o[j][q][6], o[j][q][7])), vo16[j], so + (unsigned)(HW * 16));
                    }
            } else {                                                     // fp32 NCHW: channels >= Cout fall beyond the descriptor
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[j][q][i]), ry, vo4[j], (unsigned)(((oct0 + q * 2) * 8 + i) * HW * 4), 0);
            }
        }
        }
    }
    if (p.flag && __any((int)!(xamax < 65504.f))) { if (lane == 0) atomicOr(p.flag, 1u); }
}

// ---- fp32 NCHW view <-> h2 tensor (the two ends of the fp16-stored region: conv_first's output, the trunk output) -------------
__global__ void h2_pack_kernel(const float* __restrict__ x, long long x_bs, unsigned short* __restrict__ y, long long y_bs,
                               int C, long long HW, long long total, unsigned* flag)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float amax = 0.f;                                                    // range guard of the fp16 split
    const int C8 = C >> 3;
    const long long pix = i % HW; const long long t = i / HW;
    const int oct = (int)(t % C8); const int b = (int)(t / C8);
    const float* xb = x + (long long)b * x_bs + (long long)oct * 8 * HW + pix;
    half8 h8, l8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float v = xb[(long long)j * HW]; _Float16 h, l; split2(v, h, l); h8[j] = h; l8[j] = l; amax = fmaxf(amax, fabsf(v)); }
    unsigned short* yb = y + (long long)b * y_bs + ((long long)oct * 2 * HW + pix) * 8;
    *reinterpret_cast<half8*>(yb) = h8;
    *reinterpret_cast<half8*>(yb + HW * 8) = l8;
    if (flag && !(amax < 65504.f)) atomicOr(flag, 1u);
}

// the same with fewer source channels than the h2 tensor has (the K padding of a conv's input): channel c >= Cs is zero, c < Cs is 1 * x + 0 (the copy
// bfsr_axpb_clamp made into the zero-initialised staging tensor this replaces: the same bits, signed zeros included)
__global__ void h2_pack_pad_kernel(const float* __restrict__ x, long long x_bs, unsigned short* __restrict__ y, long long y_bs,
                                   int Cs, int C, long long HW, long long total, unsigned* flag)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float amax = 0.f;
    const int C8 = C >> 3;
    const long long pix = i % HW; const long long t = i / HW;
    const int oct = (int)(t % C8); const int b = (int)(t / C8);
    const float* xb = x + (long long)b * x_bs + (long long)oct * 8 * HW + pix;
    half8 h8, l8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = 0.f;
        if (oct * 8 + j < Cs) { v = 1.0f * xb[(long long)j * HW] + 0.0f; }
        _Float16 h, l; split2(v, h, l); h8[j] = h; l8[j] = l; amax = fmaxf(amax, fabsf(v));
    }
    unsigned short* yb = y + (long long)b * y_bs + ((long long)oct * 2 * HW + pix) * 8;
    *reinterpret_cast<half8*>(yb) = h8;
    *reinterpret_cast<half8*>(yb + HW * 8) = l8;
    if (flag && !(amax < 65504.f)) atomicOr(flag, 1u);
}

__global__ void h2_unpack_kernel(const unsigned short* __restrict__ x, long long x_bs, float* __restrict__ y, long long y_bs,
                                 int C, long long HW, long long total)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int C8 = C >> 3;
    const long long pix = i % HW; const long long t = i / HW;
    const int oct = (int)(t % C8); const int b = (int)(t / C8);
    const unsigned short* xb = x + (long long)b * x_bs + ((long long)oct * 2 * HW + pix) * 8;
    const half8 h = *reinterpret_cast<const half8*>(xb);
    const half8 l = *reinterpret_cast<const half8*>(xb + HW * 8);
    float* yb = y + (long long)b * y_bs + (long long)oct * 8 * HW + pix;
#pragma unroll
    for (int j = 0; j < 8; ++j) yb[(long long)j * HW] = (float)h[j] + (float)l[j];
}

inline unsigned short f32_to_f16_bits(float v)
{
    const _Float16 h = (_Float16)v;              // round to nearest even, like the device conversion
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}

}  // namespace

// 64-cout workgroup tiles (mtile 2): built in round 2 with an all-tiles-at-once epilogue that spilled (1.05 ms against 0.61 ms for conv5 at
// 128 x 128x128, tools/exp/h2s_bench.py); rebuilt in round 6 with the per-M-tile epilogue above.  The weight image depends on the tile width,
// so the caller chooses at pack time and passes the same mtile in BfsrConvX3Args.mtile (0 = 1).
extern "C" long long bfsr_conv_packed_size_h2s_mt(int Cout, int Cin, int mtile)
{
    if (Cout <= 0 || Cin <= 0 || (Cin & 31) || (mtile != 1 && mtile != 2)) return -1;
    const int MW = mtile * 32;
    return (long long)((Cout + MW - 1) / MW) * (Cin / 16) * 9 * 2 * MW * 8;      // fp16 elements
}

extern "C" int bfsr_pack_conv_weight_h2s_mt(const float* w, int Cout, int Cin, int mtile, unsigned short* packed)
{
    // w [Cout][Cin][3][3] fp32 -> fp16 [cout group of MW = 32 * mtile][16-channel chunk][tap = dx*3 + dy][k half][MW][8],
    // zero padded (tap-column-major: the kernel consumes one tap column dx per pipeline step)
    if (!w || !packed || Cout <= 0 || Cin <= 0 || (Cin & 31) || (mtile != 1 && mtile != 2)) return -1;
    const int nchunk = Cin / 16, MW = mtile * 32;
    const long long n = bfsr_conv_packed_size_h2s_mt(Cout, Cin, mtile);
    for (long long i = 0; i < n; ++i) packed[i] = 0;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx)
                    packed[(((((long long)(co / MW) * nchunk + ci / 16) * 9 + dx * 3 + dy) * 2 + (ci % 16) / 8) * MW + co % MW) * 8 + ci % 8] =
                        f32_to_f16_bits(w[((long long)co * Cin + ci) * 9 + dy * 3 + dx]);
    return 0;
}

extern "C" long long bfsr_conv_packed_size_h2s(int Cout, int Cin) { return bfsr_conv_packed_size_h2s_mt(Cout, Cin, 1); }
extern "C" int bfsr_pack_conv_weight_h2s(const float* w, int Cout, int Cin, unsigned short* packed) { return bfsr_pack_conv_weight_h2s_mt(w, Cout, Cin, 1, packed); }

namespace {
template <int MT>
int launch_h2s(const BfsrConvX3Args* a, hipStream_t st, bool allow_res)
{
    const int tiles_x = (a->W + 31) / 32, tiles_y = (a->H + TH - 1) / TH;
    const int groups = (a->Cout + MT * 32 - 1) / (MT * 32);
    const long long nitems = (long long)tiles_x * tiles_y * groups * a->B;
    if (nitems > 0x7fffffffLL) return -1;
    if (bfsr_conv_packed_size_h2s_mt(a->Cout, a->Cin, MT) * 2 >= (1LL << 32)) return -1;
    int cus = bfsr::cu_count();                             // cached per device; no silent default
    if (cus <= 0) return -1;
    if (a->tune > 0) cus = a->tune;
    const long long grid = nitems < cus ? nitems : cus;                  // one persistent workgroup per CU
    // (round 3's ping-pong compute groups and register-staged weight loader were parity-tested and measured within box noise of this
    // kernel: tools/exp/kernels/conv_h2s_r3.hip)
    // resident weights: one output-channel group, and room for at least four input stages behind the weight tensor
    const int wb = (a->Cin / 16) * SGeo<MT>::W_BYTES;
    int ns = (160 * 1024 - wb) / IN_BYTES;
    ns = ns > 6 ? 6 : ns;
    if (allow_res && groups == 1 && ns >= 4) {
        const int lds = wb + ns * IN_BYTES;
        static std::atomic<unsigned long long> lds_res{0};
        if (bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&conv3x3_h2s_kernel<MT, 1>), 160 * 1024, lds_res) != 0) return -1;
        hipLaunchKernelGGL((conv3x3_h2s_kernel<MT, 1>), dim3((unsigned)grid), dim3((NW + NLW) * 64), lds, st, *a, tiles_x, tiles_y, groups, (int)nitems, ns);
        return (int)hipGetLastError();
    }
    static std::atomic<unsigned long long> lds_done{0};
    if (bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&conv3x3_h2s_kernel<MT, 0>), SGeo<MT>::LDS_TOTAL, lds_done) != 0) return -1;
    hipLaunchKernelGGL((conv3x3_h2s_kernel<MT, 0>), dim3((unsigned)grid), dim3((NW + NLW) * 64), SGeo<MT>::LDS_TOTAL, st, *a, tiles_x, tiles_y, groups, (int)nitems, 0);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int bfsr_conv3x3_h2s(const BfsrConvX3Args* a, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || !a->x || !a->w || !a->y) return -1;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->Cin <= 0 || (a->Cin & 31) || a->Cout <= 0) return -1;
    if (a->y_fmt < 0 || a->y_fmt > 2) return -1;
    const int mt_ = a->mtile & 0xff;
    const bool allow_res = !(a->mtile & 0x100);                          // bit 8 of mtile: keep the weights streamed (A/B switch, BFSR_H2S_RES=0)
    if (mt_ < 0 || mt_ > 2 || (a->mtile & ~0x1ff)) return -1;
    if ((a->y_fmt != 0 || a->res1 || a->res2) && (a->Cout & 7)) return -1;
    if ((long long)(a->Cin / 8) * 2 * a->H * a->W * 16 >= (1LL << 31)) return -1;      // 32-bit byte offsets inside one batch item
    if ((long long)((a->Cout + 7) / 8) * 2 * a->H * a->W * 8 >= (1LL << 31)) return -1;  // 32-bit element offsets in the epilogue
    if ((reinterpret_cast<unsigned long long>(a->x) & 15) || (a->x_bs & 7)) return -1;
    if (a->y_fmt != 0 && ((reinterpret_cast<unsigned long long>(a->y) & 15) || (a->y_bs & 7))) return -1;
    if (a->res1 && ((reinterpret_cast<unsigned long long>(a->res1) & 15) || (a->res1_bs & 7))) return -1;
    if (a->res2 && ((reinterpret_cast<unsigned long long>(a->res2) & 15) || (a->res2_bs & 7))) return -1;
    return mt_ == 2 ? launch_h2s<2>(a, st, allow_res) : launch_h2s<1>(a, st, allow_res);
}

extern "C" long long bfsr_conv_packed_size_h2x(int Cout, int Cin, int mtile)
{
    if (Cout <= 0 || Cin <= 0 || (Cin & 15) || (mtile != 1 && mtile != 2)) return -1;
    const int MW = 32 * mtile;
    return (long long)((Cout + MW - 1) / MW) * (Cin / 16) * 2 * 9 * mtile * 2 * 32 * 8;       // fp16 elements
}

extern "C" int bfsr_pack_conv_weight_h2x(const float* w, int Cout, int Cin, int mtile, float scale, unsigned short* packed)
{
    // w [Cout][Cin][3][3] fp32 -> fp16 [cout group of 32*mtile][16-channel chunk][plane hi,lo][tap = dx*3 + dy][m tile][k half][32][8] of
    // w*scale, zero padded; scale = a power of two chosen by the caller (bfsr_amd/ops.py: largest |w|*scale in [2^9, 2^10))
    if (!w || !packed || Cout <= 0 || Cin <= 0 || (Cin & 15) || !(scale > 0.f) || (mtile != 1 && mtile != 2)) return -1;
    const int nchunk = Cin / 16, MW = 32 * mtile;
    const long long n = bfsr_conv_packed_size_h2x(Cout, Cin, mtile);
    for (long long i = 0; i < n; ++i) packed[i] = 0;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx) {
                    const float v = w[((long long)co * Cin + ci) * 9 + dy * 3 + dx] * scale;
                    const _Float16 h = (_Float16)v;
                    const _Float16 l = (_Float16)(v - (float)h);
                    const unsigned short hb = f32_to_f16_bits((float)h), lb = f32_to_f16_bits((float)l);
                    const long long base = ((long long)(co / MW) * nchunk + ci / 16) * 2;
                    const long long in = ((((long long)(dx * 3 + dy) * mtile + (co % MW) / 32) * 2 + (ci % 16) / 8) * 32 + co % 32) * 8 + ci % 8;
                    packed[(base + 0) * (9 * mtile * 2 * 32 * 8) + in] = hb;
                    packed[(base + 1) * (9 * mtile * 2 * 32 * 8) + in] = lb;
                }
    return 0;
}

extern "C" int bfsr_conv3x3_h2x(const BfsrConvX3Args* a, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || !a->x || !a->w || !a->y) return -1;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->Cin <= 0 || (a->Cin & 15) || a->Cout <= 0) return -1;
    if (a->y_fmt != 0 && a->y_fmt != 1 && a->y_fmt != 2) return -1;    // fp32 NCHW | h2 | fp32 quad-major [Cout/4][H][W][4]
    if (!(a->acc_scale > 0.f)) return -1;
    if ((a->y_fmt != 0 || a->res1 || a->res2) && (a->Cout & 7)) return -1;
    if (a->y_fmt == 2 && ((reinterpret_cast<unsigned long long>(a->y) & 15) || (a->y_bs & 3))) return -1;
    if (a->up4 && (a->y_fmt != 2 || a->res1 || a->res2 || (a->H & 3) || (a->W & 3) || (reinterpret_cast<unsigned long long>(a->up4) & 15) || (a->up4_bs & 3) ||
                   (long long)(a->Cout / 4) * (a->H / 4) * (a->W / 4) * 144 >= (1LL << 31))) return -1;
    if ((long long)(a->Cin / 8) * 2 * a->H * a->W * 16 >= (1LL << 31)) return -1;      // 32-bit byte offsets inside one batch item
    if ((long long)((a->Cout + 7) / 8) * 2 * a->H * a->W * 8 >= (1LL << 31)) return -1;  // 32-bit element offsets in the epilogue
    if ((reinterpret_cast<unsigned long long>(a->x) & 15) || (a->x_bs & 7)) return -1;
    if (a->y_fmt == 1 && ((reinterpret_cast<unsigned long long>(a->y) & 15) || (a->y_bs & 7))) return -1;
    if (a->res1 && ((reinterpret_cast<unsigned long long>(a->res1) & 15) || (a->res1_bs & 7))) return -1;
    if (a->res2 && ((reinterpret_cast<unsigned long long>(a->res2) & 15) || (a->res2_bs & 7))) return -1;
    const int tiles_x = (a->W + 31) / 32, tiles_y = (a->H + TH - 1) / TH;
    const int mt = 1;                                                    // 64-cout workgroup tiles were measured 5-8 % slower and are gone
    if (a->mtile != 0 && a->mtile != 1) return -1;
    const int groups = (a->Cout + 32 * mt - 1) / (32 * mt);
    const long long nitems = (long long)tiles_x * tiles_y * groups * a->B;
    if (nitems > 0x7fffffffLL) return -1;
    if (bfsr_conv_packed_size_h2x(a->Cout, a->Cin, mt) * 2 >= (1LL << 32)) return -1;
    int cus = bfsr::cu_count();
    if (cus <= 0) return -1;
    if (a->tune > 0) cus = a->tune;
    const long long grid = nitems < cus ? nitems : cus;                  // one persistent workgroup per CU
    if (a->up4) {
        static std::atomic<unsigned long long> lds_up4{0};
        if (bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&conv3x3_h2x_kernel<1, true>), XGeo<1>::LDS, lds_up4) != 0) return -1;
        hipLaunchKernelGGL((conv3x3_h2x_kernel<1, true>), dim3((unsigned)grid), dim3((NW + NLW) * 64), XGeo<1>::LDS, st, *a, tiles_x, tiles_y, groups, (int)nitems);
        return (int)hipGetLastError();
    }
    static std::atomic<unsigned long long> lds_done{0};
    if (bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&conv3x3_h2x_kernel<1>), XGeo<1>::LDS, lds_done) != 0) return -1;
    hipLaunchKernelGGL((conv3x3_h2x_kernel<1>), dim3((unsigned)grid), dim3((NW + NLW) * 64), XGeo<1>::LDS, st, *a, tiles_x, tiles_y, groups, (int)nitems);
    return (int)hipGetLastError();
}

extern "C" int bfsr_h2_pack(const float* x, long long x_bs, unsigned short* y, long long y_bs, int B, int C, int H, int W, unsigned* flag, void* stream)
{
    if (!x || !y || B <= 0 || C <= 0 || (C & 7) || H <= 0 || W <= 0) return -1;
    if ((reinterpret_cast<unsigned long long>(y) & 15) || (y_bs & 7)) return -1;
    const long long HW = (long long)H * W, total = (long long)B * (C / 8) * HW;
    hipLaunchKernelGGL(h2_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, y, y_bs, C, HW, total, flag);
    return (int)hipGetLastError();
}

extern "C" int bfsr_h2_pack_pad(const float* x, long long x_bs, unsigned short* y, long long y_bs, int B, int Cs, int C, int H, int W, unsigned* flag, void* stream)
{
    if (!x || !y || B <= 0 || C <= 0 || (C & 7) || Cs <= 0 || Cs > C || H <= 0 || W <= 0) return -1;
    if ((reinterpret_cast<unsigned long long>(y) & 15) || (y_bs & 7)) return -1;
    const long long HW = (long long)H * W, total = (long long)B * (C / 8) * HW;
    hipLaunchKernelGGL(h2_pack_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, y, y_bs, Cs, C, HW, total, flag);
    return (int)hipGetLastError();
}

extern "C" int bfsr_h2_unpack(const unsigned short* x, long long x_bs, float* y, long long y_bs, int B, int C, int H, int W, void* stream)
{
    if (!x || !y || B <= 0 || C <= 0 || (C & 7) || H <= 0 || W <= 0) return -1;
    if ((reinterpret_cast<unsigned long long>(x) & 15) || (x_bs & 7)) return -1;
    const long long HW = (long long)H * W, total = (long long)B * (C / 8) * HW;
    hipLaunchKernelGGL(h2_unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, y, y_bs, C, HW, total);
    return (int)hipGetLastError();
}
