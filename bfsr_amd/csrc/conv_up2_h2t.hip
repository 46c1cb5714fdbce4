// conv_up2_h2t.hip -- 3x3 'same' conv over a NEAREST-x2-UPSAMPLED tensor, evaluated at the source resolution, at fp32-class accuracy on
// the fp16 matrix pipe (two-term fp16 split, three products; conv_h2s.hip's arithmetic).  The hot instance: the 256 RRDB tap
// channels (4 tapped blocks x 64, LR resolution) entering the first conv of every level-1 coupling's conditioning network through
// F.interpolate(..., mode='nearest') + torch.cat (SRFlow-LP/code/models/modules/SRFlowNet_arch.py:122-137, RRDBNet_arch.py:105-109,
// FlowAffineCouplingsAblation.py:127-135): 256 -> 16 steps x 64 channels at 2x the LR size, twice per pass (fAffine and fFeatures).
//
// Parity decomposition: output pixel (2y+py, 2x+px) sees, through its 3x3 window on the upsampled image, only the 2x2 source pixels
// rows {y-1+py, y+py} x cols {x-1+px, x+px}; the weights of the window taps that fall on the same source pixel are summed at pack
// time.  So one source pixel row of 32 pixels, one 16-channel chunk and 32 output channels are 16 (parity, tap) weight blocks x 3
// products = 48 MFMAs instead of 4 x 9 x 3.
//
// Work decomposition (what the register-staged conv_up2_bf16x3_kernel lacked: it re-splits and re-stages the fp32 taps once per 32
// output channels AND per output parity, and holds 4 accumulator blocks): the taps arrive as an h2 tensor (split once), a workgroup
// item = source tile 16 x 32 x 32 output channels x ALL FOUR parities; wave w owns source rows 2w, 2w+1 = 8 accumulator blocks
// (128 registers, 2 waves per SIMD), reads per 16-channel chunk the 4 rows x 3 column shifts x 2 planes it needs ONCE (24 fragments)
// plus the 32 weight fragments, for 96 MFMAs: 0.58 LDS reads per MFMA (conv3x3_h2x_kernel: 0.78).  No loader waves -- with them the
// workgroup would be 12 waves and the register budget 168: every wave issues its share (9 of 72 one-KiB pieces) of the NEXT chunk by
// LDS-DMA right behind the chunk barrier, a whole chunk of MFMAs (~3 us) before it is needed; the only vector-memory wait in the K loop is
// `s_waitcnt vmcnt(0)` in front of that barrier (LDS-DMA and ordinary loads of one wave do not retire in issue order relative to each
// other, conv_h2s.hip, so no counted waits).  Two LDS stages of 40 960 (input: [2 planes][2 k halves][640 positions][8]) + 32 768
// (weights: [2 planes][16 steps][2 k halves][32][8]) bytes.  Persistent workgroups; item order = [8 tiles][4 cout groups] per XCD round.
// Output: fp32 quad-major [B][Cout/4][2h][2w][4] (what the coupling pair and the fFeatures head read), y = acc/scale + pre_add: every
// lane owns whole channel quads of its pixels in the MFMA result layout, so the epilogue is 16-byte loads and stores without a swap.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include "../../include/bfsr_hip.h"
#include "launch_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

namespace {

constexpr int NWV = 8;                          // waves per workgroup; all compute, all stage
constexpr int TH = 16, PW = 34, NPOS = (TH + 2) * PW, NG = 10, NPOSP = NG * 64;
constexpr int SUB = NPOSP * 16;                 // one (plane, k half) sub-image: 8 channels of every tile position
constexpr int X_IN = 4 * SUB;                   // 40 960
constexpr int NSTEP = 16;                       // (px, b, py, a) weight blocks per chunk
constexpr int W_PL = NSTEP * 1024;              // one weight plane of a chunk
constexpr int W_ST = 2 * W_PL;                  // 32 768
constexpr int STAGE = X_IN + W_ST;              // 73 728
constexpr int LDS_BYTES = 2 * STAGE;            // 147 456
constexpr int NXP = X_IN / 1024 / NWV;          // input pieces per wave and chunk: 5
constexpr int NWP = W_ST / 1024 / NWV;          // weight pieces per wave and chunk: 4
constexpr unsigned OOB = 0x80000000u;

struct Item { int cg, b, x0, y0; };

// ---- weight-step tables.  A chunk = 16 input channels at SOURCE resolution; per chunk a list of steps, each = one weight block [32 couts][16]
// applied to the source rows j + roff (tile rows 2w + j + roff) at column shift cs, accumulated into output parity p4 = py*2 + px.
//   type 0, taps chunk (the upsampled tensor): 16 steps s = (px*2 + b)*4 + py*2 + a: source tap (a, b) of parity (py, px): roff = py + a,
//           cs = px + b - 1 -- grouped by cs: -1 (4 steps), 0 (8), +1 (4).
//   type 1 + q, key chunk of space-to-depth plane q = qy*2 + qx (channels that live at OUTPUT resolution, plane q = their pixels (2y+qy, 2x+qx)):
//           output row 2y+py reads window row dy at output-resolution row 2y+py+dy-1 = plane-qy row y+ry; per axis and plane parity q the
//           (p, r) pairs are  q = 0: (0,0) (1,0) (1,+1);  q = 1: (0,-1) (0,0) (1,0),  window index d = 2r + q - p + 1.  9 steps s = ci*3 + ri
//           (ci: column pair, ri: row pair), grouped by cs = rx: qx = 0: 0 (6 steps), +1 (3); qx = 1: -1 (3), 0 (6).
struct StepInfo { int p4, roff, cs, dy, dx; };
constexpr int axis_p(int q, int i) { return q == 0 ? (i >= 1) : (i == 2); }
constexpr int axis_r(int q, int i) { return q == 0 ? (i == 2 ? 1 : 0) : (i == 0 ? -1 : 0); }
constexpr StepInfo step_info(int type, int s)
{
    if (type == 0) {
        const int px = s >> 3, b = (s >> 2) & 1, py = (s >> 1) & 1, a = s & 1;
        return StepInfo{py * 2 + px, py + a, px + b - 1, 0, 0};
    }
    const int q = type - 1, qy = q >> 1, qx = q & 1, ci = s / 3, ri = s % 3;
    const int px = axis_p(qx, ci), rx = axis_r(qx, ci), py = axis_p(qy, ri), ry = axis_r(qy, ri);
    return StepInfo{py * 2 + px, ry + 1, rx, 2 * ry + qy - py + 1, 2 * rx + qx - px + 1};
}
constexpr int nsteps(int type) { return type == 0 ? 16 : 9; }
constexpr int group_of(int type, int s)          // index of the column-shift group step s belongs to
{
    int g = 0;
    for (int i = 1; i <= s; ++i) g += step_info(type, i).cs != step_info(type, i - 1).cs;
    return g;
}
constexpr int next_group_start(int type, int s)  // first step of the group after s's (nsteps if none)
{
    int i = s + 1;
    while (i < nsteps(type) && step_info(type, i).cs == step_info(type, s).cs) ++i;
    return i;
}

#ifndef BFSR_H2T_ABL
#define BFSR_H2T_ABL 0                          // ablation builds only (tools/exp): bit 0 no fragment reads, 1 no MFMAs, 2 no DMA, 3 no epilogue
#endif

__global__ __launch_bounds__(NWV * 64, 1) void conv_up2_h2t_kernel(BfsrUp2H2Args p, int tiles_x, int tiles_y, int groups, int nitems)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int G = gridDim.x;
    const int slot = (int)bfsr::xcd_order(blockIdx.x, (unsigned)G);
    if (slot >= nitems) return;
    const int h = p.h, w = p.w_;
    const unsigned HW16 = (unsigned)(h * w) * 16u;                       // bytes of one (octet, plane) image of the source
    const int nct = (p.Cin - 4 * p.Ckey) >> 4;                           // taps chunks; then 4 planes x Ckey/16 key chunks
    const int nck = p.Ckey >> 4;
    const int nchunk = nct + 4 * nck;
    const int ntiles = p.B * tiles_y * tiles_x;

    // items in rounds of [8 source tiles][cout groups]: the 32 workgroups of an XCD share 8 input tiles and 4 weight sets at a time
    auto decode = [&](int it) {
        Item r;
        const int per = 8 * groups;
        int tg = it / per;
        const int tgl = (ntiles - 1) >> 3;
        tg = tg < tgl ? tg : tgl;
        const int rem = it - tg * per;
        const int nt = ntiles - 8 * tg < 8 ? ntiles - 8 * tg : 8;
        r.cg = rem / nt;
        int t = tg * 8 + (rem - r.cg * nt);
        r.x0 = (t % tiles_x) * 32; t /= tiles_x;
        r.y0 = (t % tiles_y) * TH; r.b = t / tiles_y;
        return r;
    };

    // ---- staging: wave w issues input pieces w, w+8, .. (piece i = sub-image i/10 (plane i/20, k half (i/10)&1), position group i%10)
    // and weight pieces w, w+8, ..
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w), 0,
                                                                          (unsigned)((long long)groups * nchunk * W_ST), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_in;
    unsigned vg[NXP];
    int ld_cg = 0;
    auto lsetup = [&](const Item& it) {
        const unsigned short* xb = p.x + (long long)it.b * p.x_bs;
        rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xb), 0, (unsigned)(p.Cin >> 3) * 2u * HW16, 0x00020000);
        ld_cg = it.cg;
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            const int g = (wave + NWV * i) % NG;
            const int pos = g * 64 + lane;
            const int r = pos / PW, c = pos - r * PW;
            const int gy = it.y0 + r - 1, gx = it.x0 + c - 1;
            const bool ok = pos < NPOS && gy >= 0 && gy < h && gx >= 0 && gx < w;
            vg[i] = ok ? (unsigned)(gy * w + gx) * 16u : OOB;            // out of range -> the DMA writes zeros (= the padding)
        }
    };
    // piece i (0..8: five input pieces, four weight pieces) of chunk k of the item lsetup() described, into stage stg; k < 0: nothing to stage
    auto lpiece = [&](int i, int k, int stg) {
        if ((BFSR_H2T_ABL & 4) || k < 0) return;
        unsigned char* base = smem + stg * STAGE;
        if (i < NXP) {
            const int piece = wave + NWV * i;
            const int si = piece / NG;                                   // sub-image: plane si>>1, k half si&1
            const unsigned soff = (unsigned)((2 * k + (si & 1)) * 2 + (si >> 1)) * HW16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(base + piece * 1024), 16, vg[i], soff, 0, 0);
        } else {
            const int piece = wave + NWV * (i - NXP);
            if (k >= nct && (piece & 15) >= 9) return;                   // key chunk: 9 steps per weight plane
            const unsigned wsoff = (unsigned)(ld_cg * nchunk + k) * (unsigned)W_ST;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(base + X_IN + piece * 1024), 16, (unsigned)lane * 16u + (unsigned)piece * 1024u, wsoff, 0, 0);
        }
    };
    auto lstage = [&](int k, int stg) {
#pragma unroll
        for (int i = 0; i < NXP + NWP; ++i) lpiece(i, k, stg);
    };

    // ---- fragments: xin[buffer][tile row 2w + r, r = 0..3][plane] for one column shift cs; wq[buffer][plane] for one weight step
    half8 xin[2][4][2], wq[2][2];
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    auto load_x = [&](auto b_, int stg, int cs) {
        constexpr int BUF = decltype(b_)::value;
        if (BFSR_H2T_ABL & 1) return;
        const unsigned char* base = smem + stg * STAGE + (lhi * NPOSP + (2 * wave) * PW + l31 + cs + 1) * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) xin[BUF][r][pl] = *reinterpret_cast<const half8*>(base + pl * 2 * SUB + r * PW * 16);
    };
    auto load_w = [&](auto b_, int stg, int s) {
        constexpr int BUF = decltype(b_)::value;
        if (BFSR_H2T_ABL & 1) return;
        const unsigned char* base = smem + stg * STAGE + X_IN + s * 1024 + lane * 16;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wq[BUF][pl] = *reinterpret_cast<const half8*>(base + pl * W_PL);
    };
    f32x16 acc[4][2];                                                    // [parity py*2 + px][row j]
    if (BFSR_H2T_ABL & 1) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { xin[b][r][0][i] = (_Float16)(0.001f * (lane + i + r)); xin[b][r][1][i] = (_Float16)(0.0001f * (lane + i)); }
                wq[b][0][i] = (_Float16)(0.002f * (lane + i)); wq[b][1][i] = (_Float16)(0.0002f * (lane + i));
            }
    }
    auto mfma_step = [&](auto t_, auto s_) {
        constexpr int TYPE = decltype(t_)::value, S = decltype(s_)::value;
        constexpr StepInfo si = step_info(TYPE, S);
        constexpr int XB = group_of(TYPE, S) & 1, WB = S & 1;
        if (BFSR_H2T_ABL & 2) return;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // smallest terms first: w_lo*x_hi, w_hi*x_lo, w_hi*x_hi
            acc[si.p4][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[WB][1], xin[XB][j + si.roff][0], acc[si.p4][j], 0, 0, 0);
            acc[si.p4][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[WB][0], xin[XB][j + si.roff][1], acc[si.p4][j], 0, 0, 0);
            acc[si.p4][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[WB][0], xin[XB][j + si.roff][0], acc[si.p4][j], 0, 0, 0);
        }
    };
    // one chunk: the column-shift groups alternate between the two input fragment buffers; the fragments of the next group / step are
    // read while the current MFMAs run.  The nine LDS-DMA pieces of the next chunk (lk; into the other stage) go out one per step behind
    // the step's MFMAs: issued in one burst behind the barrier they kept all eight waves off the matrix pipe at the same time.
    auto step = [&](auto t_, auto s_, int stg, int lk) {
        constexpr int TYPE = decltype(t_)::value, S = decltype(s_)::value, NS = nsteps(TYPE);
        if constexpr (S < NS) {
            constexpr int G = group_of(TYPE, S), NG_ = next_group_start(TYPE, S);
            if constexpr ((S == 0 || group_of(TYPE, S - 1) != G) && NG_ < NS)      // first step of a group: prefetch the next group's rows
                load_x(std::integral_constant<int, (G + 1) & 1>(), stg, step_info(TYPE, NG_).cs);
            if constexpr (S + 1 < NS) load_w(std::integral_constant<int, (S + 1) & 1>(), stg, S + 1);
            __builtin_amdgcn_sched_barrier(0);                           // (without the fences: 6.58 vs 6.67 ms; s_setprio 1 / 3 around the MFMAs: 6.77 -- noise)
            mfma_step(t_, s_);
            if constexpr (S < NXP + NWP) lpiece(S, lk, stg ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto chunk_body = [&](auto t_, int stg, int lk) {
        constexpr int TYPE = decltype(t_)::value;
        load_x(I0(), stg, step_info(TYPE, 0).cs);
        load_w(I0(), stg, 0);
#define ST_(N_) step(t_, std::integral_constant<int, N_>(), stg, lk);
        ST_(0) ST_(1) ST_(2) ST_(3) ST_(4) ST_(5) ST_(6) ST_(7) ST_(8) ST_(9) ST_(10) ST_(11) ST_(12) ST_(13) ST_(14) ST_(15)
#undef ST_
    };

    const int H2 = 2 * h, W2 = 2 * w;
    const unsigned Q16 = (unsigned)(H2 * W2) * 16u;                      // bytes of one output channel quad image
    int it = slot;
    lsetup(decode(it));
    lstage(0, 0);
    int stg = 0;                                                         // LDS stage of the chunk to compute next
    bool drained = false;                                                // this wave's pieces of that chunk are known to have landed
    for (; it < nitems; it += G) {
        const Item cur = decode(it);
        const int nxt = it + G;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[q][0][r] = 0.f; acc[q][1][r] = 0.f; }
        auto run_chunk = [&](auto t_, int k) {
            if (!drained) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            drained = false;
            __builtin_amdgcn_s_barrier();                                // chunk k is in stage stg; every wave is past its reads of stage stg^1
            int lk = k + 1;
            if (lk == nchunk) {
                lk = -1;
                if (nxt < nitems) { lsetup(decode(nxt)); lk = 0; }
            }
            __builtin_amdgcn_sched_barrier(0);
            chunk_body(t_, stg, lk);
            stg ^= 1;
        };
        // one loop per chunk type (a type dispatch inside one loop made hipcc spill 367 registers at the join)
#pragma unroll 1
        for (int k = 0; k < nct; ++k) run_chunk(I0(), k);
#pragma unroll 1
        for (int k = 0; k < nck; ++k) run_chunk(I1(), nct + k);
#pragma unroll 1
        for (int k = 0; k < nck; ++k) run_chunk(std::integral_constant<int, 2>(), nct + nck + k);
#pragma unroll 1
        for (int k = 0; k < nck; ++k) run_chunk(std::integral_constant<int, 3>(), nct + 2 * nck + k);
#pragma unroll 1
        for (int k = 0; k < nck; ++k) run_chunk(std::integral_constant<int, 4>(), nct + 3 * nck + k);
        // ---- epilogue: y = acc * acc_scale + pre_add, fp32 quad-major.  Result layout of the 32x32 MFMA: lane (l31 = pixel, lhi),
        // register r = channel (r&3) + 8*(r>>2) + 4*lhi of the 32 -> registers 4i..4i+3 are channel quad 2i + lhi: one 16-byte access.
        if (BFSR_H2T_ABL & 8) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) t += acc[q][0][q] + acc[q][1][q + 4];
            if (t == 1234.5f) p.y[lane] = t;
            continue;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the DMA pieces of the next item's first chunk: ordinary loads follow
        drained = true;
        // descriptors of this item's 8 channel quads only (a 1024-channel DIV2K-sized output is 2.8 GB per sample: beyond 32-bit offsets)
        const long long qoff = (long long)cur.cg * 8 * (Q16 >> 2);
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y + (long long)cur.b * p.y_bs + qoff, 0, 8u * Q16, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pre_add ? p.pre_add + (long long)cur.b * p.pre_add_bs + qoff : p.y), 0,
                                                                              p.pre_add ? 8u * Q16 : 0u, 0x00020000);
        int lh = lhi, lx = l31;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(lh), "+v"(lx));                           // per-lane address arithmetic stays inside the item loop
#endif
        const int sx = cur.x0 + lx;
        // four rounds (row j, row parity py) of 8 quads (2 column parities x 4 channel quads); the pre_add loads of round n+1 are issued
        // before round n's arithmetic and stores (the fragment registers are free here): two rounds of loads in flight instead of a
        // load -> wait -> store chain per round (0.7 of 4.8 ms per launch at config 2 before)
        unsigned vo[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int sy = cur.y0 + 2 * wave + (n >> 1);
            const bool ok = sy < h && sx < w;
            vo[n] = ok ? (unsigned)lh * Q16 + (unsigned)((2 * sy + (n & 1)) * W2 + 2 * sx) * 16u : OOB;     // pixel (2sy+py, 2sx); px adds 16 bytes
        }
        float4 pre[2][2][4];
        auto load_pre = [&](auto b_, int n) {
            constexpr int BUF = decltype(b_)::value;
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    pre[BUF][px][i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, vo[n] + 16u * px, (unsigned)(2 * i) * Q16, 0));
        };
        auto finish = [&](auto b_, auto n_) {
            constexpr int BUF = decltype(b_)::value, N = decltype(n_)::value;
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x16& a = acc[(N & 1) * 2 + px][N >> 1];
                    float4 o;
                    o.x = a[4 * i + 0] * p.acc_scale + pre[BUF][px][i].x;
                    o.y = a[4 * i + 1] * p.acc_scale + pre[BUF][px][i].y;
                    o.z = a[4 * i + 2] * p.acc_scale + pre[BUF][px][i].z;
                    o.w = a[4 * i + 3] * p.acc_scale + pre[BUF][px][i].w;
                    bfsr::store_b128_stream(rs_y, __builtin_bit_cast(u32x4, o), vo[N] + 16u * px, (unsigned)(2 * i) * Q16);
                }
        };
        // (in place, y already holding pre_add, one fire-and-forget buffer_atomic_add_f32 per element instead of load + add + store was
        // measured: 21.0 ms per launch instead of 4.8 -- fp32 atomics on HBM-resident lines run at a fraction of the store rate)
        load_pre(I0(), 0);
        load_pre(I1(), 1);
        __builtin_amdgcn_sched_barrier(0);
        finish(I0(), I0());
        __builtin_amdgcn_sched_barrier(0);
        load_pre(I0(), 2);
        __builtin_amdgcn_sched_barrier(0);
        finish(I1(), I1());
        __builtin_amdgcn_sched_barrier(0);
        load_pre(I1(), 3);
        __builtin_amdgcn_sched_barrier(0);
        finish(I0(), std::integral_constant<int, 2>());
        __builtin_amdgcn_sched_barrier(0);
        finish(I1(), std::integral_constant<int, 3>());
    }
}

std::atomic<unsigned long long> g_lds_done{0};

// fp32 NCHW [B][C][2h][2w] -> h2 tensor with 4C channels at h x w: channel q*C + c = the pixels (2y+qy, 2x+qx) of channel c, q = qy*2 + qx
// (space to depth: the form in which channels that live at the output resolution enter conv_up2_h2t_kernel as key chunks)
__global__ void h2_pack_s2d_kernel(const float* __restrict__ x, long long x_bs, unsigned short* __restrict__ y, long long y_bs,
                                   int C, int h, int w, long long total, unsigned* flag)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long hw = (long long)h * w;
    const int C8 = C >> 3;
    const long long pix = i % hw; const long long t = i / hw;
    const int oct = (int)(t % (4 * C8)); const int b = (int)(t / (4 * C8));
    const int q = oct / C8, c0 = (oct - q * C8) * 8;
    const int sy = (int)(pix / w), sx = (int)(pix - (long long)sy * w);
    const float* xb = x + (long long)b * x_bs + (long long)c0 * 4 * hw + (long long)(2 * sy + (q >> 1)) * (2 * w) + 2 * sx + (q & 1);
    half8 h8, l8;
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = xb[(long long)j * 4 * hw];
        const float hf = bfsr::pin_f16(v);
        h8[j] = (_Float16)hf; l8[j] = (_Float16)(v - hf);
        amax = fmaxf(amax, fabsf(v));
    }
    unsigned short* yb = y + (long long)b * y_bs + ((long long)oct * 2 * hw + pix) * 8;
    *reinterpret_cast<half8*>(yb) = h8;
    *reinterpret_cast<half8*>(yb + hw * 8) = l8;
    if (flag && !(amax < 65504.f)) atomicOr(flag, 1u);
}

}  // namespace

// ---- C ABI ------------------------------------------------------------------------------------------------------------------------
extern "C" long long bfsr_conv_up2_h2t_packed_size(int Cout, int Ct, int Ck)
{
    if (Cout <= 0 || Ct < 0 || Ck < 0 || Ct + Ck <= 0 || Cout % 32 || Ct % 16 || Ck % 16) return -1;
    return (long long)(Cout / 32) * (Ct / 16 + 4 * (Ck / 16)) * (W_ST / 2);           // fp16 elements
}

// w_taps: OIHW 3x3 fp32 over the Ct upsampled channels; w_key (NULL when Ck = 0): OIHW 3x3 over the Ck channels at output resolution.
// packed: [cout group of 32][chunk][plane hi, lo][step][k half][32 couts][8 channels] fp16 of scale * weight, 16 step slots per plane:
//   taps chunk (16 channels): step (px*2+b)*4 + py*2+a = the sum of the window taps of output parity (py, px) that fall on source tap (a, b)
//     -- rows dy in {0} | {1, 2} for py = 0 and {0, 1} | {2} for py = 1, same for columns; formed in double, split once (22 bits);
//   key chunk (space-to-depth plane q, 16 channels): the 9 steps of step_info(1 + q, s), each ONE window tap (dy, dx); slots 9..15 zero.
extern "C" int bfsr_pack_conv_up2_h2t(const float* w_taps, const float* w_key, int Cout, int Ct, int Ck, float scale, unsigned short* packed)
{
    if (!packed || Cout <= 0 || Ct < 0 || Ck < 0 || Ct + Ck <= 0 || Cout % 32 || Ct % 16 || Ck % 16 || !(scale > 0.f)) return -1;
    if ((Ct > 0 && !w_taps) || (Ck > 0 && !w_key)) return -1;
    const int nct = Ct / 16, nck = Ck / 16, nchunk = nct + 4 * nck;
    static const int lo_[2][2] = {{0, 1}, {0, 2}}, hi_[2][2] = {{0, 2}, {1, 2}};      // [parity][tap]: window index range [lo, hi]
    _Float16* out = reinterpret_cast<_Float16*>(packed);
    auto put = [&](_Float16* blk, int s, int kh, int co, int c, double v64) {
        const float v = (float)(v64 * (double)scale);
        const _Float16 hi = (_Float16)v;
        const long long e = ((long long)(s * 2 + kh) * 32 + co) * 8 + c;
        blk[e] = hi;
        blk[W_PL / 2 + e] = (_Float16)(v - (float)hi);
    };
    for (int cg = 0; cg < Cout / 32; ++cg)
        for (int k = 0; k < nchunk; ++k) {
            _Float16* blk = out + ((long long)cg * nchunk + k) * (W_ST / 2);
            for (int e = 0; e < W_ST / 2; ++e) blk[e] = (_Float16)0.f;
            const int type = k < nct ? 0 : 1 + (k - nct) / nck;
            const int kc = k < nct ? k : (k - nct) % nck;
            for (int s = 0; s < nsteps(type); ++s) {
                const StepInfo si = step_info(type, s);
                const int px = si.p4 & 1, py = si.p4 >> 1, b = (s >> 2) & 1, a = s & 1;
                for (int kh = 0; kh < 2; ++kh)
                    for (int co = 0; co < 32; ++co)
                        for (int c = 0; c < 8; ++c) {
                            if (type == 0) {
                                const float* wp = w_taps + ((long long)(cg * 32 + co) * Ct + (16 * kc + 8 * kh + c)) * 9;
                                double sum = 0.0;
                                for (int dy = lo_[py][a]; dy <= hi_[py][a]; ++dy)
                                    for (int dx = lo_[px][b]; dx <= hi_[px][b]; ++dx) sum += (double)wp[dy * 3 + dx];
                                put(blk, s, kh, co, c, sum);
                            } else {
                                const float* wp = w_key + ((long long)(cg * 32 + co) * Ck + (16 * kc + 8 * kh + c)) * 9;
                                put(blk, s, kh, co, c, (double)wp[si.dy * 3 + si.dx]);
                            }
                        }
            }
        }
    return 0;
}

extern "C" int bfsr_conv2d_up2_h2t(const BfsrUp2H2Args* a, void* stream)
{
    if (!a || !a->x || !a->w || !a->y || a->B <= 0 || a->h <= 0 || a->w_ <= 0 || a->Cin <= 0 || a->Cin % 16 || a->Cout <= 0 || a->Cout % 32) return -1;
    if (a->Ckey < 0 || a->Ckey % 16 || 4 * a->Ckey > a->Cin) return -1;
    if (a->y_fmt != 1) return -1;                                                               // quad-major fp32 only
    // 16-byte accesses on y / pre_add (quad-major) and LDS-DMA on x: misaligned views are refused, not faulted on
    if ((reinterpret_cast<unsigned long long>(a->x) & 15) || (a->x_bs & 7)) return -1;
    if ((reinterpret_cast<unsigned long long>(a->y) & 15) || (a->y_bs & 3)) return -1;
    if (a->pre_add && ((reinterpret_cast<unsigned long long>(a->pre_add) & 15) || (a->pre_add_bs & 3))) return -1;
    if ((long long)(a->Cin / 8) * 2 * a->h * a->w_ * 16 >= (1LL << 31)) return -1;              // 32-bit buffer offsets (source, per sample)
    if (8LL * 4 * a->h * a->w_ * 16 >= (1LL << 31)) return -1;                                   // (8 output channel quads of one sample)
    const int tiles_x = (a->w_ + 31) / 32, tiles_y = (a->h + TH - 1) / TH, groups = a->Cout / 32;
    const long long nitems = (long long)a->B * tiles_x * tiles_y * groups;
    if (nitems >= (1LL << 31)) return -1;
    if (bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_up2_h2t_kernel), LDS_BYTES, g_lds_done) != 0) return -2;
    const int cus = bfsr::cu_count();
    if (cus <= 0) return -2;
    const int grid = (int)(nitems < cus ? nitems : cus);
    hipLaunchKernelGGL(conv_up2_h2t_kernel, dim3(grid), dim3(NWV * 64), LDS_BYTES, static_cast<hipStream_t>(stream), *a, tiles_x, tiles_y, groups, (int)nitems);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" int bfsr_h2_pack_s2d(const float* x, long long x_bs, unsigned short* y, long long y_bs, int B, int C, int h, int w, unsigned* flag, void* stream)
{
    if (!x || !y || B <= 0 || C <= 0 || (C & 7) || h <= 0 || w <= 0) return -1;
    if ((reinterpret_cast<unsigned long long>(y) & 15) || (y_bs & 7)) return -1;
    const long long total = (long long)B * (C / 2) * h * w;
    hipLaunchKernelGGL(h2_pack_s2d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, x_bs, y, y_bs, C, h, w, total, flag);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
