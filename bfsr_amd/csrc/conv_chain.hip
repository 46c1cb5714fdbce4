// conv_chain.hip -- a CHAIN of 3x3 'same' convs over h2 tensors in ONE persistent launch: the dense blocks of the RRDB encoder
// (SRFlow-LP/code/models/modules/RRDBNet_arch.py:25-65, LINF-LP/models/rrdb.py:38-74: five convs per ResidualDenseBlock, three blocks
// per RRDB, 23 RRDBs per trunk), at fp32-class accuracy on the fp16 matrix pipe (the two-term split of conv3x3_h2x_kernel, conv_h2s.hip:
// x = hi + lo in an h2 tensor, weights split and pre-scaled by a power of two, three products lo*hi + hi*lo + hi*hi per operand pair).
//
// Why (round 5): as one launch per conv the dense blocks were 1/3 of the config-2 step at 0.36 of the split's matrix-pipe bound -- not
// traffic, not bank conflicts: (a) tile quantisation (400 items on 256 persistent workgroups = 2 rounds for 1.56 rounds of work, every
// conv) and 8.5 us of ramp per launch x 345 launches, and (b) 2.7 us per 16-channel chunk against 1.8 us of MFMA time: with two LDS
// stages of 59 KB a chunk's DMA is issued exactly one chunk before it is needed (profiles/r04_g_h2x_ablation.txt).
//
// What this kernel does about it:
//   * the items of ALL convs of the chain form one list, (conv, sample, tile row, tile column, cout group) in that order, dealt round-robin
//     to one persistent workgroup per CU; there is no grid-wide barrier between convs: an item of conv c waits only until the (up to 9)
//     tiles of its 3x3 tile neighbourhood have been finished by conv c-1 (a per-tile progress counter in global memory).  Because every
//     conv waits for its predecessor on the neighbourhood, all earlier readers and writers of anything the item touches are complete by
//     induction (ring buffers of the dense blocks included), so the chain may be as long as the caller likes (an RDB, an RRDB, the trunk);
//   * hand-off between workgroups (probe: tools/exp/xcd_handoff_probe.hip, profiles/r05_xcd_handoff_probe.txt):
//     the producer's h2 outputs are 16-byte WRITE-THROUGH stores (`sc1`), each compute wave drains them (`s_waitcnt vmcnt(0)`) and then adds
//     1 to the tile's counter with an agent-scope atomic; the consumer's loader waves poll the counters (relaxed agent loads) and read
//     activations ONLY with `sc1` LDS-DMA / `sc1` buffer loads, which bypass the CU's L1 (never refreshed by other CUs' stores); measured
//     0 stale words in 1.3e8 across and inside XCDs, false sharing of a line included, against 100 % stale for plain loads;
//   * the wait sits in the loader waves, which run two LDS stages ahead of the matrix pipe and keep serving the stage barriers of the
//     current item while the next item's tiles are not ready (a workgroup may wait for its own previous item);
//   * tile = 32 rows x 32 pixels, a compute wave owns FOUR rows (4 accumulator blocks: every weight fragment feeds four MFMAs), an LDS
//     stage = ONE channel octet: input [2 planes][34 x 34 positions, padded to 1216][8] + weights [2 planes][9 taps][32][8] = 48 128 B,
//     THREE stages: the DMA of a stage is issued two stages (~3.7 us) before it is consumed instead of one chunk (1.8 us).
//     With 8-channel stages the K = 16 of v_mfma_f32_32x32x16_f16 is TWO TAPS x 8 channels (lanes 0-31 hold the first tap's operand,
//     lanes 32-63 the second's: an LDS address per lane): taps (dy, 0 | dy, 1) for dy = 0..2 -- their B fragment of an input row serves
//     three output rows --, (0, 2 | 1, 2), and the ninth tap (2, 2) as [w_hi | w_hi] x [x_hi | x_lo] (hi*hi + hi*lo in one instruction)
//     plus [w_lo | 0] x [x_hi | x_lo]: 14 MFMAs per row and octet for 13.5 of arithmetic, 34 ds_read_b128 per 56 MFMAs and wave.
//   * epilogue = conv3x3_h2x_kernel's (bias / affine / activation / two h2 residuals, h2 | fp32 NCHW | fp32 quad-major output, range
//     guard of the fp16 split), plus an optional second fp32 NCHW copy of the result (tapped RRDB outputs).
// The spin on the counters is bounded: after ~2 s without progress a workgroup raises bit 2 of the status word and every waiter gives
// up (the results are then garbage and the host raises) -- a hung GPU is never the failure mode.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "../../include/bfsr_hip.h"
#include "launch_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4c __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) unsigned gu32;

#ifndef BFSR_CHAIN_ABL
#define BFSR_CHAIN_ABL 0                        // ablation builds only (tools/exp/chain_abl.sh): bit 0 no fragment reads, 1 no MFMAs, 2 no DMA, 3 no epilogue, 4 no drain before the publish, 5 plain (not sc1) loads and stores
#endif

namespace {

constexpr int NW = 8, NLW = 4;                  // compute waves, loader waves
constexpr int TW = 32, PW = TW + 2;
constexpr int WPL = 9 * 512;                    // one weight plane of a stage: [tap = dy*3 + dx][32 couts][8] fp16
constexpr int W_BYTES = 2 * WPL;                // 9 216 = nine 1-KiB DMA pieces
// R = rows per compute wave: 4 (tile 32 x 32, three LDS stages of 48 128 B) or 2 (tile 16 x 32, five stages of 29 696 B: twice the tiles
// and half the time per item -- for batches whose 32-row tiles would not fill the chip: the per-tile chain of a dense block is sequential)
template <int R> struct Geo {
    static constexpr int TH = NW * R, PR = TH + 2;
    static constexpr int NPOS = PR * PW;            // 1156 | 612 positions of the haloed tile
    static constexpr int NG = (NPOS + 63) / 64;     // 19 | 10 groups of 64 positions (one LDS-DMA instruction each, per plane)
    static constexpr int NPOSP = NG * 64;           // 1216 | 640
    static constexpr int PLANE = NPOSP * 16;        // one plane (hi or lo) of a stage's input octet
    static constexpr int IN_BYTES = 2 * PLANE;
    static constexpr int STAGE = IN_BYTES + W_BYTES;
    static constexpr int NS = R == 4 ? 3 : 5;
    static constexpr int LDS_TOTAL = NS * STAGE;    // 144 384 | 148 480
    static constexpr int ZERO_OFF = NPOS * 16;      // first padding position of plane 0: the DMA writes zeros there in every stage
    static_assert(NPOSP > NPOS, "the [w_lo | 0] operand needs a padding position");
};
constexpr unsigned OOB = 0x80000000u;
constexpr int AUX_SC1 = (BFSR_CHAIN_ABL & 32) ? 0 : 16;                     // cache-policy bit of the buffer builtins: sc1 (agent scope: bypass L1 / write through)
constexpr unsigned POLL_LIMIT = 1u << 21;       // unsuccessful polls (~1 us each) before a workgroup gives up

struct ChainHeader {                            // 64 bytes
    int magic, nconv, B, H, W, tiles_x, tiles_y, nitems;
    int rows;                                   // rows per compute wave: 4 (32-row tiles) or 2 (16-row tiles)
    int pad[7];
};
struct ChainRec {                               // one conv of the chain, device-side (opaque to the callers: bfsr_conv_chain_prepare fills it)
    const unsigned short* x; long long x_bs;
    const unsigned short* w;
    void* y; long long y_bs;
    const float* epi;
    const unsigned short* res1; long long res1_bs;
    const unsigned short* res2; long long res2_bs;
    float* y2; long long y2_bs;
    int Cin, Cout, y_fmt, act;
    float slope, alpha1, alpha2, acc_scale;
    int groups, noct, item_base;
    unsigned wait_target;                       // progress every tile of the 3x3 neighbourhood must have reached (0: no wait)
};
constexpr int CHAIN_MAGIC = 0x43484e31;

__device__ __forceinline__ void split2c(float v, _Float16& h, _Float16& l)
{
    h = (_Float16)v;
    l = (_Float16)(v - (float)h);
}

__device__ __forceinline__ void wait_vmcnt_c(int n)
{
    switch (n) {
#define W_(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
        W_(7) W_(8) W_(11) W_(12) W_(14) W_(16) W_(21) W_(22) W_(24) W_(28) W_(32)
#undef W_
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

struct CItem { int b, ty, tx, grp; };

__device__ __forceinline__ f32x16 mm_(half8 a, half8 b, f32x16 c)
{
    if (BFSR_CHAIN_ABL & 2) { c[0] += (float)a[0] + (float)b[0]; return c; }      // keeps the fragment reads alive
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <int R>
__global__ __launch_bounds__((NW + NLW) * 64, 1) void conv_chain_kernel(const ChainRec* __restrict__ recs, int B, int H, int W, int tiles_x, int tiles_y,
                                                                         int nitems, unsigned* progress, unsigned* status)
{
    typedef Geo<R> GE;
    constexpr int TH = GE::TH, NPOS = GE::NPOS, NG = GE::NG, PLANE = GE::PLANE, IN_BYTES = GE::IN_BYTES, STAGE = GE::STAGE, NS = GE::NS, ZERO_OFF = GE::ZERO_OFF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int G = gridDim.x;
    const int slot = (int)bfsr::xcd_order(blockIdx.x, (unsigned)G);      // one XCD walks a contiguous range of the item list in every round
    if (slot >= nitems) return;
    const long long HW = (long long)H * W;
    const unsigned HW16 = (unsigned)(H * W) * 16u;                       // bytes of one (octet, plane) image

    auto decode = [&](const ChainRec& r, int it) {
        CItem c;
        int t = it - r.item_base;
        c.grp = t % r.groups; t /= r.groups;
        c.tx = t % tiles_x; t /= tiles_x;
        c.ty = t % tiles_y; c.b = t / tiles_y;
        return c;
    };

    if (wave >= NW) {
        // ================================ loader waves: LDS-DMA only, plus the dependency polls ================================
        const int ld = wave - NW;
        const int pl = ld & 1, g0 = ld >> 1;                             // its plane; its position groups g0, g0 + 2, ...
        // the nine weight pieces: R = 4 (10 | 10 | 9 | 9 input pieces): 2, 1, 3, 3;  R = 2 (5 input pieces each): 3, 2, 2, 2
        const int wcount = R == 4 ? (ld == 0 ? 2 : (ld == 1 ? 1 : 3)) : (ld == 0 ? 3 : 2);
        const int wfirst = R == 4 ? (ld == 0 ? 0 : (ld == 1 ? 2 : (ld == 2 ? 3 : 6))) : (ld == 0 ? 0 : 1 + 2 * ld);
        const int np = (NG - g0 + 1) / 2 + wcount;                       // pieces per stage: 12, 11, 12, 12 | 8, 7, 7, 7
        constexpr int NGL = 10;                                          // position groups per loader, at most (fixed bound: see DESIGN.md on hipcc and template-dependent array bounds)
        unsigned vg[NGL];
        __amdgpu_buffer_rsrc_t rs_in, rs_w;
        int cur_noct = 0, cur_grp = 0;
        int it_issue = slot, oct_issue = 0, c_issue = 0;
        bool have = true, ready = false;
        unsigned polls = 0;
        int issued = 0, consumed = 0;

        auto deps_ready = [&](const ChainRec& r, const CItem& c) -> bool {
            if (r.wait_target == 0) return true;
            const int dy = lane / 3 - 1, dx = lane - (lane / 3) * 3 - 1;
            const int ty = c.ty + dy, tx = c.tx + dx;
            const bool nb = lane < 9 && ty >= 0 && ty < tiles_y && tx >= 0 && tx < tiles_x;
            unsigned v = 0xffffffffu;
            if (nb) v = __hip_atomic_load((gu32*)(progress + ((long long)c.b * tiles_y + ty) * tiles_x + tx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 9) v = __hip_atomic_load((gu32*)status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const bool ok = lane == 9 ? true : v >= r.wait_target;
            const bool dead = lane == 9 && (v & 4u);
            if (__any((int)dead)) return true;                            // another workgroup gave up: do not add a second timeout on top
            if (__all((int)ok)) { polls = 0; return true; }
            if (++polls > POLL_LIMIT) {
                if (lane == 0) atomicOr(status, 4u);
                return true;
            }
            return false;
        };
        auto lsetup = [&](const ChainRec& r, const CItem& c) {
            const unsigned short* xb = r.x + (long long)c.b * r.x_bs;
            rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xb), 0, (unsigned)(r.Cin >> 3) * 2u * HW16, 0x00020000);
            rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(r.w), 0, (unsigned)((long long)r.groups * r.noct * W_BYTES), 0x00020000);
            cur_noct = r.noct; cur_grp = c.grp;
            const int y0 = c.ty * TH, x0 = c.tx * TW;
#pragma unroll
            for (int j = 0; j < NGL; ++j) {
                const int g = g0 + 2 * j;
                const int pos = g * 64 + lane;
                const int rr = pos / PW, cc = pos - rr * PW;
                const int gy = y0 + rr - 1, gx = x0 + cc - 1;
                const bool ok = g < NG && pos < NPOS && gy >= 0 && gy < H && gx >= 0 && gx < W;
                vg[j] = ok ? (unsigned)(gy * W + gx) * 16u : OOB;        // out of range -> the DMA writes zeros (= the padding)
            }
        };
        auto lstage = [&](int oct, int buf) {
            if (BFSR_CHAIN_ABL & 4) return;
            unsigned char* base = smem + buf * STAGE;
            const unsigned soff = (unsigned)(oct * 2 + pl) * HW16;
#pragma unroll
            for (int j = 0; j < NGL; ++j) {
                const int g = g0 + 2 * j;
                if (g < NG)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(base + pl * PLANE + g * 1024), 16, vg[j], soff, 0, AUX_SC1);
            }
            const unsigned wsoff = (unsigned)(cur_grp * cur_noct + oct) * (unsigned)W_BYTES;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < wcount) {
                    const int piece = wfirst + j;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(base + IN_BYTES + piece * 1024), 16,
                                                             (unsigned)lane * 16u + (unsigned)piece * 1024u, wsoff, 0, 0);
                }
        };
        auto try_issue = [&]() -> bool {
            if (oct_issue == 0 && !ready) {
                while (it_issue >= recs[c_issue + 1].item_base) ++c_issue;
                const ChainRec& r = recs[c_issue];
                const CItem c = decode(r, it_issue);
                if (!deps_ready(r, c)) return false;
                lsetup(r, c);
                ready = true;
            }
            lstage(oct_issue, issued % NS);
            ++issued;
            if (++oct_issue == cur_noct) { oct_issue = 0; ready = false; it_issue += G; have = it_issue < nitems; }
            return true;
        };
        while (true) {
            // stage s may be issued once its LDS slot is free: s < NS, or barrier s - NS + 1 has been passed (the compute waves pass barrier
            // k only after their last read of stage k - 1)
            while (have && (issued < NS || issued <= consumed + NS - 2)) {
                if (!try_issue()) break;
            }
            if (consumed < issued) {
                wait_vmcnt_c((issued - consumed - 1) * np);              // all but the stages issued after stage `consumed`
                __builtin_amdgcn_s_barrier();
                ++consumed;
            } else if (have) {
                __builtin_amdgcn_s_sleep(8);                              // nothing in flight: the next item's tiles are not ready yet
            } else {
                break;
            }
        }
        return;
    }

    // ==================================================== compute waves ====================================================
    // wave w owns rows R*w .. R*w + R-1 of the tile.  Per stage (one channel octet), five tap groups:
    //   P(dy), dy = 0..2: taps (dy,0 | dy,1): B fragment of input row i = [x(i, c) | x(i, c+1)], shared by the output rows i - dy
    //   Q: taps (0,2 | 1,2): B = [x(r, c+2) | x(r+1, c+2)] per output row r        S: tap (2,2): B = [x_hi(r+2, c+2) | x_lo(r+2, c+2)]
    const unsigned row0 = (unsigned)R * (unsigned)wave;
    const unsigned oP = ((row0) * PW + l31 + lhi) * 16u;                 // + i * PW*16 + plane * PLANE
    const unsigned oQ = ((row0 + lhi) * PW + l31 + 2) * 16u;            // + r * PW*16 + plane * PLANE
    const unsigned oS = (unsigned)lhi * PLANE + ((row0 + 2) * PW + l31 + 2) * 16u;       // + r * PW*16
    const unsigned aP = IN_BYTES + (unsigned)lhi * 512u + l31 * 16u;    // + dy * 1536 + plane * WPL   (tap dy*3 + lhi)
    const unsigned aQ = IN_BYTES + ((unsigned)lhi * 3u + 2u) * 512u + l31 * 16u;          // taps 2 | 5, + plane * WPL
    const unsigned aSh = IN_BYTES + 8u * 512u + l31 * 16u;              // tap 8, hi plane, in both halves
    const unsigned aSl = lhi ? (unsigned)ZERO_OFF : IN_BYTES + WPL + 8u * 512u + l31 * 16u;   // [w_lo | 0]

    f32x16 acc[4];
    half8 rP[6][2], wP[3][2], rQ[4][2], wQ[2], rS[4], wS[2];
    auto ldh = [&](const unsigned char* sb, unsigned off) {
        if (BFSR_CHAIN_ABL & 1) { half8 v; for (int i = 0; i < 8; ++i) v[i] = (_Float16)(0.001f * (float)(lane + i + (int)(off & 15u))); return v; }
        return *reinterpret_cast<const half8*>(sb + off);
    };
    auto load_first = [&](const unsigned char* sb) {                     // rows 0..R-1 of P and the weights of P(0)
#pragma unroll
        for (int i = 0; i < R; ++i) { rP[i][0] = ldh(sb, oP + i * (PW * 16)); rP[i][1] = ldh(sb, oP + i * (PW * 16) + PLANE); }
        wP[0][0] = ldh(sb, aP); wP[0][1] = ldh(sb, aP + WPL);
    };
    // P(dy): output row r takes input row r + dy.  Smallest terms first (w_lo*x_hi, w_hi*x_lo, w_hi*x_hi), the rows interleaved:
    // independent accumulators back to back
    auto mfmaP = [&](auto dy_) {
        constexpr int dy = decltype(dy_)::value;
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = mm_(wP[dy][1], rP[r + dy][0], acc[r]);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = mm_(wP[dy][0], rP[r + dy][1], acc[r]);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = mm_(wP[dy][0], rP[r + dy][0], acc[r]);
    };
    auto mfmaQ = [&](auto a_) {                                          // rows a, a + 1
        constexpr int a = decltype(a_)::value;
        acc[a] = mm_(wQ[1], rQ[a][0], acc[a]);
        acc[a + 1] = mm_(wQ[1], rQ[a + 1][0], acc[a + 1]);
        acc[a] = mm_(wQ[0], rQ[a][1], acc[a]);
        acc[a + 1] = mm_(wQ[0], rQ[a + 1][1], acc[a + 1]);
        acc[a] = mm_(wQ[0], rQ[a][0], acc[a]);
        acc[a + 1] = mm_(wQ[0], rQ[a + 1][0], acc[a + 1]);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;

    int st = 0;                                                          // LDS slot of the next stage
    auto stage_body = [&](bool last) {
        const unsigned char* sb = smem + st * STAGE;
        rP[R][0] = ldh(sb, oP + R * (PW * 16)); rP[R][1] = ldh(sb, oP + R * (PW * 16) + PLANE);
        wP[1][0] = ldh(sb, aP + 1536); wP[1][1] = ldh(sb, aP + 1536 + WPL);
        __builtin_amdgcn_sched_barrier(0);
        mfmaP(I0());
        __builtin_amdgcn_sched_barrier(0);
        rP[R + 1][0] = ldh(sb, oP + (R + 1) * (PW * 16)); rP[R + 1][1] = ldh(sb, oP + (R + 1) * (PW * 16) + PLANE);
        wP[2][0] = ldh(sb, aP + 3072); wP[2][1] = ldh(sb, aP + 3072 + WPL);
        __builtin_amdgcn_sched_barrier(0);
        mfmaP(I1());
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 2; ++r) { rQ[r][0] = ldh(sb, oQ + r * (PW * 16)); rQ[r][1] = ldh(sb, oQ + r * (PW * 16) + PLANE); }
        wQ[0] = ldh(sb, aQ); wQ[1] = ldh(sb, aQ + WPL);
        __builtin_amdgcn_sched_barrier(0);
        mfmaP(I2());
        __builtin_amdgcn_sched_barrier(0);
        if (R == 4) {
#pragma unroll
            for (int r = 2; r < 4; ++r) { rQ[r][0] = ldh(sb, oQ + r * (PW * 16)); rQ[r][1] = ldh(sb, oQ + r * (PW * 16) + PLANE); }
            __builtin_amdgcn_sched_barrier(0);
            mfmaQ(I0());
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) rS[r] = ldh(sb, oS + r * (PW * 16));
        wS[0] = ldh(sb, aSh); wS[1] = ldh(sb, aSl);
        __builtin_amdgcn_sched_barrier(0);
        if (R == 4) mfmaQ(I2()); else mfmaQ(I0());
        __builtin_amdgcn_sched_barrier(0);
        const int nst = st + 1 == NS ? 0 : st + 1;
        if (!last) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // every fragment of this stage is in registers: the slot may be refilled
            __builtin_amdgcn_s_barrier();                                // the next stage has landed
            load_first(smem + nst * STAGE);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = mm_(wS[1], rS[r], acc[r]);  // w_lo * x_hi
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = mm_(wS[0], rS[r], acc[r]);  // w_hi * (x_hi + x_lo)
        __builtin_amdgcn_sched_barrier(0);
        st = nst;
    };

    float xamax = 0.f;                                                   // range guard: max |value| handed to the fp16 split (h2 output)
    int c = 0;
    for (int it = slot; it < nitems; it += G) {
        while (it >= recs[c + 1].item_base) ++c;
        const ChainRec& rec = recs[c];
        const CItem cur = decode(rec, it);
        const int Cout = rec.Cout, y_fmt = rec.y_fmt;
        const float slope = rec.act == BFSR_ACT_NONE ? 1.f : (rec.act == BFSR_ACT_RELU ? 0.f : rec.slope);
        const float acc_scale = rec.acc_scale;
        const float4* epi = reinterpret_cast<const float4*>(rec.epi);
        float4 pm;
        {
            int ln = lane;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(ln));                                 // per-lane address arithmetic stays inside the item loop
#endif
            const int idx = cur.grp * 64 + ln;
            pm = (ln & 1) ? make_float4(1.f, 0.f, 0.f, 0.f) : make_float4(0.f, 0.f, 1.f, 0.f);
            if (epi && (idx >> 1) < Cout) pm = epi[idx];
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
        const int noct = rec.noct;
        __builtin_amdgcn_s_barrier();                                    // the item's first stage has landed
        load_first(smem + st * STAGE);
        for (int o = 0; o + 1 < noct; ++o) stage_body(false);
        stage_body(true);

        if (BFSR_CHAIN_ABL & 8) {                                        // keep the accumulators alive
            if (acc[0][0] + acc[1][5] + acc[R - 2][7] + acc[R - 1][9] == 1234.5f) reinterpret_cast<float*>(rec.y)[lane] = acc[0][1];
            if (lane == 0)
                __hip_atomic_fetch_add((gu32*)(progress + ((long long)cur.b * tiles_y + cur.ty) * tiles_x + cur.tx), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        // ---- epilogue (the loaders are already staging the next item): conv3x3_h2x_kernel's, two rows at a time
        const bool plain = (lane & 1) ? pm.x == 1.f : (pm.y == 0.f && pm.z == 1.f && pm.w == 0.f);
        const bool bias_only = __all((int)plain);
        const bool fast = bias_only && slope >= 0.f && slope <= 1.f;
        auto fetch = [&](float val, int src_lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane * 4, __float_as_int(val))); };
        int lh = lhi, lx = l31;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(lh), "+v"(lx));
#endif
        const int gx = cur.tx * TW + lx;
        const int oct0 = cur.grp * 4;                                    // first of this item's four channel octets (+ q*2 + lh)
        const unsigned h2_bytes = (unsigned)((long long)(Cout >> 3) * 2 * HW * 16);
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");               // MFMA result -> VALU read inside the asm below
#pragma unroll
        for (int jj = 0; jj < R / 2; ++jj) {
            // every global access is a buffer instruction whose VGPR offset is out of range for pixels outside the image (and whose
            // descriptor ends at Cout channels): no `if (inside)` branch per access
            unsigned vo16[2], vo4[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int gy = cur.ty * TH + R * wave + 2 * jj + j;
                const bool ok = gy < H && gx < W;
                vo16[j] = ok ? (unsigned)(((long long)lh * 2 * HW + (long long)gy * W + gx) * 16) : OOB;
                vo4[j] = ok ? (unsigned)(((long long)lh * 8 * HW + (long long)gy * W + gx) * 4) : OOB;
            }
            half8 rh[2][2], rl[2][2];
            auto load_res = [&](const unsigned short* res, long long bs) {
                const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(res + (long long)cur.b * bs), 0, h2_bytes, 0x00020000);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const unsigned so = (unsigned)((oct0 + q * 2) * 2) * (unsigned)(HW * 16);
                        rh[j][q] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rr, vo16[j], so, AUX_SC1));
                        rl[j][q] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rr, vo16[j], so + (unsigned)(HW * 16), AUX_SC1));
                    }
            };
            if (rec.res1) load_res(rec.res1, rec.res1_bs);               // lands under the swaps / parameter exchange / activation
            float o[2][2][8];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float lo = acc[2 * jj + j][8 * q + i] * acc_scale, hi = acc[2 * jj + j][8 * q + 4 + i] * acc_scale;
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
                            o[j][q][i] = fmaxf(u, u * slope);            // = u > 0 ? u : u*slope for 0 <= slope <= 1
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int src = ((q * 2 + lh) * 8 + i) * 2;     // lane holding this channel's first float4
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
            if (rec.res1) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int i = 0; i < 8; ++i) o[j][q][i] = rec.alpha1 * o[j][q][i] + ((float)rh[j][q][i] + (float)rl[j][q][i]);
            }
            if (rec.res2) {
                load_res(rec.res2, rec.res2_bs);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int i = 0; i < 8; ++i) o[j][q][i] = rec.alpha2 * o[j][q][i] + ((float)rh[j][q][i] + (float)rl[j][q][i]);
            }
            if (y_fmt == 1) {                                            // h2: write-through (another workgroup reads it inside this launch)
                const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned short*>(rec.y) + (long long)cur.b * rec.y_bs, 0, h2_bytes, 0x00020000);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        half8 h8, l8;
#pragma unroll
                        for (int i = 0; i < 8; ++i) { _Float16 h, l; split2c(o[j][q][i], h, l); h8[i] = h; l8[i] = l; xamax = fmaxf(xamax, fabsf(o[j][q][i])); }
                        const unsigned so = (unsigned)((oct0 + q * 2) * 2) * (unsigned)(HW * 16);
                        // literal soffset 0 (launch_util.h, store_b128: the gfx950 store-data hazard), sc1
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4c, h8), ry, vo16[j] + so, 0, AUX_SC1);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4c, l8), ry, vo16[j] + so + (unsigned)(HW * 16), 0, AUX_SC1);
                    }
            } else {
                const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(rec.y) + (long long)cur.b * rec.y_bs, 0,
                                                                                    (unsigned)((long long)Cout * HW * 4), 0x00020000);
                if (y_fmt == 2) {                                        // fp32 quad-major [Cout/4][H][W][4]
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const unsigned so = (unsigned)((oct0 + q * 2) * 2) * (unsigned)(HW * 16);
                            bfsr::store_b128(ry, __builtin_bit_cast(u32x4c, make_float4(o[j][q][0], o[j][q][1], o[j][q][2], o[j][q][3])), vo16[j], so);
                            bfsr::store_b128(ry, __builtin_bit_cast(u32x4c, make_float4(o[j][q][4], o[j][q][5], o[j][q][6], o[j][q][7])), vo16[j], so + (unsigned)(HW * 16));
                        }
                } else {                                                 // fp32 NCHW: channels >= Cout fall beyond the descriptor
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int i = 0; i < 8; ++i)
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[j][q][i]), ry, vo4[j], (unsigned)(((oct0 + q * 2) * 8 + i) * HW * 4), 0);
                }
            }
            if (rec.y2) {                                                // second copy, fp32 NCHW (tapped block outputs; read after the launch)
                const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(rec.y2 + (long long)cur.b * rec.y2_bs, 0,
                                                                                    (unsigned)((long long)Cout * HW * 4), 0x00020000);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[j][q][i]), r2, vo4[j], (unsigned)(((oct0 + q * 2) * 8 + i) * HW * 4), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- publish: this wave's stores have left the CU (write-through) -> one more finished wave on the tile's counter
        if (!(BFSR_CHAIN_ABL & 16)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
            __hip_atomic_fetch_add((gu32*)(progress + ((long long)cur.b * tiles_y + cur.ty) * tiles_x + cur.tx), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (status && __any((int)!(xamax < 65504.f))) { if (lane == 0) atomicOr(status, 1u); }
}

inline unsigned short f32_to_f16_bits_c(float v)
{
    const _Float16 h = (_Float16)v;              // round to nearest even, like the device conversion
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}

}  // namespace

extern "C" long long bfsr_conv_packed_size_h2c(int Cout, int Cin)
{
    if (Cout <= 0 || Cin <= 0 || (Cin & 7)) return -1;
    return (long long)((Cout + 31) / 32) * (Cin / 8) * (W_BYTES / 2);      // fp16 elements
}

extern "C" int bfsr_pack_conv_weight_h2c(const float* w, int Cout, int Cin, float scale, unsigned short* packed)
{
    // w [Cout][Cin][3][3] fp32 -> fp16 [cout group of 32][channel octet][plane hi,lo][tap = dy*3 + dx][32 couts][8 channels] of w*scale,
    // zero padded; scale = a power of two chosen by the caller (largest |w|*scale in [2^9, 2^10))
    if (!w || !packed || Cout <= 0 || Cin <= 0 || (Cin & 7) || !(scale > 0.f)) return -1;
    const int noct = Cin / 8;
    const long long n = bfsr_conv_packed_size_h2c(Cout, Cin);
    for (long long i = 0; i < n; ++i) packed[i] = 0;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx) {
                    const float v = w[((long long)co * Cin + ci) * 9 + dy * 3 + dx] * scale;
                    const _Float16 h = (_Float16)v;
                    const _Float16 l = (_Float16)(v - (float)h);
                    const long long blk = ((long long)(co / 32) * noct + ci / 8) * (W_BYTES / 2);
                    const long long in = ((long long)(dy * 3 + dx) * 32 + co % 32) * 8 + ci % 8;
                    packed[blk + in] = f32_to_f16_bits_c((float)h);
                    packed[blk + WPL / 2 + in] = f32_to_f16_bits_c((float)l);
                }
    return 0;
}

extern "C" long long bfsr_conv_chain_table_size(int nconv)
{
    if (nconv <= 0) return -1;
    return (long long)sizeof(ChainHeader) + (long long)(nconv + 1) * sizeof(ChainRec);
}

extern "C" int bfsr_conv_chain_prepare(const BfsrChainConv* convs, int nconv, int B, int H, int W, int rows, void* table)
{
    if (!convs || !table || nconv <= 0 || B <= 0 || H <= 0 || W <= 0) return -1;
    if (rows != 0 && rows != 2 && rows != 4) return -1;
    ChainHeader hd;
    std::memset(&hd, 0, sizeof(hd));
    hd.magic = CHAIN_MAGIC; hd.nconv = nconv; hd.B = B; hd.H = H; hd.W = W;
    hd.tiles_x = (W + TW - 1) / TW;
    if (rows == 0) {
        // 32-row tiles move fewer bytes per MFMA, but the convs of a dense block are sequential per tile: below ~2 tiles per CU the chain is
        // bound by that dependency, not by the work (tools/exp/chain_sim.py; measured at 8 x 160^2: profiles/r05_c_chain_bench.txt)
        int cus = bfsr::cu_count();
        if (cus <= 0) cus = 256;
        const long long t32 = (long long)hd.tiles_x * ((H + 31) / 32) * B;
        rows = t32 >= 2LL * cus ? 4 : 2;
    }
    hd.rows = rows;
    const int TH = NW * rows;
    hd.tiles_y = (H + TH - 1) / TH;
    const long long tiles = (long long)hd.tiles_x * hd.tiles_y * B;
    ChainRec* recs = reinterpret_cast<ChainRec*>(static_cast<unsigned char*>(table) + sizeof(ChainHeader));
    long long items = 0;
    unsigned long long target = 0;
    for (int i = 0; i < nconv; ++i) {
        const BfsrChainConv& a = convs[i];
        if (!a.x || !a.w || !a.y) return -1;
        if (a.Cin <= 0 || (a.Cin & 7) || a.Cout <= 0) return -1;
        if (a.y_fmt < 0 || a.y_fmt > 2) return -1;
        if (!(a.acc_scale > 0.f)) return -1;
        if ((a.y_fmt != 0 || a.res1 || a.res2) && (a.Cout & 7)) return -1;
        if (i + 1 < nconv && a.y_fmt != 1) return -1;                    // only the last conv may leave the h2 world (its output is not read inside the launch)
        if ((long long)(a.Cin / 8) * 2 * H * W * 16 >= (1LL << 31)) return -1;      // 32-bit byte offsets inside one batch item
        if ((long long)((a.Cout + 7) / 8) * 2 * H * W * 16 >= (1LL << 31)) return -1;
        if ((long long)a.Cout * H * W * 4 >= (1LL << 31) && (a.y_fmt != 1 || a.y2)) return -1;
        if ((reinterpret_cast<unsigned long long>(a.x) & 15) || (a.x_bs & 7)) return -1;
        if (a.y_fmt == 1 && ((reinterpret_cast<unsigned long long>(a.y) & 15) || (a.y_bs & 7))) return -1;
        if (a.y_fmt == 2 && ((reinterpret_cast<unsigned long long>(a.y) & 15) || (a.y_bs & 3))) return -1;
        if (a.res1 && ((reinterpret_cast<unsigned long long>(a.res1) & 15) || (a.res1_bs & 7))) return -1;
        if (a.res2 && ((reinterpret_cast<unsigned long long>(a.res2) & 15) || (a.res2_bs & 7))) return -1;
        if (bfsr_conv_packed_size_h2c(a.Cout, a.Cin) * 2 >= (1LL << 32)) return -1;
        ChainRec& r = recs[i];
        std::memset(&r, 0, sizeof(r));
        r.x = a.x; r.x_bs = a.x_bs; r.w = a.w; r.y = a.y; r.y_bs = a.y_bs; r.epi = a.epi;
        r.res1 = a.res1; r.res1_bs = a.res1_bs; r.res2 = a.res2; r.res2_bs = a.res2_bs; r.y2 = a.y2; r.y2_bs = a.y2_bs;
        r.Cin = a.Cin; r.Cout = a.Cout; r.y_fmt = a.y_fmt; r.act = a.act;
        r.slope = a.slope; r.alpha1 = a.alpha1; r.alpha2 = a.alpha2; r.acc_scale = a.acc_scale;
        r.groups = (a.Cout + 31) / 32; r.noct = a.Cin / 8;
        if (items > 0x7fffffffLL) return -1;
        r.item_base = (int)items;
        r.wait_target = (unsigned)target;                                // conv 0: 0 = no wait (its inputs were written before the launch)
        items += tiles * r.groups;
        target += (unsigned long long)NW * r.groups;                     // every compute wave of every item adds 1
        if (target >= 0xfffffff0ull) return -1;
    }
    if (items > 0x7fffffffLL) return -1;
    std::memset(&recs[nconv], 0, sizeof(ChainRec));
    recs[nconv].item_base = 0x7fffffff;                                  // sentinel for the conv search
    hd.nitems = (int)items;
    std::memcpy(table, &hd, sizeof(hd));
    return 0;
}

extern "C" long long bfsr_conv_chain_progress_words(const void* table_host)
{
    if (!table_host) return -1;
    const ChainHeader* hd = static_cast<const ChainHeader*>(table_host);
    if (hd->magic != CHAIN_MAGIC) return -1;
    return (long long)hd->tiles_x * hd->tiles_y * hd->B;
}

extern "C" int bfsr_conv_chain_launch(const void* table_host, const void* table_dev, unsigned* progress, unsigned* status, int tune, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!table_host || !table_dev || !progress || !status) return -1;
    const ChainHeader* hd = static_cast<const ChainHeader*>(table_host);
    if (hd->magic != CHAIN_MAGIC || hd->nitems <= 0) return -1;
    int cus = bfsr::cu_count();
    if (cus <= 0) return -1;
    // The dependency waits need every workgroup of the grid to be resident at the same time: one workgroup per CU (144 KB of LDS each), never
    // more workgroups than CUs.  `tune` may only shrink the grid.
    if (tune > 0 && tune < cus) cus = tune;
    const int grid = hd->nitems < cus ? hd->nitems : cus;
    static std::atomic<unsigned long long> lds_done4{0}, lds_done2{0};
    if (hd->rows == 4) { if (bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_chain_kernel<4>), Geo<4>::LDS_TOTAL, lds_done4) != 0) return -1; }
    else if (hd->rows == 2) { if (bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_chain_kernel<2>), Geo<2>::LDS_TOTAL, lds_done2) != 0) return -1; }
    else return -1;
    const long long words = (long long)hd->tiles_x * hd->tiles_y * hd->B;
    if (hipMemsetAsync(progress, 0, (size_t)words * sizeof(unsigned), st) != hipSuccess) return -1;
    const ChainRec* recs = reinterpret_cast<const ChainRec*>(static_cast<const unsigned char*>(table_dev) + sizeof(ChainHeader));
    if (hd->rows == 4)
        hipLaunchKernelGGL((conv_chain_kernel<4>), dim3((unsigned)grid), dim3((NW + NLW) * 64), Geo<4>::LDS_TOTAL, st, recs, hd->B, hd->H, hd->W, hd->tiles_x,
                           hd->tiles_y, hd->nitems, progress, status);
    else
        hipLaunchKernelGGL((conv_chain_kernel<2>), dim3((unsigned)grid), dim3((NW + NLW) * 64), Geo<2>::LDS_TOTAL, st, recs, hd->B, hd->H, hd->W, hd->tiles_x,
                           hd->tiles_y, hd->nitems, progress, status);
    return (int)hipGetLastError();
}
