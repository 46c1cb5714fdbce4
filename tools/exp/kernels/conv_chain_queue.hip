// ARCHIVED EXPERIMENT (round 6, not built): conv_chain.hip with CLAIMED items -- one queue per XCD (BFSR_CHAIN_DEAL=xcd) or one global queue (=global)
// instead of the static round-robin deal (=static).  Bit-identical results (tests/test_conv_chain.py: 79 passed), no gain: config 2 60.40 ms (xcd) against
// 60.17 (static), config 4 433.1 against 432.6; the global queue loses the L2 locality of neighbouring tiles (+13 % on the launch).  profiles/r06s_*.
// To rebuild the experiment: copy over bfsr_amd/csrc/conv_chain.hip, bfsr_amd/csrc/build.sh (bfsr_conv_chain_progress_words grows by 257 words).
// conv_chain.hip -- a CHAIN of bfsr_conv3x3_h2x convs in ONE persistent launch: the dense blocks of the RRDB encoder
// (SRFlow-LP/code/models/modules/RRDBNet_arch.py:25-65, LINF-LP/models/rrdb.py:38-74: five convs per ResidualDenseBlock, three blocks
// per RRDB, 23 RRDBs per trunk).  The arithmetic is conv3x3_h2x_kernel's (conv_h2s.hip), instruction for instruction: 16 x 32 tiles, two rows
// per compute wave, 16-channel chunks, one tap per pipeline step, three products lo*hi + hi*lo + hi*hi in the same order -- a chain of N convs
// produces the BITS of N bfsr_conv3x3_h2x launches (tests/test_conv_chain.py), so results do not depend on which of the two the host picks.
//
// Why (round 5): as one launch per conv the dense blocks pay tile quantisation on every conv (config 2: 400 items on 256 persistent
// workgroups = 2 rounds for 1.56 rounds of work) and ~8.5 us of ramp per launch, 345 launches per step.  Here the items of ALL convs of the
// chain form one list, (conv, sample, tile row, tile column, cout group) in that order, dealt round-robin to one persistent workgroup per CU;
// there is no grid-wide barrier between convs: an item of conv c waits only until the (up to 9) tiles of its 3 x 3 tile neighbourhood have been
// finished by conv c-1 (a per-tile progress counter in global memory).  Because every conv waits for its predecessor on the neighbourhood, all
// earlier readers and writers of anything the item touches are complete by induction (the ring buffers of the dense blocks included), so the
// chain may be as long as the caller likes (an RDB, an RRDB, the whole trunk).
//   * Hand-off between workgroups (probe: tools/exp/xcd_handoff_probe.hip, profiles/r05_xcd_handoff_probe.txt): the producer's h2 outputs are
//     16-byte WRITE-THROUGH stores (`sc1`), each compute wave drains them (`s_waitcnt vmcnt(0)`) and then adds 1 to the tile's counter with an
//     agent-scope atomic; the consumer's loader waves poll the counters (relaxed agent loads) and read activations ONLY with `sc1` LDS-DMA /
//     `sc1` buffer loads, which bypass the CU's L1 (never refreshed by other CUs' stores): 0 stale words in 1.3e8 across and inside XCDs, false
//     sharing of a cache line included, against 100 % stale for plain loads.
//   * The wait sits in the loader waves, which keep serving the chunk barriers of the current item while the next item's tiles are not ready
//     (a workgroup may wait for its own previous item).
//   * What was measured on the way (profiles/r05_*chain*.txt): a first version with its own arithmetic -- 32 x 32 tiles, four
//     rows per wave, 8-channel LDS stages three deep, K = two taps x 8 channels -- needs 40 % fewer LDS reads and 20 % fewer staged bytes per MFMA
//     and ran NO faster under sustained load: the chip is power-limited there (MFMA busy 0.76 at ~1.65 GHz against conv_h2x's 0.84 at ~1.51 GHz,
//     the same product), and that version's half-empty MFMA for the ninth tap (14 instead of 13.5 per row and octet) cost exactly its 3.7 %.
// The spin on the counters is bounded: after ~2 s without progress a workgroup raises the launch's PRIVATE give-up word (one word behind the
// progress counters, zeroed with them at every launch) and every waiter of THIS launch gives up; bit 2 of the caller's status word is raised as
// well so that the host can tell (the results are then garbage and the host raises) -- a hung GPU is never the failure mode, and a launch that
// gave up does not poison the launches after it (round 6: the give-up bit used to live in the shared, sticky status word only).
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

#ifndef BFSR_CHAIN_TRACE
#define BFSR_CHAIN_TRACE 0                      // measurement builds only (tools/exp/chain_trace.sh): per-item phase timestamps of a few workgroups
#endif

namespace {

#if BFSR_CHAIN_TRACE
constexpr int TR_WG = 16, TR_ITEMS = 1024, TR_F = 12;                    // workgroups with slot % 16 == 0, items per workgroup, fields per item
__device__ unsigned long long g_chain_trace[TR_WG * TR_ITEMS * TR_F];
__device__ __forceinline__ void tr_put(int slot, int n, int f, unsigned long long v, int lane)
{
    if (lane == 0 && (slot & 15) == 0 && (slot >> 4) < TR_WG && n < TR_ITEMS) g_chain_trace[((slot >> 4) * TR_ITEMS + n) * TR_F + f] = v;
}
#define TR_PUT(n_, f_, v_) tr_put(slot, n_, f_, v_, lane)
#else
#define TR_PUT(it_, f_, v_) ((void)0)
#endif

constexpr int NW = 8, NLW = 4;                  // compute waves, loader waves
constexpr int TH = 16, TW = 32, PW = TW + 2, NPOS = (TH + 2) * PW, NG = 10, NPOSP = NG * 64;
constexpr int SUB = NPOSP * 16;                 // bytes of one (plane, k half) sub-image of a chunk: 8 channels of every tile position
constexpr int X_IN = 4 * SUB;                   // 40 960: [plane hi,lo][k half][640 positions][8]
constexpr int X_WPL = 9 * 1024;                 // one weight plane of a chunk: [tap = dx*3 + dy][k half][32 couts][8]
constexpr int X_W = 2 * X_WPL;                  // 18 432 = eighteen 1-KiB DMA pieces
constexpr int STAGE = X_IN + X_W;               // 59 392
constexpr int NS = 2;
constexpr int RING = NS * STAGE;                // 118 784
constexpr int LDS_TOTAL = RING + 16 + 32;       // + one word: the newest item (sequence number) whose dependencies loader 0 has seen satisfied; + four {sequence number, item} pairs: the item queue
constexpr unsigned OOB = 0x80000000u;
constexpr int AUX_SC1 = (BFSR_CHAIN_ABL & 32) ? 0 : 16;                     // cache-policy bit of the buffer builtins: sc1 (agent scope: bypass L1 / write through)
constexpr unsigned POLL_LIMIT = 1u << 21;       // unsuccessful polls (~1 us each) before a workgroup gives up

struct ChainHeader {                            // 64 bytes
    int magic, nconv, B, H, W, tiles_x, tiles_y, nitems;
    int pad[8];
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
    int groups, nchunk, item_base;
    unsigned wait_target;                       // progress every tile of the 3x3 neighbourhood must have reached (0: no wait)
};
constexpr int CHAIN_MAGIC = 0x43484e31;

__device__ __forceinline__ void split2c(float v, _Float16& h, _Float16& l)
{
    const float hf = bfsr::pin_f16(v);
    h = (_Float16)hf;
    l = (_Float16)(v - hf);
}

__device__ __forceinline__ void wait_vmcnt_c(int n)
{
    switch (n) {
#define W_(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
        W_(14) W_(15)
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

__global__ __launch_bounds__((NW + NLW) * 64, 1) void conv_chain_kernel(const ChainRec* __restrict__ recs, int B, int H, int W, int tiles_x, int tiles_y,
                                                                         int nitems, unsigned* progress, unsigned* giveup, unsigned* queue, int qcpx, unsigned* status, int defer_ok)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    volatile int* lds_ready = reinterpret_cast<volatile int*>(smem + RING);
    // The item queue (round 6).  The n-th item of a workgroup is slot + n * G when `queue` is null (static round-robin deal) and otherwise -- for
    // n >= 1 -- claimed by loader 0 with one agent-scope atomic add when it is about to stage the item's first chunk, and handed as {n, item} to
    // the other eleven waves through four LDS slots.  qcpx > 0 (workgroups per XCD): every XCD keeps the contiguous range of each round the static
    // deal gives it (neighbouring tiles and the cout groups of a tile in one L2) and its workgroups claim from that sequence in order through the
    // XCD's own counter; qcpx == 0: one counter for the whole list (loses the L2 locality: measured slower, kept for A/B).  Either way a claimed
    // item's dependencies are earlier items of the list, and the earliest unfinished item of the launch is always running or claimable by a free
    // workgroup of its XCD (the items a workgroup holds are all earlier than any it has not claimed) -- no deadlock, as with the static deal.
    // Why: with the static deal a workgroup that falls behind keeps its share of every later round, and at 1.5 rounds between a tile and its
    // consumers (config 2: 400 tiles on 256 workgroups) its lateness becomes dependency waits of workgroups that are not behind (measured with
    // -DBFSR_CHAIN_TRACE: 3-5 us per conv1 / conv2 item, p90 10-15 us, where a launch with far-apart dependencies shows 1 us).
    volatile unsigned long long* lds_q = reinterpret_cast<volatile unsigned long long*>(smem + RING + 16);      // (sequence number << 32) | item
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
        // loader `ld` = sub-image (plane ld >> 1, k half ld & 1) of every position group + its share of the 18 weight pieces (ld, ld + 4, ...)
        const int np = NG + (X_W / 1024 - ld + NLW - 1) / NLW;           // pieces per chunk: 15, 15, 14, 14
        unsigned vg[NG];
        __amdgpu_buffer_rsrc_t rs_in, rs_w;
        int cur_nchunk = 0, cur_grp = 0;
        int it_issue = slot, n_issue = 0, k_issue = 0, c_issue = 0;
        bool have = true, ready = false, claimed = true;                 // item 0 is the workgroup's slot
        unsigned polls = 0;
        int issued = 0, consumed = 0;
#if BFSR_CHAIN_TRACE
        bool tr_polled = false;
#endif
        if (ld == 0 && lane == 0) {
            *lds_ready = -1;                                             // read by the compute waves at the end of their first item at the earliest
#pragma unroll
            for (int q = 0; q < 4; ++q) lds_q[q] = ~0ull;                // (first looked at after the first chunk barrier of item 0)
        }

        auto deps_ready = [&](const ChainRec& r, const CItem& c) -> bool {
            if (r.wait_target == 0) return true;
            const int dy = lane / 3 - 1, dx = lane - (lane / 3) * 3 - 1;
            const int ty = c.ty + dy, tx = c.tx + dx;
            const bool nb = lane < 9 && ty >= 0 && ty < tiles_y && tx >= 0 && tx < tiles_x;
            unsigned v = 0xffffffffu;
            if (nb) v = __hip_atomic_load((gu32*)(progress + ((long long)c.b * tiles_y + ty) * tiles_x + tx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 9) v = __hip_atomic_load((gu32*)giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const bool ok = lane == 9 ? true : v >= r.wait_target;
            const bool dead = lane == 9 && (v & 4u);
            if (__any((int)dead)) return true;                            // another workgroup gave up: do not add a second timeout on top
            if (__all((int)ok)) { polls = 0; return true; }
            if (++polls > POLL_LIMIT) {
                if (lane == 0) { atomicOr(giveup, 4u); atomicOr(status, 4u); }
                return true;
            }
            return false;
        };
        auto lsetup = [&](const ChainRec& r, const CItem& c) {
            const unsigned short* xb = r.x + (long long)c.b * r.x_bs;
            rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xb), 0, (unsigned)(r.Cin >> 3) * 2u * HW16, 0x00020000);
            rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(r.w), 0, (unsigned)((long long)r.groups * r.nchunk * X_W), 0x00020000);
            cur_nchunk = r.nchunk; cur_grp = c.grp;
            const int y0 = c.ty * TH, x0 = c.tx * TW;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int pos = g * 64 + lane;
                const int rr = pos / PW, cc = pos - rr * PW;
                const int gy = y0 + rr - 1, gx = x0 + cc - 1;
                const bool ok = pos < NPOS && gy >= 0 && gy < H && gx >= 0 && gx < W;
                vg[g] = ok ? (unsigned)(gy * W + gx) * 16u : OOB;        // out of range -> the DMA writes zeros (= the padding)
            }
        };
        auto lstage = [&](int k, int buf) {
            if (BFSR_CHAIN_ABL & 4) return;
            unsigned char* base = smem + buf * STAGE;
            const unsigned soff = (unsigned)((2 * k + (ld & 1)) * 2 + (ld >> 1)) * HW16;      // octet 2k + k half, plane ld >> 1
#pragma unroll
            for (int g = 0; g < NG; ++g)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(base + ld * SUB + g * 1024), 16, vg[g], soff, 0, AUX_SC1);
            const unsigned wsoff = (unsigned)(cur_grp * cur_nchunk + k) * (unsigned)X_W;
#pragma unroll
            for (int j = 0; j < (X_W / 1024 + NLW - 1) / NLW; ++j) {
                const int piece = ld + j * NLW;
                if (piece < X_W / 1024)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(base + X_IN + piece * 1024), 16,
                                                             (unsigned)lane * 16u + (unsigned)piece * 1024u, wsoff, 0, 0);
            }
        };
        auto try_issue = [&]() -> bool {
            if (k_issue == 0 && !ready) {
                if (!claimed) {
                    if (queue == nullptr) {
                        it_issue = slot + n_issue * G;
                    } else if (ld == 0) {
                        unsigned v = 0;
                        const int xcd = (int)(blockIdx.x & 7u);
                        if (lane == 0) v = __hip_atomic_fetch_add((gu32*)(queue + (qcpx > 0 ? xcd * 32 : 0)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v = __builtin_amdgcn_readfirstlane(v);
                        if (qcpx > 0) {
                            const unsigned m = (unsigned)qcpx + v, r = m / (unsigned)qcpx;
                            const unsigned long long i64 = (unsigned long long)r * (unsigned)G + (unsigned)(xcd * qcpx) + (m - r * (unsigned)qcpx);
                            it_issue = i64 > (unsigned long long)nitems ? nitems : (int)i64;
                        } else {
                            it_issue = v > (unsigned)(nitems - G) ? nitems : G + (int)v;
                        }
                        if (it_issue > nitems) it_issue = nitems;
                        if (lane == 0) lds_q[n_issue & 3] = ((unsigned long long)(unsigned)n_issue << 32) | (unsigned)it_issue;
                    } else {
                        const unsigned long long e = lds_q[n_issue & 3];
                        if ((int)(e >> 32) != n_issue) return false;     // loader 0 has not claimed it yet
                        it_issue = __builtin_amdgcn_readfirstlane((int)(unsigned)e);
                    }
                    claimed = true;
                    if (it_issue >= nitems) { have = false; return false; }
                }
                while (it_issue >= recs[c_issue + 1].item_base) ++c_issue;
                const ChainRec& r = recs[c_issue];
                const CItem c = decode(r, it_issue);
#if BFSR_CHAIN_TRACE
                if (ld == 0 && !tr_polled) { tr_polled = true; TR_PUT(n_issue, 7, wall_clock64()); }
#endif
                if (!deps_ready(r, c)) return false;
#if BFSR_CHAIN_TRACE
                if (ld == 0) { TR_PUT(n_issue, 8, wall_clock64()); tr_polled = false; }
#endif
                if (ld == 0 && lane == 0) *lds_ready = n_issue;          // the workgroup's item n_issue WILL start (the counters only grow): see the deferred publish
                lsetup(r, c);
                ready = true;
            }
            lstage(k_issue, issued % NS);
            ++issued;
            if (++k_issue == cur_nchunk) { k_issue = 0; ready = false; claimed = false; ++n_issue; }
            return true;
        };
        while (true) {
            // chunk s may be issued once its LDS slot is free: s < NS, or barrier s - NS + 1 has been passed (the compute waves pass barrier
            // k only after their last read of chunk k - 1): with NS = 2, chunk k + 1 goes out right behind barrier k -- conv3x3_h2x_kernel's protocol
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
    // conv3x3_h2x_kernel's K loop (conv_h2s.hip): wave w owns rows 2w, 2w + 1; a step = one tap (dx, dy): 2 input rows x 2 planes + the tap's 2
    // weight planes -> 6 MFMAs; the fragments of step t + 1 are read while the MFMAs of step t run (register double buffer).
    half8 bq[2][2][2], aq[2][2];                                         // [buffer][plane][row] | [buffer][plane]
    auto load_step = [&](auto buf_, int stg, int t) {
        constexpr int BUF = decltype(buf_)::value;
        if (BFSR_CHAIN_ABL & 1) return;
        const int dx = t / 3, dy = t - 3 * dx;
        const unsigned char* sIn = smem + stg * STAGE;
        const unsigned char* inB = sIn + (lhi * NPOSP + (2 * wave + dy) * PW + l31 + dx) * 16;
        const unsigned char* wA = sIn + X_IN + lane * 16 + t * 1024;     // tap = dx*3 + dy = t
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int r = 0; r < 2; ++r) bq[BUF][pl][r] = *reinterpret_cast<const half8*>(inB + pl * 2 * SUB + r * PW * 16);
            aq[BUF][pl] = *reinterpret_cast<const half8*>(wA + pl * X_WPL);
        }
    };
    f32x16 acc[2];
    if (BFSR_CHAIN_ABL & 1) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) bq[b][pl][r][i] = (_Float16)(0.001f * (lane + i + r));
#pragma unroll
                for (int i = 0; i < 8; ++i) aq[b][pl][i] = (_Float16)(0.002f * (lane + i));
            }
    }
    auto mfma_step = [&](auto buf_) {
        constexpr int BUF = decltype(buf_)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // smallest terms first: w_lo*x_hi, w_hi*x_lo, w_hi*x_hi
            acc[j] = mm_(aq[BUF][1], bq[BUF][0][j], acc[j]);
            acc[j] = mm_(aq[BUF][0], bq[BUF][1][j], acc[j]);
            acc[j] = mm_(aq[BUF][0], bq[BUF][0][j], acc[j]);
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    int st = 0;                                                          // LDS slot of the next chunk
    // One barrier per chunk, passed EARLY: chunk k + 1's barrier sits before the last tap of chunk k (whose fragments are already in registers), so
    // the first fragments of chunk k + 1 are in flight under that tap's MFMAs and the loaders may refill the slot one tap earlier.  Nine taps per
    // chunk flip the fragment-buffer parity from chunk to chunk: chunks are processed in pairs.
    auto chunk_body = [&](auto p_, auto q_, bool last) {                 // p_: buffer holding tap 0's fragments (already loaded)
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
            load_step(q_, st, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(p_);
            __builtin_amdgcn_sched_barrier(0);
            load_step(p_, st, t + 2);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(q_);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!last) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // tap 8's fragments have left the slot
            __builtin_amdgcn_s_barrier();                                // chunk k + 1 has landed in the other slot; this one is free again
            load_step(q_, st ^ 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(p_);
        __builtin_amdgcn_sched_barrier(0);
        st ^= 1;
    };

    // Publishing an item = waiting for this wave's (write-through) stores and adding 1 to the tile's counter.  Done right after the epilogue it
    // exposes the store drain (~1-2 us per item).  When the workgroup's NEXT item is already known to start (its dependencies were seen
    // satisfied by the loader, which runs ahead), the publish is deferred into that item's K loop, where the wait costs nothing.  It must not
    // be deferred otherwise: the next item may wait -- through other workgroups whose own publications are deferred the same way -- for this
    // very publication (a first rule, "defer unless the next item is the next conv on a neighbouring tile", deadlocked exactly like that).
    float xamax = 0.f;                                                   // range guard: max |value| handed to the fp16 split (h2 output)
    long long pend = -1;
    auto publish_pending = [&]() {
        if (!(BFSR_CHAIN_ABL & 16)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add((gu32*)(progress + pend), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend = -1;
    };
    int c = 0;
    for (int n = 0;; ++n) {
        int it = slot + n * G;
        if (queue != nullptr && n > 0) {                                 // the loaders claimed it while the previous item was in the matrix pipe
            unsigned long long e = lds_q[n & 3];
            while ((int)(e >> 32) != n) { __builtin_amdgcn_s_sleep(1); e = lds_q[n & 3]; }
            it = __builtin_amdgcn_readfirstlane((int)(unsigned)e);
        }
        if (it >= nitems) break;
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
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
        const int nchunk = rec.nchunk;
#if BFSR_CHAIN_TRACE
        if (wave == 0) { TR_PUT(n, 0, wall_clock64()); TR_PUT(n, 5, clock64()); TR_PUT(n, 6, (unsigned long long)c | ((unsigned long long)nchunk << 16) | ((unsigned long long)cur.grp << 32)); }
#endif
        __builtin_amdgcn_s_barrier();                                    // the item's first chunk has landed
#if BFSR_CHAIN_TRACE
        if (wave == 0) TR_PUT(n, 1, wall_clock64());
#endif
        load_step(I0(), st, 0);
        {
            int k = 0;
            for (; k + 2 <= nchunk; k += 2) {                            // pairs of chunks: the fragment-buffer parity is static inside a pair
                chunk_body(I0(), I1(), false);
                chunk_body(I1(), I0(), k + 2 == nchunk);
                if (pend >= 0) publish_pending();                        // the previous item's stores drained long ago: this wait is free
            }
            if (k < nchunk) chunk_body(I0(), I1(), true);
        }
        if (pend >= 0) publish_pending();
#if BFSR_CHAIN_TRACE
        if (wave == 0) TR_PUT(n, 2, wall_clock64());
#endif

        if (BFSR_CHAIN_ABL & 8) {                                        // keep the accumulators alive
            if (acc[0][0] + acc[1][5] == 1234.5f) reinterpret_cast<float*>(rec.y)[lane] = acc[0][1];
            if (lane == 0 && progress)
                __hip_atomic_fetch_add((gu32*)(progress + ((long long)cur.b * tiles_y + cur.ty) * tiles_x + cur.tx), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        // ---- epilogue (the loaders are already staging the next item): conv3x3_h2x_kernel's
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
        {
            // every global access is a buffer instruction whose VGPR offset is out of range for pixels outside the image (and whose
            // descriptor ends at Cout channels): no `if (inside)` branch per access
            unsigned vo16[2], vo4[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int gy = cur.ty * TH + 2 * wave + j;
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
                        float lo = acc[j][8 * q + i] * acc_scale, hi = acc[j][8 * q + 4 + i] * acc_scale;
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
                        for (int i = 0; i < 8; ++i) {
                            _Float16 h, l;
                            split2c(o[j][q][i], h, l);
                            h8[i] = h; l8[i] = l;
                            xamax = fmaxf(xamax, fabsf(o[j][q][i]));
                            o[j][q][i] = (float)h + (float)l;            // what later convs read: the second copy (y2) holds the same 22-bit value (= bfsr_h2_unpack of y)
                        }
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
#if BFSR_CHAIN_TRACE
        if (wave == 0) TR_PUT(n, 3, wall_clock64());
#endif
        // ---- publish: this wave's stores have left the CU (write-through) -> one more finished wave on the tile's counter
        if (!progress) continue;                                         // a chain of one: nobody waits
        pend = ((long long)cur.b * tiles_y + cur.ty) * tiles_x + cur.tx;
        // defer only if the workgroup's next item is known to start: loader 0 has already seen its dependencies satisfied (it runs ahead of the
        // matrix pipe, so this is the common case) -- then the publication cannot be what anybody is (transitively) waiting for before that start
        const bool defer = defer_ok && *lds_ready >= n + 1;
        if (!defer) publish_pending();
#if BFSR_CHAIN_TRACE
        if (wave == 0) { TR_PUT(n, 4, defer ? 0ull : (unsigned long long)wall_clock64()); TR_PUT(n, 9, clock64()); }
#endif
    }
    if (pend >= 0) publish_pending();
    if (status && __any((int)!(xamax < 65504.f))) { if (lane == 0) atomicOr(status, 1u); }
}

}  // namespace

#if BFSR_CHAIN_TRACE
extern "C" int bfsr_chain_trace_read(void* dst, long long bytes)          // measurement builds only: [16 workgroups][1024 items][12 fields] u64
{
    if (bytes > (long long)sizeof(g_chain_trace)) bytes = sizeof(g_chain_trace);
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_chain_trace), (size_t)bytes, 0, hipMemcpyDeviceToHost);
}
extern "C" int bfsr_chain_trace_clear()
{
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_chain_trace)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(g_chain_trace));
}
#endif

extern "C" long long bfsr_conv_chain_table_size(int nconv)
{
    if (nconv <= 0) return -1;
    return (long long)sizeof(ChainHeader) + (long long)(nconv + 1) * sizeof(ChainRec);
}

extern "C" int bfsr_conv_chain_prepare(const BfsrChainConv* convs, int nconv, int B, int H, int W, void* table)
{
    if (!convs || !table || nconv <= 0 || B <= 0 || H <= 0 || W <= 0) return -1;
    ChainHeader hd;
    std::memset(&hd, 0, sizeof(hd));
    hd.magic = CHAIN_MAGIC; hd.nconv = nconv; hd.B = B; hd.H = H; hd.W = W;
    hd.tiles_x = (W + TW - 1) / TW;
    hd.tiles_y = (H + TH - 1) / TH;
    const long long tiles = (long long)hd.tiles_x * hd.tiles_y * B;
    ChainRec* recs = reinterpret_cast<ChainRec*>(static_cast<unsigned char*>(table) + sizeof(ChainHeader));
    long long items = 0;
    unsigned long long target = 0;
    for (int i = 0; i < nconv; ++i) {
        const BfsrChainConv& a = convs[i];
        if (!a.x || !a.w || !a.y) return -1;
        if (a.Cin <= 0 || (a.Cin & 15) || a.Cout <= 0) return -1;
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
        if (bfsr_conv_packed_size_h2x(a.Cout, a.Cin, 1) * 2 >= (1LL << 32)) return -1;
        ChainRec& r = recs[i];
        std::memset(&r, 0, sizeof(r));
        r.x = a.x; r.x_bs = a.x_bs; r.w = a.w; r.y = a.y; r.y_bs = a.y_bs; r.epi = a.epi;
        r.res1 = a.res1; r.res1_bs = a.res1_bs; r.res2 = a.res2; r.res2_bs = a.res2_bs; r.y2 = a.y2; r.y2_bs = a.y2_bs;
        r.Cin = a.Cin; r.Cout = a.Cout; r.y_fmt = a.y_fmt; r.act = a.act;
        r.slope = a.slope; r.alpha1 = a.alpha1; r.alpha2 = a.alpha2; r.acc_scale = a.acc_scale;
        r.groups = (a.Cout + 31) / 32; r.nchunk = a.Cin / 16;
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
    return (long long)hd->tiles_x * hd->tiles_y * hd->B + 2 + 8 * 32;    // one counter per tile + the launch's give-up word + the heads of its item queues (one per XCD, 128 bytes apart)
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
    int defer_ok = !(tune & 0x10000);                                    // bit 16 of tune / BFSR_CHAIN_DEFER=0: publish every item right after its epilogue (A/B)
    if (const char* e = std::getenv("BFSR_CHAIN_DEFER")) defer_ok = e[0] != '0';
    // how items reach workgroups: 1 static round-robin deal (rounds 5 / 6a), 2 one queue per XCD (default), 3 one global queue; bits 17-18 of tune or
    // BFSR_CHAIN_DEAL=static|xcd|global (measurements)
    int deal = (tune >> 17) & 3;
    if (deal == 0) {
        const char* e = std::getenv("BFSR_CHAIN_DEAL");
        deal = e && e[0] == 's' ? 1 : (e && e[0] == 'g' ? 3 : 2);
    }
    tune &= 0xffff;
    if (tune > 0 && tune < cus) cus = tune;
    const int grid = hd->nitems < cus ? hd->nitems : cus;
    static std::atomic<unsigned long long> lds_done{0};
    if (bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_chain_kernel), LDS_TOTAL, lds_done) != 0) return -1;
    const long long tiles = (long long)hd->tiles_x * hd->tiles_y * hd->B;
    const long long words = tiles + 2 + 8 * 32;
    unsigned* giveup = progress + tiles;
    if ((grid & 7) && deal == 2) deal = 1;                               // the per-XCD queues need the same number of workgroups on every XCD
    unsigned* queue = deal == 1 || hd->nconv == 1 ? nullptr : progress + (tiles + 2);
    const int qcpx = deal == 2 ? grid / 8 : 0;
    if (hd->nconv == 1) progress = nullptr;                              // a chain of one has no dependencies: no counters, no memset node (and nothing polls the give-up word)
    else if (hipMemsetAsync(progress, 0, (size_t)words * sizeof(unsigned), st) != hipSuccess) return -1;
    const ChainRec* recs = reinterpret_cast<const ChainRec*>(static_cast<const unsigned char*>(table_dev) + sizeof(ChainHeader));
    hipLaunchKernelGGL(conv_chain_kernel, dim3((unsigned)grid), dim3((NW + NLW) * 64), LDS_TOTAL, st, recs, hd->B, hd->H, hd->W, hd->tiles_x, hd->tiles_y,
                       hd->nitems, progress, giveup, queue, qcpx, status, defer_ok);
    return (int)hipGetLastError();
}
