// conv_x3s.hip -- 3x3 'same' conv on the bf16 matrix pipe at fp32 accuracy, operating on activations that are STORED as the exact
// 3-term bf16 split ("x3 tensors"), for the dense blocks of the RRDB encoder (RRDBNet_arch.py:25-65, LINF-LP/models/rrdb.py:38-76).
//
// Why a second kernel next to conv_bf16x3.hip (same arithmetic, same packed weights): PMC on the RDB shapes shows the matrix pipe
// of conv_bf16x3_kernel only 57-64 % busy (profiles/r02_a_pmc_rdb.txt) -- per 16-channel chunk every wave stalls on two barriers
// around a register-staged fp32 -> 3 x bf16 split (global load -> VALU -> ds_write), and the short K loops (4-12 chunks) pay a
// cold prologue per tile.  Here the producer's epilogue writes the split once, so tiles are staged with LDS-DMA
// (`buffer_load ... lds`: no registers, no VALU, no ds_write) by TWO DEDICATED LOADER WAVES, the LDS stage is double-buffered
// (ONE barrier per chunk, the next chunk's DMA in flight under this chunk's MFMAs), the eight compute waves issue nothing but
// ds_read_b128 + MFMA, and the workgroup is persistent (one per CU walks its tiles in an XCD-aware order; the loaders run ahead
// across tile boundaries, so a tile's first chunk lands under the previous tile's epilogue).
// Measured on MI355X at the RDB shapes (tools/exp/x3s_abl.py; current variants in profiles/r02_b_x3s_ablation.txt, the staging
// variants were measured while the kernel was built and are quoted in DESIGN.md section 5): the bare fragment-read + MFMA loop of
// this tile shape tops out at 190-245 TFLOP/s; DMA issued by the compute waves costs 20-25 % of that (an LDS-DMA instruction
// blocks its wave's issue for 60-185 cycles), dedicated loader waves give most of it back: 139-202 TFLOP/s at B=8 x 160x160 and
// 185-220 at B=32, against 120-156 / 154-179 for conv_bf16x3_kernel.  A producer/consumer variant that kept fp32 tensors (producer
// waves doing the split, tools/exp/kernels/conv_bf16x3_ps.hip) measured NO gain over conv_bf16x3_kernel: the split's ds_write
// traffic and VALU issue compete with the consumers either way; only removing them (x3 tensors + DMA) pays.  The epilogue is
// 6-11 % of the kernel; a second accumulator chain changes nothing (the single chain is not what idles the pipe).
//
// x3 tensor layout: [B][C/8][3 planes h,m,l][H][W][8] bf16, x = h + m + l EXACTLY (8+8+8 significant bits: a lossless 48-bit
// encoding of an fp32 value); a 16-byte unit = 8 consecutive channels of one pixel = half of an MFMA B operand, so a 64-lane
// LDS-DMA instruction moves 64 consecutive tile positions of one (octet, plane) image.  Channel slices at multiples of 8 are
// views (pointer offset), which is what the dense block needs (64 | 32 | 32 | 32 | 32 channels in one buffer).
//
// GEMM view per workgroup: M = 32 output channels, N = 8 rows x 32 pixels (wave w owns row w), K = 16 channels per chunk x 9 taps,
// six bf16 MFMAs (32x32x16) per operand pair.  LDS stage = input [3][2 k-halves][384 positions][8] (10 x 34 tile + padding to a
// multiple of 64 positions) + weights [3][9 taps][2][32][8]  =  36 864 + 27 648 B; two stages = 129 024 B -> one workgroup per CU.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/bfsr_hip.h"
#include "launch_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

namespace {

constexpr int NW = 8, TH = 8, PW = 34, NPOS = (TH + 2) * PW, NPOSP = 384, NG = NPOSP / 64;
constexpr int SUB = NPOSP * 16;                 // bytes of one (plane, k half) sub-image
constexpr int IN_BYTES = 6 * SUB;               // 36 864
constexpr int WPL = 9 * 2 * 32 * 16;            // bytes of one weight plane per chunk
constexpr int W_BYTES = 3 * WPL;                // 27 648
constexpr int STAGE = IN_BYTES + W_BYTES;
constexpr int LDS_TOTAL = 2 * STAGE;            // 129 024
constexpr int LDS_PARK = NW * 16 * 64 * 4;      // + 32 768: the deferred epilogue's parked accumulators (one [16][64] fp32 block per compute wave)
constexpr int W_INSTR = W_BYTES / 1024;         // 27 LDS-DMA wave-instructions per weight slab
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l)
{
    h = (__bf16)v;
    const float r1 = v - (float)h;        // exact
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);          // exact residual, <= 8 significant bits
}

struct Item { int cg, b, x0, y0, th; };     // th = tile height: TH, or TH/2 for the split tiles of the last partial round

// ABL: the product is ABL = 16: dedicated loader waves (2 + ((ABL >> 5) & 3) of them, waves 8..) issue every LDS-DMA piece.
// The other values are the ablation switches of tools/exp/x3s_abl.py (built with -DBFSR_X3S_ABL only): 0 = the compute waves
// issue the DMA themselves right after the barrier, 8 = one piece per tap between the MFMAs, 1 = no DMA after the first stage
// (timing only), 2 = no barrier (timing only), 4 = two accumulator chains
template <int ABL>
__global__ __launch_bounds__((NW + ((ABL & 16) ? 2 + ((ABL >> 5) & 3) : 0)) * 64, 1) void conv3x3_x3s_kernel(BfsrConvX3Args p, int tiles_x, int tiles_y, int groups, int nitems, int n_full, int waitvm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int G = gridDim.x;
    // XCD-aware slot: the workgroups of one XCD (blockIdx % 8) take consecutive work items, so the cout groups of one tile and
    // vertically adjacent tiles (shared halo rows) are staged through the same L2 at about the same time
    const int slot = (int)bfsr::xcd_order(blockIdx.x, (unsigned)G);
    const int H = p.H, W = p.W;
    const unsigned HW16 = (unsigned)(H * W) * 16u;           // bytes of one (octet, plane) image
    const int nchunk = p.Cin >> 4;

    // items [0, n_full) are whole tiles; the tiles of the last partial round of persistent workgroups follow as two half-height
    // items each (rows 0-3 / 4-7, computed by waves 0-3): 800 tiles on 256 CUs take 3.5 rounds instead of 4
    auto decode = [&](int it_) {
        Item r;
        const int half = it_ >= n_full ? (it_ - n_full) & 1 : -1;
        const int it = it_ >= n_full ? n_full + ((it_ - n_full) >> 1) : it_;
        r.cg = it % groups; int t = it / groups;
        const int ty = t % tiles_y; t /= tiles_y;
        r.x0 = (t % tiles_x) * 32; r.y0 = ty * TH; r.b = t / tiles_x;
        r.th = TH;
        if (half >= 0) { r.th = TH / 2; r.y0 += half * (TH / 2); }
        return r;
    };

    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w), 0,
                                                                          (unsigned)((long long)groups * nchunk * W_BYTES), 0x00020000);
    // ---- staging: waves 0-5 own one 64-position group of the input tile (6 sub-images each) + 2 weight pieces, waves 6-7 the
    // remaining 15 weight pieces: ~8 LDS-DMA instructions per wave and chunk
    unsigned voff_in = OOB;                      // this lane's byte offset inside an (octet, plane) image, for the staged item
    __amdgpu_buffer_rsrc_t rs_in_stage;
    int stage_cg = 0;
    auto stage_setup = [&](const Item& it) {
        const unsigned short* xb = p.x + (long long)it.b * p.x_bs;
        rs_in_stage = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xb), 0, (unsigned)(p.Cin >> 3) * 3u * HW16, 0x00020000);
        stage_cg = it.cg;
        if (wave < NG) {
            const int pos = wave * 64 + lane;
            const int r = pos / PW, c = pos - r * PW;
            const int gy = it.y0 + r - 1, gx = it.x0 + c - 1;
            const bool ok = pos < (it.th + 2) * PW && gy >= 0 && gy < H && gx >= 0 && gx < W;
            voff_in = ok ? (unsigned)(gy * W + gx) * 16u : OOB;
        }
    };
    // piece j = 0..7 of a stage: waves 0-5: j < 6 -> input sub-image j (plane j/2, k half j&1) of their position group, j = 6,7 ->
    // weight pieces 2*wave, 2*wave+1; waves 6-7: weight pieces 12 + (wave-6) + 2j
    auto stage_piece = [&](int j, int k, int buf) {
        unsigned char* base = smem + buf * STAGE;
        const unsigned wsoff = (unsigned)(stage_cg * nchunk + k) * (unsigned)W_BYTES;
        const unsigned wv = (unsigned)lane * 16u;
        if (wave < NG && j < 6) {
            const unsigned soff = (unsigned)((2 * k + (j & 1)) * 3 + (j >> 1)) * HW16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in_stage, (lds_void*)(base + j * SUB + wave * 1024), 16, voff_in, soff, 0, 0);
        } else {
            const int piece = wave < NG ? wave * 2 + (j - 6) : 2 * NG + (wave - NG) + 2 * j;
            if (piece < W_INSTR)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(base + IN_BYTES + piece * 1024), 16, wv + (unsigned)piece * 1024u, wsoff, 0, 0);
        }
    };
    auto stage = [&](int k, int buf) {
#pragma unroll
        for (int j = 0; j < 8; ++j) stage_piece(j, k, buf);
    };

    int it = slot;
    if (it >= nitems) return;
    Item cur = decode(it);
    if constexpr ((ABL & 256) != 0) { if (wave < NW) __builtin_amdgcn_s_setprio(1); }       // experiment: compute waves win the arbitration
    if constexpr ((ABL & 512) != 0) { if (wave >= NW) __builtin_amdgcn_s_setprio(1); }      // experiment: loader waves win
    if constexpr ((ABL & 16) != 0) {
        if (wave >= NW) {
            // ---- loader waves: all 63 pieces of every stage (loader 0: input groups 0-2 + weight pieces 0-12, loader 1: groups 3-5 + 13-26)
            constexpr int NL = 2 + ((ABL >> 5) & 3);
            const int ld = wave - NW;
            unsigned vg[NG];
            auto lsetup = [&](const Item& itx) {
                const unsigned short* xb = p.x + (long long)itx.b * p.x_bs;
                rs_in_stage = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xb), 0, (unsigned)(p.Cin >> 3) * 3u * HW16, 0x00020000);
                stage_cg = itx.cg;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int pos = g * 64 + lane;
                    const int r = pos / PW, c = pos - r * PW;
                    const int gy = itx.y0 + r - 1, gx = itx.x0 + c - 1;
                    const bool ok = pos < (itx.th + 2) * PW && gy >= 0 && gy < H && gx >= 0 && gx < W;
                    vg[g] = ok ? (unsigned)(gy * W + gx) * 16u : OOB;
                }
            };
            // piece i = 0..62 (36 input pieces: group i/6, sub-image i%6; then 27 weight pieces) goes to loader i % NL
            auto lstage = [&](int k, int buf_) {
                unsigned char* base = smem + buf_ * STAGE;
                const unsigned wsoff = (unsigned)(stage_cg * nchunk + k) * (unsigned)W_BYTES;
                const unsigned wv = (unsigned)lane * 16u;
#pragma unroll
                for (int i = 0; i < 36 + W_INSTR; ++i) {
                    if (i % NL != ld) continue;
                    if (i < 36) {
                        const int g = i / 6, s_ = i % 6;
                        const unsigned soff = (unsigned)((2 * k + (s_ & 1)) * 3 + (s_ >> 1)) * HW16;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in_stage, (lds_void*)(base + s_ * SUB + g * 1024), 16, vg[g], soff, 0, 0);
                    } else {
                        const int piece = i - 36;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(base + IN_BYTES + piece * 1024), 16, wv + (unsigned)piece * 1024u, wsoff, 0, 0);
                    }
                }
            };
            lsetup(cur);
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
    } else {
        stage_setup(cur);
        stage(0, 0);
    }
    int buf = 0;

    // ---- DEFERRED EPILOGUE (ABL & 1024; NOT the product -- a measured negative result, profiles/r03_x3s_deferred_epilogue.txt): the
    // epilogue of item i runs in SLICES between the taps of item i+1's first chunks, where its VALU / LDS-crossbar / memory
    // instructions were meant to issue beside the matrix pipe instead of behind it (with the accumulators kept alive the epilogue
    // is 9-15 % of the kernel, tools/exp/x3s_abl.py "no epilogue").  Result: one RDB 695 us vs 577 us at 8 x 160^2, 3008 vs 2571 us at
    // 16 x 256^2 -- the compute waves are issue / latency bound (matrix pipe ~53 % busy), so instructions added between the MFMA
    // groups of every wave lengthen the barrier-synchronised chunk instead of filling idle slots.  The 16 results per
    // lane wait in a per-wave LDS scratch (registers: the kernel sits at the 168-VGPR limit of three waves per SIMD), already
    // paired into channel octets.  Phase = chunk index of the next item; per octet q the stages are
    //   A (only with res1): slot 0 loads r1, slots 1-8 parameters + activation of one channel each (value back to the scratch)
    //   B (only with res2): slots 0-7 add r1, slot 8 loads r2
    //   C: slots 0-7 one channel each: (activation | + r1 | + r2), re-encoding; slot 8: the 16-byte stores (fp32 output: per channel)
    constexpr bool DEF = (ABL & 1024) != 0;
    bool pend = false;
    Item pit = {0, 0, 0, 0, TH};
    float4 ppm = make_float4(0.f, 0.f, 1.f, 0.f);
    bool pvalid[2] = {false, false}, pbias_only = true, pfast = true;
    long long ppix = 0;
    bf16x8 rr[3], ph8, pm8, pl8;
    float* sP = reinterpret_cast<float*>(smem + LDS_TOTAL) + (wave < NW ? wave : 0) * 16 * 64 + lane;      // [16 values][64 lanes] of this wave
    const float eslope = p.act == BFSR_ACT_NONE ? 1.f : (p.act == BFSR_ACT_RELU ? 0.f : p.slope);
    const long long HWl = (long long)H * W;
    const int nstage = 1 + (p.res1 ? 1 : 0) + (p.res2 ? 1 : 0);   // stages per octet: C | A C | A B C
    auto pfetch = [&](float val, int src_lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane * 4, __float_as_int(val))); };
    auto activate = [&](float v, int q, int j, int lh) {
        const int src = lh * 16 + (q * 16 + j) * 2;              // lane holding this channel's first float4
        const float e0 = pfetch(ppm.x, src);
        if (pfast) {
            const float u = v + e0;
            return fmaxf(u, u * eslope);                          // = u > 0 ? u : u*slope for 0 <= slope <= 1
        }
        float e1 = 0.f, e2 = 1.f, e3 = 0.f, e4 = 1.f;
        if (!pbias_only) { e1 = pfetch(ppm.y, src); e2 = pfetch(ppm.z, src); e3 = pfetch(ppm.w, src); e4 = pfetch(ppm.x, src + 1); }
        float u = v + e0;
        u = (u + e1) * e2 + e3;
        u = u > 0.f ? u : u * eslope;
        return u * e4;
    };
    auto load_res = [&](const unsigned short* base, long long bs, int q, int lh) {
        const int oct = pit.cg * 4 + q * 2 + lh;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int e = 0; e < 8; ++e) rr[pl][e] = (__bf16)0.f;
        if (pvalid[q]) {
            const unsigned short* rb = base + (long long)pit.b * bs + ((long long)oct * 3 * HWl + ppix) * 8;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) rr[pl] = *reinterpret_cast<const bf16x8*>(rb + pl * HWl * 8);
        }
    };
    // stage: 0 = A, 1 = B, 2 = C (wave-uniform runtime values); q, t: constants after inlining / unrolling.  The octet being
    // processed lives in registers (pvq) from its first phase on; slot 0 of that phase also fetches the eight bias values, so the
    // channel slots are VALU only (a parked-value read + ds_bpermute per slot left their LDS latencies exposed between the taps:
    // the first version of this scheme was 15-35 % SLOWER than the epilogue behind the K loop)
    float pvq[8], pe0[8];
    auto first_slot = [&](int q, int lh) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pvq[j] = sP[(q * 8 + j) * 64];
#pragma unroll
        for (int j = 0; j < 8; ++j) pe0[j] = pfetch(ppm.x, lh * 16 + (q * 16 + j) * 2);
    };
    auto act_j = [&](int q, int j, int lh) {
        if (pfast) {
            const float u = pvq[j] + pe0[j];
            pvq[j] = fmaxf(u, u * eslope);                        // = u > 0 ? u : u*slope for 0 <= slope <= 1
        } else {
            const int src = lh * 16 + (q * 16 + j) * 2;           // lane holding this channel's first float4
            float e1 = 0.f, e2 = 1.f, e3 = 0.f, e4 = 1.f;
            if (!pbias_only) { e1 = pfetch(ppm.y, src); e2 = pfetch(ppm.z, src); e3 = pfetch(ppm.w, src); e4 = pfetch(ppm.x, src + 1); }
            float u = pvq[j] + pe0[j];
            u = (u + e1) * e2 + e3;
            u = u > 0.f ? u : u * eslope;
            pvq[j] = u * e4;
        }
    };
    auto slice_q = [&](int q, int stage, int t, int lh) {
        if (stage == 0) {                                           // A: first phase of an octet with residuals
            if (t == 0) { first_slot(q, lh); load_res(p.res1, p.res1_bs, q, lh); }
            else act_j(q, t - 1, lh);
        } else if (stage == 1) {                                    // B: + r1, then fetch r2
            if (t < 8) pvq[t] = p.alpha1 * pvq[t] + (((float)rr[0][t] + (float)rr[1][t]) + (float)rr[2][t]);
            else load_res(p.res2, p.res2_bs, q, lh);
        } else {                                                    // C: last phase of the octet
            const int oct = pit.cg * 4 + q * 2 + lh;
            if (t == 0) {
                if (nstage == 1) first_slot(q, lh);
            } else {
                const int j = t - 1;
                if (nstage == 1) act_j(q, j, lh);
                else if (nstage == 2) pvq[j] = p.alpha1 * pvq[j] + (((float)rr[0][j] + (float)rr[1][j]) + (float)rr[2][j]);
                else pvq[j] = p.alpha2 * pvq[j] + (((float)rr[0][j] + (float)rr[1][j]) + (float)rr[2][j]);
                if (p.y_fmt == 1) {
                    __bf16 h, m, l;
                    split3(pvq[j], h, m, l);
                    ph8[j] = h; pm8[j] = m; pl8[j] = l;
                } else if (pvalid[q] && oct * 8 + j < p.Cout) {
                    (reinterpret_cast<float*>(p.y) + (long long)pit.b * p.y_bs + ppix)[(long long)(oct * 8 + j) * HWl] = pvq[j];
                }
                if (t == 8 && p.y_fmt == 1 && pvalid[q]) {
                    unsigned short* yb = reinterpret_cast<unsigned short*>(p.y) + (long long)pit.b * p.y_bs + ((long long)oct * 3 * HWl + ppix) * 8;
                    *reinterpret_cast<bf16x8*>(yb) = ph8;
                    *reinterpret_cast<bf16x8*>(yb + HWl * 8) = pm8;
                    *reinterpret_cast<bf16x8*>(yb + 2 * HWl * 8) = pl8;
                }
            }
        }
    };
    // `lh` = lane >> 5 derived from an OPAQUE copy of the lane id per chunk: everything per-lane the slices need (ds_bpermute
    // addresses, octet numbers) would otherwise be loop-invariant, hoisted out of the item loop by hipcc and spilled
    auto slice = [&](int phase, int t, int lh) {                    // phase: wave-uniform, t: constant after unrolling
        if (!pend || phase >= 2 * nstage) return;
        const int q = phase >= nstage ? 1 : 0;
        const int pl_ = phase - q * nstage;                         // stages of an octet: C | A C | A B C
        const int stage = pl_ == nstage - 1 ? 2 : (pl_ == 0 ? 0 : 1);
        if (q == 0) slice_q(0, stage, t, lh); else slice_q(1, stage, t, lh);
    };
    auto phase_all = [&](int phase) {
        int lo = lane;
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (int t = 0; t < 9; ++t) slice(phase, t, lo >> 5);
    };

    while (true) {
        f32x16 acc, acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
        // Per-channel epilogue parameters of this cout group: ONE coalesced load per lane, issued here so that it lands under the
        // K loop (lane L holds float4 #L of the group's [32][8] block: channel L>>1; bias/shift/scale/post if L is even, post_scale
        // if odd); the epilogue hands them to the lanes that need them with ds_bpermute.  Reading epi[co] directly costs 32
        // broadcast loads of 1 KiB per wave and tile, issued after the last MFMA: measured (tools/exp/x3s_abl.py), the epilogue
        // then took as long as the K loop of a 64-channel conv.
        float4 pm = (lane & 1) ? make_float4(1.f, 0.f, 0.f, 0.f) : make_float4(0.f, 0.f, 1.f, 0.f);
        {
            const int idx = cur.cg * 64 + lane;
            if (p.epi && (idx >> 1) < p.Cout) pm = reinterpret_cast<const float4*>(p.epi)[idx];
        }
        // ... and so are the operands of the first residual (`x5*0.2 + x`, 2 octets x 3 planes): the K loop hides their round trip
        const int gy = cur.y0 + wave, gx = cur.x0 + l31;
        const long long HW = (long long)H * W;
        const long long pix = (long long)gy * W + gx;
        bool valid[2];
        bf16x8 r1[2][3];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int oct = cur.cg * 4 + q * 2 + lhi;             // channel octet of the output tensor
            valid[q] = wave < cur.th && gy < H && gx < W && oct * 8 < p.Cout;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int j = 0; j < 8; ++j) r1[q][pl][j] = (__bf16)0.f;
            if (!DEF && p.res1 && valid[q]) {
                const unsigned short* rb = p.res1 + (long long)cur.b * p.res1_bs + ((long long)oct * 3 * HW + pix) * 8;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) r1[q][pl] = *reinterpret_cast<const bf16x8*>(rb + pl * HW * 8);
            }
        }
        const int nxt = it + G;
        const bool has_next = nxt < nitems;
        Item nitem = cur;
        for (int k = 0; k < nchunk; ++k) {
            // this wave's DMA pieces of stage `buf` have landed (with dedicated loader waves a compute wave has none: the wait would
            // only drain its own epilogue stores and residual loads at every chunk boundary) ...
            // ... which measures both ways (same-run A/B, tools/exp/x3s_abl.py): without the wait one RDB takes 559 instead of 579 us
            // at 8 x 160^2 (3 rounds of tiles per CU), with it 2556-2634 instead of 2632-2701 us at 16 x 256^2 (16 rounds: draining
            // the stores throttles the compute waves' traffic against the loaders' DMA) -> `waitvm` is set by the launcher for long runs
            if (!(ABL & 16) || (ABL & 2048) || waitvm) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(ABL & 2)) __builtin_amdgcn_s_barrier();         // ... and so have everybody else's; stage buf^1 is free again
            int sk = -1;                                          // chunk to stage into buf^1 (-1: nothing)
            if (k + 1 < nchunk) sk = k + 1;
            else if (has_next) { nitem = decode(nxt); stage_setup(nitem); sk = 0; }
            if (ABL & (1 | 16)) sk = -1;
            if (!(ABL & 8) && sk >= 0) stage(sk, buf ^ 1);

            const unsigned char* sIn = smem + buf * STAGE;
            const unsigned char* sW = sIn + IN_BYTES;
            const unsigned char* inB = sIn + (lhi * NPOSP + wave * PW + l31) * 16;        // plane stride 2*SUB
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
            if (wave >= cur.th) {                                  // upper waves of a half-height tile: barriers (+ their pending epilogue) only
                if constexpr (DEF) phase_all(k);
                buf ^= 1;
                continue;
            }
            int plane_ = lane;
            if constexpr (DEF) asm volatile("" : "+v"(plane_));
            const int plh = plane_ >> 5;
            load_b(0);
            load_a(0, 0, 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dx = t / 3, dy = t % 3, ab = t & 1;
                if (t > 0 && dy == 0) load_b(dx);
                if (t + 1 < 9) load_a(ab ^ 1, (t + 1) / 3, (t + 1) % 3);
                if ((ABL & 8) && t < 8 && sk >= 0) stage_piece(t, sk, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#define BFSR_TERM(ACC_, PA_, PB_) ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ab][PA_], bfr[PB_][dy], ACC_, 0, 0, 0);
                if ((ABL & 4) && (t & 1)) {
                    BFSR_TERM(acc2, 2, 0) BFSR_TERM(acc2, 0, 2) BFSR_TERM(acc2, 1, 1) BFSR_TERM(acc2, 1, 0) BFSR_TERM(acc2, 0, 1) BFSR_TERM(acc2, 0, 0)
                } else {
                    BFSR_TERM(acc, 2, 0) BFSR_TERM(acc, 0, 2) BFSR_TERM(acc, 1, 1) BFSR_TERM(acc, 1, 0) BFSR_TERM(acc, 0, 1) BFSR_TERM(acc, 0, 0)
                }
#undef BFSR_TERM
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (DEF) { slice(k, t, plh); __builtin_amdgcn_sched_barrier(0); }  // a slice of the PREVIOUS item's epilogue
            }
            buf ^= 1;
        }
        if (ABL & 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
        }

        // ---- epilogue of item `cur` (the next item's first chunk is already in flight).  acc[r] = channel (r&3)+8(r>>2)+4*lhi
        // of pixel (y0+wave, x0+l31); v_permlane32_swap pairs the two half-waves so that every lane ends up with two complete
        // channel octets of its pixel: octet q*2+lhi in v[q][0..7]
        if constexpr ((ABL & 128) != 0) {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" :: "v"(acc));
#endif
            if (!has_next) break; it = nxt; cur = nitem; continue; }     // ablation: no epilogue (acc kept alive)
        if constexpr (DEF) {
            for (int k = nchunk; k < 2 * nstage; ++k) phase_all(k);  // fewer chunks than phases (the 64-channel trunk_conv with residual)
            asm volatile("s_nop 11" ::: "memory");                 // MFMA result -> VALU read inside the asm below
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float lo = acc[8 * q + i], hi = acc[8 * q + 4 + i];
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
                    sP[(q * 8 + i) * 64] = lo;                      // octet q*2+lhi of this lane's pixel, channels i and 4+i
                    sP[(q * 8 + 4 + i) * 64] = hi;
                }
            pend = true; pit = cur; ppm = pm; ppix = pix;
            pvalid[0] = valid[0]; pvalid[1] = valid[1];
            pbias_only = __all((lane & 1) ? pm.x == 1.f : (pm.y == 0.f && pm.z == 1.f && pm.w == 0.f));
            pfast = pbias_only && eslope >= 0.f && eslope <= 1.f;
            if (!has_next) {
#pragma unroll 1
                for (int k = 0; k < 2 * nstage; ++k) phase_all(k);   // the last item's epilogue has no K loop to hide under
                break;
            }
            it = nxt;
            cur = nitem;
            continue;
        }
        float v[2][8];
        // inline asm: hipcc (ROCm 7.2) folds eight __builtin_amdgcn_permlane32_swap calls on MFMA result elements into ONE
        // swap of element 0 (every output channel became channel 0); asm statements are opaque to it.  The compiler pads
        // neither the MFMA -> VALU-read hazard (12 wait states for an 8-pass MFMA) nor the VALU-write -> permlane hazard
        // (2 states) around an asm statement, so both pads are inside the strings.
        asm volatile("s_nop 11" ::: "memory");
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float lo = acc[8 * q + i], hi = acc[8 * q + 4 + i];
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
                v[q][i] = lo;
                v[q][4 + i] = hi;
            }
        const bool bias_only = __all((lane & 1) ? pm.x == 1.f : (pm.y == 0.f && pm.z == 1.f && pm.w == 0.f));
        auto fetch = [&](float val, int src_lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane * 4, __float_as_int(val))); };
        const float slope = p.act == BFSR_ACT_NONE ? 1.f : (p.act == BFSR_ACT_RELU ? 0.f : p.slope);
        const bool fast = bias_only && slope >= 0.f && slope <= 1.f;
        float o[2][8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float e0[8], e1[8], e2[8], e3[8], e4[8];              // all lanes take part in the exchange (before any divergence)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int src = ((q * 2 + lhi) * 8 + j) * 2;      // lane holding this channel's first float4
                e0[j] = fetch(pm.x, src);
                e1[j] = 0.f; e2[j] = 1.f; e3[j] = 0.f; e4[j] = 1.f;
                if (!bias_only) { e1[j] = fetch(pm.y, src); e2[j] = fetch(pm.z, src); e3[j] = fetch(pm.w, src); e4[j] = fetch(pm.x, src + 1); }
            }
            if (fast) {                                           // bias + (leaky) ReLU only: 3 VALU ops per channel instead of 8
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float u = v[q][j] + e0[j];
                    o[q][j] = fmaxf(u, u * slope);                // = u > 0 ? u : u*slope for 0 <= slope <= 1
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float u = v[q][j] + e0[j];
                    u = (u + e1[j]) * e2[j] + e3[j];
                    u = u > 0.f ? u : u * slope;
                    o[q][j] = u * e4[j];
                }
            }
        }
        if (p.res1) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) o[q][j] = p.alpha1 * o[q][j] + (((float)r1[q][0][j] + (float)r1[q][1][j]) + (float)r1[q][2][j]);
        }
        if (p.res2) {                                             // both octets' loads in flight before the first use: one round trip
            bf16x8 r2[2][3];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int oct = cur.cg * 4 + q * 2 + lhi;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int j = 0; j < 8; ++j) r2[q][pl][j] = (__bf16)0.f;
                if (valid[q]) {
                    const unsigned short* rb = p.res2 + (long long)cur.b * p.res2_bs + ((long long)oct * 3 * HW + pix) * 8;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) r2[q][pl] = *reinterpret_cast<const bf16x8*>(rb + pl * HW * 8);
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) o[q][j] = p.alpha2 * o[q][j] + (((float)r2[q][0][j] + (float)r2[q][1][j]) + (float)r2[q][2][j]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!valid[q]) continue;
            const int oct = cur.cg * 4 + q * 2 + lhi;
            if (p.y_fmt == 1) {
                bf16x8 h8, m8, l8;
#pragma unroll
                for (int j = 0; j < 8; ++j) { __bf16 h, m, l; split3(o[q][j], h, m, l); h8[j] = h; m8[j] = m; l8[j] = l; }
                unsigned short* yb = reinterpret_cast<unsigned short*>(p.y) + (long long)cur.b * p.y_bs + ((long long)oct * 3 * HW + pix) * 8;
                *reinterpret_cast<bf16x8*>(yb) = h8;
                *reinterpret_cast<bf16x8*>(yb + HW * 8) = m8;
                *reinterpret_cast<bf16x8*>(yb + 2 * HW * 8) = l8;
            } else {
                float* yb = reinterpret_cast<float*>(p.y) + (long long)cur.b * p.y_bs + pix;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (oct * 8 + j < p.Cout) yb[(long long)(oct * 8 + j) * HW] = o[q][j];
            }
        }
        if (!has_next) break;
        it = nxt;
        cur = nitem;
    }
}

// ---- fp32 NCHW view <-> x3 tensor (the boundaries of an x3 region: conv_first / trunk taps) ----------------------------------
__global__ void x3_pack_kernel(const float* __restrict__ x, long long x_bs, unsigned short* __restrict__ y, long long y_bs,
                               int C, long long HW, long long total)
{
    // one thread = one (b, octet, pixel): reads 8 channel planes (coalesced along pixels), writes 3 x 16 B
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int C8 = C >> 3;
    const long long pix = i % HW; const long long t = i / HW;
    const int oct = (int)(t % C8); const int b = (int)(t / C8);
    const float* xb = x + (long long)b * x_bs + (long long)oct * 8 * HW + pix;
    bf16x8 h8, m8, l8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { __bf16 h, m, l; split3(xb[(long long)j * HW], h, m, l); h8[j] = h; m8[j] = m; l8[j] = l; }
    unsigned short* yb = y + (long long)b * y_bs + ((long long)oct * 3 * HW + pix) * 8;
    *reinterpret_cast<bf16x8*>(yb) = h8;
    *reinterpret_cast<bf16x8*>(yb + HW * 8) = m8;
    *reinterpret_cast<bf16x8*>(yb + 2 * HW * 8) = l8;
}

__global__ void x3_unpack_kernel(const unsigned short* __restrict__ x, long long x_bs, float* __restrict__ y, long long y_bs,
                                 int C, long long HW, long long total)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int C8 = C >> 3;
    const long long pix = i % HW; const long long t = i / HW;
    const int oct = (int)(t % C8); const int b = (int)(t / C8);
    const unsigned short* xb = x + (long long)b * x_bs + ((long long)oct * 3 * HW + pix) * 8;
    const bf16x8 h = *reinterpret_cast<const bf16x8*>(xb);
    const bf16x8 m = *reinterpret_cast<const bf16x8*>(xb + HW * 8);
    const bf16x8 l = *reinterpret_cast<const bf16x8*>(xb + 2 * HW * 8);
    float* yb = y + (long long)b * y_bs + (long long)oct * 8 * HW + pix;
#pragma unroll
    for (int j = 0; j < 8; ++j) yb[(long long)j * HW] = ((float)h[j] + (float)m[j]) + (float)l[j];
}

}  // namespace

extern "C" int bfsr_conv3x3_x3s(const BfsrConvX3Args* a, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || !a->x || !a->w || !a->y) return -1;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->Cin <= 0 || (a->Cin & 15) || a->Cout <= 0) return -1;
    if (a->y_fmt != 0 && a->y_fmt != 1) return -1;
    if ((a->y_fmt == 1 || a->res1 || a->res2) && (a->Cout & 7)) return -1;
    // 32-bit byte offsets inside one batch item of the input, and of one weight tensor
    if ((long long)(a->Cin / 8) * 3 * a->H * a->W * 16 >= (1LL << 31)) return -1;
    if ((reinterpret_cast<unsigned long long>(a->x) & 15) || (a->x_bs & 7)) return -1;
    if (a->y_fmt == 1 && ((reinterpret_cast<unsigned long long>(a->y) & 15) || (a->y_bs & 7))) return -1;
    if (a->res1 && ((reinterpret_cast<unsigned long long>(a->res1) & 15) || (a->res1_bs & 7))) return -1;
    if (a->res2 && ((reinterpret_cast<unsigned long long>(a->res2) & 15) || (a->res2_bs & 7))) return -1;
    const int tiles_x = (a->W + 31) / 32, tiles_y = (a->H + TH - 1) / TH;
    const int groups = (a->Cout + 31) / 32;
    const long long nitems = (long long)tiles_x * tiles_y * groups * a->B;
    if (nitems > 0x7fffffffLL) return -1;
    if ((long long)groups * (a->Cin / 16) * W_BYTES >= (1LL << 32)) return -1;
    int cus = bfsr::cu_count();                             // cached per device; no silent default
    if (cus <= 0) return -1;
    if (a->tune > 0) cus = a->tune;
    long long grid = nitems < cus ? nitems : cus;          // one persistent workgroup per CU
    long long n_full = nitems, n_items = nitems;
    const long long rem = nitems % grid;
    if (nitems > grid && rem > 0 && 2 * rem <= grid) { n_full = nitems - rem; n_items = n_full + 2 * rem; }     // split the last partial round
    if (n_items > 0x7fffffffLL) return -1;
    const int waitvm = (a->tune != -16 && nitems >= 8 * grid * groups) ? 1 : 0;        // >= 8 rounds of tiles per CU (tune -16: never, for A/B)
#define BFSR_LAUNCH(ABL_)                                                                                                            \
    {                                                                                                                                \
        static std::atomic<unsigned long long> lds_done{0};                                                                          \
        constexpr int LDS_ = LDS_TOTAL + (((ABL_) & 1024) ? LDS_PARK : 0);                                                           \
        if (bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&conv3x3_x3s_kernel<ABL_>), LDS_, lds_done) != 0) return -1;      \
        hipLaunchKernelGGL(conv3x3_x3s_kernel<ABL_>, dim3((unsigned)grid), dim3((NW + (((ABL_) & 16) ? 2 + (((ABL_) >> 5) & 3) : 0)) * 64), LDS_, st, *a, tiles_x, tiles_y, groups, \
                           (int)n_items, (int)n_full, waitvm);                                                                      \
        return (int)hipGetLastError();                                                                                               \
    }
#ifdef BFSR_X3S_ABL
    switch (a->tune < 0 ? -a->tune : 0) {
        case 1: BFSR_LAUNCH(1) case 3: BFSR_LAUNCH(3) case 4: BFSR_LAUNCH(4) case 5: BFSR_LAUNCH(5) case 7: BFSR_LAUNCH(7)
        case 8: BFSR_LAUNCH(8) case 12: BFSR_LAUNCH(12) case 16: BFSR_LAUNCH(16) case 48: BFSR_LAUNCH(48) case 80: BFSR_LAUNCH(80)
        case 272: BFSR_LAUNCH(272) case 528: BFSR_LAUNCH(528) case 1040: BFSR_LAUNCH(1040) case 2064: BFSR_LAUNCH(2064) case 17: BFSR_LAUNCH(17) case 144: BFSR_LAUNCH(144) case 145: BFSR_LAUNCH(145) case 20: BFSR_LAUNCH(20) case 148: BFSR_LAUNCH(148)
        default: break;
    }
#endif
    if (a->tune == -1040) BFSR_LAUNCH(1040)                       // 16 | 1024: + deferred epilogue (measured 17-20 % slower, see the kernel)
    BFSR_LAUNCH(16)                                               // dedicated loader waves, epilogue behind the K loop
#undef BFSR_LAUNCH
}

extern "C" int bfsr_x3_pack(const float* x, long long x_bs, unsigned short* y, long long y_bs, int B, int C, int H, int W, void* stream)
{
    if (!x || !y || B <= 0 || C <= 0 || (C & 7) || H <= 0 || W <= 0) return -1;
    if ((reinterpret_cast<unsigned long long>(y) & 15) || (y_bs & 7)) return -1;
    const long long HW = (long long)H * W, total = (long long)B * (C / 8) * HW;
    hipLaunchKernelGGL(x3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, y, y_bs, C, HW, total);
    return (int)hipGetLastError();
}

extern "C" int bfsr_x3_unpack(const unsigned short* x, long long x_bs, float* y, long long y_bs, int B, int C, int H, int W, void* stream)
{
    if (!x || !y || B <= 0 || C <= 0 || (C & 7) || H <= 0 || W <= 0) return -1;
    if ((reinterpret_cast<unsigned long long>(x) & 15) || (x_bs & 7)) return -1;
    const long long HW = (long long)H * W, total = (long long)B * (C / 8) * HW;
    hipLaunchKernelGGL(x3_unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_bs, y, y_bs, C, HW, total);
    return (int)hipGetLastError();
}
