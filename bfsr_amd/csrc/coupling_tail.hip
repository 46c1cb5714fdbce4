// coupling_tail.hip -- the tail of the sequential part of a conditional-affine FlowStep (FlowAffineCouplingsAblation.py:57-97,
// FlowStep.py:113-129), levels with C in {12, 24} flow channels:
//
//   h_aff = (conv3x3(hid; W4) + b4) * exp(3 logs4)        fAffine.4 = Conv2dZeros 64 -> 2*(C - C/2)  (flow.py:68-83)
//   then the step's pointwise chain with h_aff taken from the accumulators (the semantics of bfsr_flow_pointwise):
//     reverse: z2 = z2/scale - shift; z = z/scaleFt - shiftFt; z = Winv z; z = z*exp(-logs) - bias          (this step)
//     forward: z2 = (z2 + shift)*scale   (this step's self-conditional)   then, if given, the NEXT step's head:
//              z = (z + bias)*exp(logs); z = W z; z = (z + shiftFt)*scaleFt
//
// `hid` is the h2 tensor written by coupling_head_kernel ([B][8 octets][2 planes hi,lo][H][W][8] fp16); the conv runs the two-term
// fp16 split on v_mfma_f32_32x32x16_f16 (weights pre-scaled by a power of two, accumulators x acc_scale in the epilogue).
//
// Structure = conv3x3_h2x_kernel's (conv_h2s.hip): 16-row x 32-px tiles, eight compute waves (two rows each) + four LDS-DMA loader
// waves, persistent workgroups in XCD-aware order -- re-cut for a conv with <= 32 output channels, which has too little matrix work
// per staged byte for that kernel's two-stage pipeline (measured round 4: with 16-channel chunks a chunk is ~1.1 us of MFMA time but
// an L2 -> LDS round trip under load is 2-3 us, so every chunk waited for its DMA: 130 us per level-1 launch whatever the MFMA count):
//   * a chunk is ONE channel octet (two planes): the MFMA's K = 16 spans two TAPS x 8 channels (lanes 0-31 read tap 2s, lanes 32-63 tap
//     2s+1 of the same octet; the tenth tap has zero weights), five steps per chunk, eight chunks per item;
//   * the input ring has FOUR stages of 20 KiB: the loaders run three chunks (~2 us of compute) ahead, across item boundaries;
//   * the whole weight tensor (72.5 KiB) stays resident in LDS: with one output-channel group it is the same for every item;
//   * <= 16 output channels (C = 12): the three products of the split need only TWO matrix instructions per operand pair -- the idle
//     rows 16-31 of the 32-row tile carry the lo plane of the weights:  acc += [w_hi | w_lo] . x_hi;  acc += [w_hi | 0] . x_lo  leaves
//     w_hi.x_hi + w_hi.x_lo in rows 0-15 and w_lo.x_hi in rows 16-31 = accumulator registers r and r + 8 of the same lane.
// Epilogue: one v_permlane32_swap per accumulator register hands every lane ALL h_aff channels of ONE pixel (lane (l31, lhi) <- pixel
// (row 2*wave + lhi, column l31)); z and h_ft of the item are loaded while its last chunks are in the matrix pipe (C = 12).
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
constexpr int SUB = NPOSP * 16;                 // bytes of one plane of one octet over the tile positions (padded to 10 x 64)
constexpr int STG = 2 * SUB, NSTG = 4;          // a stage = one octet, both planes: 20 480 B; four of them
constexpr int NCH = 8, NSTEP = 5;               // octet chunks per item (Cin = 64), tap-pair steps per chunk
constexpr int WTAP = 32 * 8 * 2;                // one tap of one (octet, plane): [32 rows][8] fp16 = 512 B
constexpr int WOCT = 2 * 9 * WTAP;              // [plane][tap = dx*3 + dy][32][8]: 9 216 B per octet
constexpr int WRES = NCH * WOCT + WTAP;         // + one block of zeros (the tenth tap): 74 240 B
constexpr int LDS_CONV = NSTG * STG + WRES;     // 156 160 B; behind it the per-channel parameters of the epilogue (PARAM_FLOATS<C>)
constexpr int NPL = NG / 2;                     // DMA pieces per loader wave and chunk
constexpr unsigned OOB = 0x80000000u;
template <int CF> struct TailGeo {
    static constexpr int CO2 = 2 * (CF - CF / 2);
    static constexpr int PW_ = 0, PB_ = CF * CF, PS_ = PB_ + CO2, AB_ = PS_ + CO2, AE_ = AB_ + CF, NPAR = AE_ + CF;   // float offsets: wmat, bias, post_scale, an_bias, an_escale
    static constexpr int LDS = LDS_CONV + NPAR * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

struct Item { int b, x0, y0; };

__device__ __forceinline__ void wait_vmcnt_ring(int chunks)      // everything but the youngest `chunks` chunks (NPL pieces each) has landed
{
    if (chunks >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (chunks == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
static_assert(NPL == 5, "wait_vmcnt_ring's immediates");

// REV (= args.reverse) is a template parameter: as a run-time flag every per-channel `reverse ? a : b` became a branch.
// MODE 0: the coupling tail (CF = flow channels).  MODE 1: the same conv with a PLAIN conv epilogue (bfsr_conv3x3_h2r: 64 -> <= 32 channels over an
// h2 tensor, fp32 NCHW or quad-major output) -- CF = 16 or 32 is the output-channel class, and the argument struct is reused: bias = the
// [Cout][2] float4 epilogue table of bfsr_pack_epilogue (or null), z_out = y, C = Cout, h_ft_fmt = quad-major output, reverse = act, eps = slope.
template <int MODE, int CF, int REV>
__global__ __launch_bounds__((NW + NLW) * 64, 1) void coupling_tail_kernel(BfsrCouplingTailArgs q, int tiles_x, int tiles_y, int nitems)
{
    constexpr int CFN = CF / 2, CO2 = MODE ? CF : 2 * (CF - CFN);
    constexpr bool TWO = CO2 <= 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sW = smem + NSTG * STG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int G = gridDim.x;
    const int slot = (int)bfsr::xcd_order(blockIdx.x, (unsigned)G);
    if (slot >= nitems) return;
    const int H = q.H, W = q.W;
    const long long HW = (long long)H * W;
    const unsigned HW16 = (unsigned)(H * W) * 16u;                       // bytes of one (octet, plane) image

    auto decode = [&](int it) {
        Item r;
        int t = it;
        const int ty = t % tiles_y; t /= tiles_y;
        r.x0 = (t % tiles_x) * 32; r.y0 = ty * TH; r.b = t / tiles_x;
        return r;
    };
    typedef TailGeo<CF> Geo;
    float* sPar = reinterpret_cast<float*>(smem + LDS_CONV);
    {   // all twelve waves: weights -> LDS once (ordinary loads, landed before the barrier) ...
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(q.w);
        uint4* dst = reinterpret_cast<uint4*>(sW);
        for (int i = tid; i < WRES / 16; i += (NW + NLW) * 64) dst[i] = src[i];
        // ... and the per-channel parameters of the epilogue, absent ones as identities (no run-time branches per channel later; read from
        // global memory inside the item loop they become ~100 dependent vector loads per wave and item: hipcc cannot use scalar loads for
        // memory the kernel's own stores might alias, and that was half of the kernel's time -- profiles/r04_c_tail_ablation.txt)
        if constexpr (MODE == 1) {                                       // [5][32]: bias, shift, scale, post-add, post-scale per channel (identity beyond Cout / without a table)
            const float4* __restrict__ epi = reinterpret_cast<const float4*>(q.bias);
            for (int i = tid; i < 160; i += (NW + NLW) * 64) {
                const int k = i >> 5, co = i & 31;
                float v = (k == 2 || k == 4) ? 1.f : 0.f;
                if (epi && co < q.C) { const float4 e0 = epi[co * 2]; v = k == 0 ? e0.x : k == 1 ? e0.y : k == 2 ? e0.z : k == 3 ? e0.w : epi[co * 2 + 1].x; }
                sPar[i] = v;
            }
        } else
        for (int i = tid; i < Geo::NPAR; i += (NW + NLW) * 64) {
            float v;
            if (i < Geo::PB_) v = q.wmat ? q.wmat[i] : ((i / CF) == (i % CF) ? 1.f : 0.f);
            else if (i < Geo::PS_) v = q.bias[i - Geo::PB_];
            else if (i < Geo::AB_) v = q.post_scale[i - Geo::PS_];
            else if (i < Geo::AE_) v = q.an_bias ? q.an_bias[i - Geo::AB_] : 0.f;
            else v = q.an_bias ? q.an_escale[i - Geo::AE_] : 1.f;
            sPar[i] = v;
        }
        __syncthreads();
    }
    const int T = ((nitems - slot + G - 1) / G) * NCH;                   // chunks this workgroup walks through

    if (wave >= NW) {
        // ---- loader waves: LDS-DMA only.  Loader ld stages plane ld&1, position groups g with g % 2 == ld>>1, of every chunk.
        // Barrier n = "chunk n is in LDS, and every compute wave is past its reads of chunk n-1", so the stage of chunk n-1 is refilled
        // with chunk n+3: three chunks in flight.  `s_waitcnt vmcnt(k * NPL)` = "all but the youngest k chunks have landed" because a
        // loader issues NPL pieces per chunk and nothing else.
        const int ld = wave - NW, pl = ld & 1, gpar = ld >> 1;
        __amdgpu_buffer_rsrc_t rs_in;
        unsigned vg[NPL];
        auto lsetup = [&](const Item& it) {
            rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(q.hid + (long long)it.b * q.hid_bs), 0, 16u * HW16, 0x00020000);
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const int pos = (gpar + 2 * j) * 64 + lane;
                const int r = pos / PW, c = pos - r * PW;
                const int gy = it.y0 + r - 1, gx = it.x0 + c - 1;
                const bool ok = pos < NPOS && gy >= 0 && gy < H && gx >= 0 && gx < W;
                vg[j] = ok ? (unsigned)(gy * W + gx) * 16u : OOB;        // out of range -> the DMA writes zeros (= the padding)
            }
        };
        int issued = 0, iss_it = slot, iss_c = 0;
        auto issue = [&]() {
            if (iss_c == 0) lsetup(decode(iss_it));
            unsigned char* base = smem + (issued & (NSTG - 1)) * STG + pl * SUB;
            const unsigned soff = (unsigned)(iss_c * 2 + pl) * HW16;
#pragma unroll
            for (int j = 0; j < NPL; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(base + (gpar + 2 * j) * 1024), 16, vg[j], soff, 0, 0);
            ++issued;
            if (++iss_c == NCH) { iss_c = 0; iss_it += G; }
        };
        for (int i = 0; i < NSTG - 1 && issued < T; ++i) issue();
        for (int n = 0; n < T; ++n) {
            wait_vmcnt_ring(issued - n - 1);
            __builtin_amdgcn_s_barrier();
            if (issued < T) issue();
        }
        return;
    }

    // ---- compute waves: wave w owns rows 2w, 2w+1.  A step = one tap PAIR of one octet: 2 input rows x 2 planes + the two weight planes;
    // the fragments of step s+1 are read while the MFMAs of step s run (register double buffer).
    half8 bq[2][2][2], aq[2][2];                                         // [buffer][plane][row] | [buffer][plane]
    auto load_step = [&](auto buf_, int st, int s, int c) {
        constexpr int BUF = decltype(buf_)::value;
        const int ta = 2 * s, tb = 2 * s + 1 < 9 ? 2 * s + 1 : 8;       // tap = dx*3 + dy; the tenth tap re-reads tap 8 (its weights are zeros)
        const int pa = ((ta % 3) * PW + ta / 3) * 16, pb = ((tb % 3) * PW + tb / 3) * 16;
        const unsigned char* inB = smem + st * STG + ((2 * wave) * PW + l31) * 16 + (lhi ? pb : pa);
        const int wa = c * WOCT + ta * WTAP, wb = 2 * s + 1 < 9 ? c * WOCT + (2 * s + 1) * WTAP : NCH * WOCT;
        const unsigned char* wA = sW + (lhi ? wb : wa) + l31 * 16;
        const int wpl = (2 * s + 1 < 9) ? 9 * WTAP : (lhi ? 0 : 9 * WTAP);                 // the zero block serves both planes
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int r = 0; r < 2; ++r) bq[BUF][pl][r] = *reinterpret_cast<const half8*>(inB + pl * SUB + r * PW * 16);
            aq[BUF][pl] = *reinterpret_cast<const half8*>(wA + pl * wpl);
        }
    };
    f32x16 acc[2];
    auto mfma_step = [&](auto buf_) {
        constexpr int BUF = decltype(buf_)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (TWO) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[BUF][1], bq[BUF][1][j], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[BUF][0], bq[BUF][0][j], acc[j], 0, 0, 0);
            } else {                                                     // smallest terms first: w_lo*x_hi, w_hi*x_lo, w_hi*x_hi
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[BUF][1], bq[BUF][0][j], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[BUF][0], bq[BUF][1][j], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[BUF][0], bq[BUF][0][j], acc[j], 0, 0, 0);
            }
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    int st = 0;                                                          // LDS stage of the current chunk
    unsigned bad = 0u;
    for (int it = slot; it < nitems; it += G) {
        const Item cur = decode(it);
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        // One barrier per chunk, passed EARLY: chunk c+1's barrier sits before the last step of chunk c (whose fragments are already in
        // registers).  Five steps per chunk flip the fragment-buffer parity from chunk to chunk: chunks are processed in pairs.
        auto chunk_body = [&](auto p_, auto q_, bool last, int c) {       // p_: buffer holding step 0's fragments (already loaded)
#pragma unroll
            for (int s = 0; s < NSTEP - 1; s += 2) {
                load_step(q_, st, s + 1, c);
                __builtin_amdgcn_sched_barrier(0);
                mfma_step(p_);
                __builtin_amdgcn_sched_barrier(0);
                load_step(p_, st, s + 2, c);
                __builtin_amdgcn_sched_barrier(0);
                mfma_step(q_);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!last) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the last step's fragments have left stage `st`
                __builtin_amdgcn_s_barrier();                            // chunk c+1 has landed; stage `st` may be refilled
                load_step(q_, (st + 1) & (NSTG - 1), 0, c + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(p_);
            __builtin_amdgcn_sched_barrier(0);
            st = (st + 1) & (NSTG - 1);
        };
        // this lane's pixel and the operands of its pointwise chain
        float cz[MODE ? 1 : CF], cft[MODE ? 1 : 2 * CF];
        const int cgy = cur.y0 + 2 * wave + lhi, cgx = cur.x0 + l31;
        const bool con = cgy < H && cgx < W;
        constexpr bool EARLY = MODE == 0 && CF <= 12;                     // C = 24: 72 more live registers through the K loop would spill
        auto tail_prefetch = [&]() {
            if constexpr (MODE == 0) {
            const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.z_in + (long long)cur.b * q.z_in_bs), 0,
                                                                                (unsigned)(CF * HW * 4), 0x00020000);
            const unsigned vo = con ? (unsigned)(((long long)cgy * W + cgx) * 4) : OOB;
#pragma unroll
            for (int c = 0; c < CF; ++c) cz[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, vo, (unsigned)(c * HW * 4), 0));
            if (q.h_ft) {
                const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.h_ft + (long long)cur.b * q.h_ft_bs), 0,
                                                                                    (unsigned)(2 * CF * HW * 4), 0x00020000);
                if (q.h_ft_fmt == 1) {                                   // quad-major [2*CF/4][H][W][4]
                    const unsigned vq = con ? (unsigned)(((long long)cgy * W + cgx) * 16) : OOB;
#pragma unroll
                    for (int c4 = 0; c4 < 2 * CF / 4; ++c4) {
                        const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rf, vq, (unsigned)(c4 * HW * 16), 0));
                        cft[4 * c4] = v.x; cft[4 * c4 + 1] = v.y; cft[4 * c4 + 2] = v.z; cft[4 * c4 + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 2 * CF; ++c) cft[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, vo, (unsigned)(c * HW * 4), 0));
                }
            }
            }
        };
        const float eps = q.eps;
        // IEEE division and expf are ~10 instructions each and all eight compute waves are in this epilogue at once: quotients are formed as
        // v_rcp_f32 + one Newton step on the quotient (<= 1 ulp), exp as v_exp_f32 of x * log2(e) (<= 2 ulp on arguments of a few units)
        auto fdiv = [](float a, float b) {
            const float r = __builtin_amdgcn_rcpf(b);
            const float qt = a * r;
            return fmaf(fmaf(-b, qt, a), r, qt);
        };
        auto sscale = [&](float raw) { return fdiv(1.f, 1.f + __expf(-(raw + 2.f))) + eps; };
        const bool hf = q.h_ft != nullptr;
        // the feature-conditional scales depend on h_ft only: formed under the item's last two chunks (the transcendental pipe is free while the
        // matrix pipe works), in place of the raw values -- reverse: 1 / scale (the division becomes a multiplication, <= 1.5 ulp), forward: scale
        auto ft_scales = [&]() {
            if constexpr (MODE == 0) if (hf) {
#pragma unroll
                for (int c = 0; c < CF; ++c) {
                    const float t = 1.f + __expf(-(cft[2 * c + 1] + 2.f));
                    cft[2 * c + 1] = REV ? fdiv(t, fmaf(eps, t, 1.f)) : fdiv(1.f, t) + eps;   // 1/(1/t + eps) = t / (1 + eps t)
                }
            }
        };
        __builtin_amdgcn_s_barrier();                                    // the item's first chunk has landed in stage `st`
        load_step(I0(), st, 0, 0);
#pragma unroll
        for (int c = 0; c < NCH; c += 2) {                               // pairs of chunks: the parity is static inside a pair
            if (EARLY && c == NCH - 4) tail_prefetch();                  // under the item's last four chunks (~2 us)
            if (EARLY && c == NCH - 2) ft_scales();
#if defined(BFSR_TAIL_ABL) && (BFSR_TAIL_ABL & 2)
            __builtin_amdgcn_s_barrier();                                // ablation: the barrier sequence without LDS reads / MFMAs
            if (c + 2 < NCH) __builtin_amdgcn_s_barrier();
            st = (st + 2) & (NSTG - 1);
#else
            chunk_body(I0(), I1(), false, c);
            chunk_body(I1(), I0(), c + 2 == NCH, c + 1);
#endif
        }
        if (MODE == 0 && !EARLY) { tail_prefetch(); ft_scales(); }

#if defined(BFSR_TAIL_ABL) && (BFSR_TAIL_ABL & 1)
        {                                                                // ablation: no epilogue (accumulators and operands kept alive)
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" :: "v"(acc[0]), "v"(acc[1]));
#pragma unroll
            for (int c = 0; c < CF; ++c) asm volatile("" :: "v"(cz[c]));
#endif
            continue;
        }
#endif
        if constexpr (MODE == 1) {
            // ---- plain conv epilogue: lane (l31, lhi) holds channels 8g + 4*lhi + 0..3 (registers 4g..4g+3) of pixels (row 2*wave + j, column l31)
            int po = 0;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(po));                                 // per-item opaque offset: the parameter reads must not be hoisted into SGPRs
#endif
            const float* sp = sPar + po;
            const float slope = q.reverse == 0 ? 1.f : (q.reverse == 1 ? 0.f : q.eps);
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(q.z_out + (long long)cur.b * q.z_out_bs, 0, (unsigned)(q.C * HW * 4), 0x00020000);
#pragma unroll
            for (int g = 0; g < CO2 / 8; ++g) {
                const int c0 = 8 * g + 4 * lhi;
                const float4 pb = *reinterpret_cast<const float4*>(&sp[c0]), psh = *reinterpret_cast<const float4*>(&sp[32 + c0]),
                             psc = *reinterpret_cast<const float4*>(&sp[64 + c0]), ppa = *reinterpret_cast<const float4*>(&sp[96 + c0]),
                             pps = *reinterpret_cast<const float4*>(&sp[128 + c0]);
                const float b4[4] = {pb.x, pb.y, pb.z, pb.w}, sh4[4] = {psh.x, psh.y, psh.z, psh.w}, sc4[4] = {psc.x, psc.y, psc.z, psc.w},
                            pa4[4] = {ppa.x, ppa.y, ppa.z, ppa.w}, ps4[4] = {pps.x, pps.y, pps.z, pps.w};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int gy = cur.y0 + 2 * wave + j, gx = cur.x0 + l31;
                    const bool ok = gy < H && gx < W;
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = TWO ? (acc[j][4 * g + e] + acc[j][4 * g + e + 8]) * q.acc_scale : acc[j][4 * g + e] * q.acc_scale;
                        v += b4[e];
                        v = (v + sh4[e]) * sc4[e] + pa4[e];
                        v = v > 0.f ? v : v * slope;
                        o[e] = v * ps4[e];
                    }
                    if (q.h_ft_fmt == 1) {
                        const unsigned vo = ok ? (unsigned)(((long long)lhi * HW + (long long)gy * W + gx) * 16) : OOB;
                        bfsr::store_b128(ry, __builtin_bit_cast(bfsr::store_u32x4, make_float4(o[0], o[1], o[2], o[3])), vo, (unsigned)(2 * g * HW * 16));
                    } else {
                        const unsigned vo = ok ? (unsigned)(((long long)gy * W + gx + 4LL * lhi * HW) * 4) : OOB;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[e]), ry, vo, (unsigned)((8 * g + e) * HW * 4), 0);
                    }
                }
            }
            continue;
        }
        // ---- epilogue.  The accumulators are read by compiler-visible VALU code first (hipcc inserts the MFMA -> VALU wait states), the
        // swap statement only sees VALU results (2 wait states, inside the string).
        constexpr int NR = (CO2 + 7) / 8 * 4;                            // accumulator registers that hold channels < CO2 (rows (r&3) + 8(r>>2) + 4*lhi)
        float ha[32];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float x, y;
            if constexpr (TWO) { x = (acc[0][r] + acc[0][r + 8]) * q.acc_scale; y = (acc[1][r] + acc[1][r + 8]) * q.acc_scale; }
            else { x = acc[0][r] * q.acc_scale; y = acc[1][r] * q.acc_scale; }
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
            ha[(r & 3) + 8 * (r >> 2)] = x;                              // this lane's pixel: channel (r&3) + 8(r>>2) ...
            ha[(r & 3) + 8 * (r >> 2) + 4] = y;                          // ... and + 4
        }
        // parameter reads go through a per-item opaque VGPR offset: as loop invariants hipcc would hoist all of them into SGPRs (and spill)
        int po = 0;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(po));
#endif
        const float* sp = sPar + po;
#pragma unroll
        for (int c = CFN; c < CF; ++c) {                                 // this step's self-conditional affine on z2
            const int co = 2 * (c - CFN);
            const float2 bb = *reinterpret_cast<const float2*>(&sp[Geo::PB_ + co]), pp = *reinterpret_cast<const float2*>(&sp[Geo::PS_ + co]);
            const float sh = (ha[co] + bb.x) * pp.x;
            const float sc = sscale((ha[co + 1] + bb.y) * pp.y);
            if constexpr (REV) cz[c] = fdiv(cz[c], sc) - sh; else cz[c] = (cz[c] + sh) * sc;
        }
        const __amdgpu_buffer_rsrc_t rzo = __builtin_amdgcn_make_buffer_rsrc(q.z_out + (long long)cur.b * q.z_out_bs, 0, (unsigned)(CF * HW * 4), 0x00020000);
        const unsigned vzo = con ? (unsigned)(((long long)cgy * W + cgx) * 4) : OOB;      // out-of-image lanes: dropped by the range check, no branch
        auto matvec = [&](int ci) {                                      // (W x)[ci] (identity if absent), the reference's accumulation order
            float a = 0.f;
#pragma unroll
            for (int j4 = 0; j4 < CF; j4 += 4) {
                const float4 w4 = *reinterpret_cast<const float4*>(&sp[Geo::PW_ + ci * CF + j4]);
                a = fmaf(w4.x, cz[j4], a); a = fmaf(w4.y, cz[j4 + 1], a); a = fmaf(w4.z, cz[j4 + 2], a); a = fmaf(w4.w, cz[j4 + 3], a);
            }
            return a;
        };
        auto put = [&](int ci, float a) {
            bad |= (unsigned)!(fabsf(a) < 3.0e38f);                        // NaN / inf guard of the flow state
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, a), rzo, vzo, (unsigned)(ci * HW * 4), 0);
        };
        if constexpr (REV) {
            if (hf) {
#pragma unroll
                for (int c = 0; c < CF; ++c) cz[c] = cz[c] * cft[2 * c + 1] - cft[2 * c];
            }
#pragma unroll
            for (int ci = 0; ci < CF; ++ci) put(ci, matvec(ci) * sp[Geo::AE_ + ci] - sp[Geo::AB_ + ci]);  // ActNorm inverse (identity if absent)
        } else {
#pragma unroll
            for (int c = 0; c < CF; ++c) cz[c] = (cz[c] + sp[Geo::AB_ + c]) * sp[Geo::AE_ + c];           // the next step's ActNorm (identity if absent)
            if (hf) {
#pragma unroll
                for (int ci = 0; ci < CF; ++ci) put(ci, (matvec(ci) + cft[2 * ci]) * cft[2 * ci + 1]);   // ... and its feature-conditional affine
            } else {
#pragma unroll
                for (int ci = 0; ci < CF; ++ci) put(ci, matvec(ci));
            }
        }
    }
    if (q.flag && __any((int)bad)) { if (lane == 0) atomicOr(q.flag, 2u); }
}

template <int MODE, int CF, int REV>
int launch_tail(const BfsrCouplingTailArgs& a, hipStream_t st)
{
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + TH - 1) / TH;
    const long long nitems = (long long)tiles_x * tiles_y * a.B;
    if (nitems <= 0 || nitems > 0x7fffffffLL / NCH) return -1;
    int cus = bfsr::cu_count();
    if (cus <= 0) return -1;
    const long long grid = nitems < cus ? nitems : cus;                  // one persistent workgroup per CU
    static std::atomic<unsigned long long> lds_done{0};
    if (bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&coupling_tail_kernel<MODE, CF, REV>), (MODE ? LDS_CONV + 160 * 4 : TailGeo<CF>::LDS), lds_done) != 0) return -1;
    hipLaunchKernelGGL((coupling_tail_kernel<MODE, CF, REV>), dim3((unsigned)grid), dim3((NW + NLW) * 64), (MODE ? LDS_CONV + 160 * 4 : TailGeo<CF>::LDS), st, a, tiles_x, tiles_y, (int)nitems);
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

// fAffine.4 of a coupled FlowStep (Conv2dZeros [Cout][64][3][3]) as the LDS image of coupling_tail_kernel: [octet][plane][tap = dx*3 + dy]
// [32 rows][8] fp16 of w * scale, then one block of zeros.  Cout > 16: plane 0 = hi, plane 1 = lo.  Cout <= 16, the two-instruction
// form: plane 0 = [rows 0-15: hi | rows 16-31: lo], plane 1 = [rows 0-15: hi | rows 16-31: 0].
extern "C" long long bfsr_coupling_tail_packed_size(int Cin, int Cout)
{
    if (Cin != 64 || Cout <= 0 || Cout > 32) return -1;
    return WRES / 2;                                                      // fp16 elements
}

extern "C" int bfsr_pack_coupling_tail(const float* w, int Cin, int Cout, float scale, unsigned short* packed)
{
    if (!w || !packed || Cin != 64 || Cout <= 0 || Cout > 32 || !(scale > 0.f)) return -1;
    for (long long i = 0; i < WRES / 2; ++i) packed[i] = 0;
    const bool two = Cout <= 16;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx) {
                    const float v = w[((long long)co * Cin + ci) * 9 + dy * 3 + dx] * scale;
                    const _Float16 h = (_Float16)v;
                    const float lo = v - (float)h;
                    const unsigned short hb = f16_bits((float)h), lb = f16_bits(lo);
                    auto at = [&](int plane, int row) { return ((((long long)(ci / 8) * 2 + plane) * 9 + (dx * 3 + dy)) * 32 + row) * 8 + ci % 8; };
                    packed[at(0, co)] = hb;
                    if (two) { packed[at(0, 16 + co)] = lb; packed[at(1, co)] = hb; }
                    else packed[at(1, co)] = lb;
                }
    return 0;
}

extern "C" int bfsr_coupling_tail(const BfsrCouplingTailArgs* a, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || !a->hid || !a->w || !a->bias || !a->post_scale || !a->z_in || !a->z_out) return -1;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->Cin != 64) return -1;
    if (a->an_bias && !a->an_escale) return -1;
    if (!(a->acc_scale > 0.f)) return -1;
    if (a->h_ft_fmt != 0 && a->h_ft_fmt != 1) return -1;
    if ((reinterpret_cast<unsigned long long>(a->hid) & 15) || (a->hid_bs & 7)) return -1;
    if (a->h_ft && a->h_ft_fmt == 1 && ((reinterpret_cast<unsigned long long>(a->h_ft) & 15) || (a->h_ft_bs & 3))) return -1;
    if ((long long)(a->Cin / 8) * 2 * a->H * a->W * 16 >= (1LL << 31)) return -1;
    if ((long long)2 * a->C * a->H * a->W * 4 >= (1LL << 31)) return -1;
    switch (a->C) {
        case 12: return a->reverse ? launch_tail<0, 12, 1>(*a, st) : launch_tail<0, 12, 0>(*a, st);
        case 24: return a->reverse ? launch_tail<0, 24, 1>(*a, st) : launch_tail<0, 24, 0>(*a, st);
        default: return -1;
    }
}

// The same kernel as a plain 3x3 'same' conv 64 -> Cout <= 32 over an h2 tensor (the Conv2dZeros of the hoisted fFeatures nets, flow.py:68-83,
// FlowAffineCouplingsAblation.py:127-135): a->x h2 view, a->w from bfsr_pack_coupling_tail(w, 64, Cout, scale), a->acc_scale = 1/scale, a->y fp32
// [B,Cout,H,W] (y_fmt 0) or quad-major [B][Cout/4][H][W][4] (y_fmt 2), a->epi / act / slope as for bfsr_conv2d.  No residuals.
extern "C" int bfsr_conv3x3_h2r(const BfsrConvX3Args* a, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || !a->x || !a->w || !a->y || a->res1 || a->res2) return -1;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->Cin != 64 || a->Cout <= 0 || a->Cout > 32) return -1;
    if (a->y_fmt != 0 && a->y_fmt != 2) return -1;
    if (a->y_fmt == 2 && ((a->Cout & 3) || (reinterpret_cast<unsigned long long>(a->y) & 15) || (a->y_bs & 3))) return -1;
    if (!(a->acc_scale > 0.f)) return -1;
    if ((reinterpret_cast<unsigned long long>(a->x) & 15) || (a->x_bs & 7)) return -1;
    if ((long long)8 * 2 * a->H * a->W * 16 >= (1LL << 31) || (long long)a->Cout * a->H * a->W * 4 >= (1LL << 31)) return -1;
    BfsrCouplingTailArgs q{};
    q.hid = a->x; q.hid_bs = a->x_bs; q.Cin = 64;
    q.w = a->w; q.acc_scale = a->acc_scale;
    q.bias = reinterpret_cast<const float*>(a->epi);
    q.z_out = reinterpret_cast<float*>(a->y); q.z_out_bs = a->y_bs;
    q.B = a->B; q.C = a->Cout; q.H = a->H; q.W = a->W;
    q.h_ft_fmt = a->y_fmt == 2 ? 1 : 0; q.reverse = a->act; q.eps = a->slope;
    return a->Cout <= 16 ? launch_tail<1, 16, 0>(q, st) : launch_tail<1, 32, 0>(q, st);
}
