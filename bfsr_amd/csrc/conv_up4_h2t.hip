// conv_up4_h2t.hip -- 3x3 'same' conv over a NEAREST-x4-UPSAMPLED tensor, evaluated at the source resolution, at fp32-class accuracy on the fp16
// matrix pipe (two-term fp16 split, three products; conv_h2s.hip's arithmetic).  The hot instance: the 256 RRDB tap channels of the 8x model
// (BASELINE config 4) entering the first conv of every level-1 coupling's conditioning network at 4x the LR size through
// F.interpolate(..., mode='nearest') + torch.cat (SRFlow-LP/code/models/modules/SRFlowNet_arch.py:122-137, RRDBNet_arch.py:105-112,
// FlowAffineCouplingsAblation.py:127-135): 256 -> 16 steps x 64 channels, twice per pass (fAffine and fFeatures).
//
// Phase decomposition: output pixel (4y+py, 4x+px) sees, through its 3x3 window on the upsampled image, per axis
//   p = 0: source offsets -1 (window index 0) and 0 (window indices 1, 2);  p = 1, 2: offset 0 (all three indices);  p = 3: offsets 0 (0, 1) and +1 (2)
// -- phases 1 and 2 see the SAME weights, so per axis there are 3 output classes {0, 12, 3} and 5 (class, offset) entries, in two dimensions
// 9 classes and 25 pre-summed weight blocks: one source pixel row of 32 pixels, one 16-channel chunk and 32 output channels are 25 x 3 = 75
// MFMAs for 16 output pixels per source pixel (a direct 3x3 conv: 16 x 9 x 3).  Same decomposition as the register-staged conv2d_up4 kernel
// (conv_bf16x3.hip) this one replaces on the fast path; what changes is the work decomposition, the one of conv_up2_h2t.hip:
// the taps arrive as an h2 tensor (split once), a workgroup item = source tile 8 x 32 x 32 output channels x ALL nine classes, wave w owns source
// row w = 9 accumulator blocks (144 registers, 2 waves per SIMD) and reads per chunk its 3 rows x 3 column shifts x 2 planes ONCE (18 fragments)
// plus the 50 weight fragments for 75 MFMAs.  No loader waves: every wave issues its share (3 input + up to 7 weight one-KiB pieces) of the NEXT
// chunk by LDS-DMA, one piece per step behind the step's MFMAs; the only vector-memory wait in the K loop is `s_waitcnt vmcnt(0)` in front of the
// chunk barrier.  Two LDS stages of 24 576 (input: [2 planes][2 k halves][384 positions][8]) + 51 200 (weights: [2 planes][25 steps][2 k halves][32][8])
// bytes.  Persistent workgroups; item order = [8 tiles][cout groups] per XCD round.
// Output: fp32 quad-major [B][Cout/4][4h][4w][4], y = acc/scale + pre_add (the channels that live at output resolution enter through pre_add):
// every lane owns whole channel quads of its 16 output pixels in the MFMA result layout -- 64 contiguous bytes per (row, quad), 2 KiB per wave.
#include <hip/hip_runtime.h>
#include <atomic>
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
constexpr int TR = 8, PW = 34, NPOS = (TR + 2) * PW, NG = 6, NPOSP = NG * 64;
constexpr int SUB = NPOSP * 16;                 // one (plane, k half) sub-image: 8 channels of every tile position (6 144)
constexpr int X_IN = 4 * SUB;                   // 24 576
constexpr int NSTEP = 25;                       // (column entry, row entry) weight blocks per chunk
constexpr int W_PL = NSTEP * 1024;              // one weight plane of a chunk
constexpr int W_ST = 2 * W_PL;                  // 51 200
constexpr int STAGE = X_IN + W_ST;              // 75 776
constexpr int LDS_BYTES = 2 * STAGE;            // 151 552
constexpr int NXP = X_IN / 1024 / NWV;          // input pieces per wave and chunk: 3
constexpr int NWPC = W_ST / 1024;               // weight pieces per chunk: 50
constexpr int NWP = (NWPC + NWV - 1) / NWV;     // per wave: 7 (the last round has 2)
constexpr unsigned OOB = 0x80000000u;

struct Item { int cg, b, x0, y0; };

// per-axis entries e = 0..4: (class, source offset) = (0,-1) (0,0) (12,0) (3,0) (3,+1); window indices summed at pack time: {0} {1,2} {0,1,2} {0,1} {2}
constexpr int ax_cls(int e) { return e <= 1 ? 0 : (e == 2 ? 1 : 2); }
constexpr int ax_off(int e) { return e == 0 ? -1 : (e == 4 ? 1 : 0); }
constexpr int ax_lo(int e) { return e == 1 ? 1 : (e == 4 ? 2 : 0); }
constexpr int ax_hi(int e) { return e == 0 ? 0 : (e == 3 ? 1 : 2); }
constexpr int cls_of_phase(int p) { return p == 0 ? 0 : (p == 3 ? 2 : 1); }
// step s = ci*5 + ri (column entry major: the steps of one column shift are contiguous -- shift -1: 0..4, 0: 5..19, +1: 20..24)
struct StepInfo { int cls, roff, cs; };
constexpr StepInfo step_info(int s) { return StepInfo{ax_cls(s % 5) * 3 + ax_cls(s / 5), ax_off(s % 5), ax_off(s / 5)}; }
constexpr int group_of(int s) { return s < 5 ? 0 : (s < 20 ? 1 : 2); }
constexpr int group_start(int g) { return g == 0 ? 0 : (g == 1 ? 5 : (g == 2 ? 20 : NSTEP)); }

// COMPACT (y_fmt 3) is a separate instantiation: as a run-time branch its 36 extra stores pushed the default kernel over its register budget (43 spills, 30 -> 40 ms)
template <bool COMPACT>
__global__ __launch_bounds__(NWV * 64, 1) void conv_up4_h2t_kernel(BfsrUp2H2Args p, int tiles_x, int tiles_y, int groups, int nitems)
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
    const int nchunk = p.Cin >> 4;
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
        r.y0 = (t % tiles_y) * TR; r.b = t / tiles_y;
        return r;
    };

    // ---- staging: wave v issues input pieces v, v+8, v+16 (piece i = sub-image i/6 (plane i/12, k half (i/6)&1), position group i%6)
    // and weight pieces v, v+8, .. < 50
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
    // piece i (0..9: three input pieces, seven weight pieces) of chunk k of the item lsetup() described, into stage stg; k < 0: nothing to stage
    auto lpiece = [&](int i, int k, int stg) {
        if (k < 0) return;
        unsigned char* base = smem + stg * STAGE;
        if (i < NXP) {
            const int piece = wave + NWV * i;
            const int si = piece / NG;                                   // sub-image: plane si>>1, k half si&1
            const unsigned soff = (unsigned)((2 * k + (si & 1)) * 2 + (si >> 1)) * HW16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(base + piece * 1024), 16, vg[i], soff, 0, 0);
        } else {
            const int piece = wave + NWV * (i - NXP);
            if (piece >= NWPC) return;
            const unsigned wsoff = (unsigned)(ld_cg * nchunk + k) * (unsigned)W_ST;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(base + X_IN + piece * 1024), 16, (unsigned)lane * 16u + (unsigned)piece * 1024u, wsoff, 0, 0);
        }
    };
    auto lstage = [&](int k, int stg) {
#pragma unroll
        for (int i = 0; i < NXP + NWP; ++i) lpiece(i, k, stg);
    };

    // ---- fragments: xin[buffer][tile row w + r, r = 0..2][plane] for one column shift cs; wq[buffer][plane] for one weight step
    half8 xin[2][3][2], wq[2][2];
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    auto load_x = [&](auto b_, int stg, int cs) {
        constexpr int BUF = decltype(b_)::value;
        const unsigned char* base = smem + stg * STAGE + (lhi * NPOSP + wave * PW + l31 + cs + 1) * 16;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) xin[BUF][r][pl] = *reinterpret_cast<const half8*>(base + pl * 2 * SUB + r * PW * 16);
    };
    auto load_w = [&](auto b_, int stg, int s) {
        constexpr int BUF = decltype(b_)::value;
        const unsigned char* base = smem + stg * STAGE + X_IN + s * 1024 + lane * 16;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wq[BUF][pl] = *reinterpret_cast<const half8*>(base + pl * W_PL);
    };
    f32x16 acc[9];                                                       // [row class * 3 + column class]
    auto mfma_step = [&](auto s_) {
        constexpr int S = decltype(s_)::value;
        constexpr StepInfo si = step_info(S);
        constexpr int XB = group_of(S) & 1, WB = S & 1, R = si.roff + 1;
        // smallest terms first: w_lo*x_hi, w_hi*x_lo, w_hi*x_hi
        acc[si.cls] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[WB][1], xin[XB][R][0], acc[si.cls], 0, 0, 0);
        acc[si.cls] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[WB][0], xin[XB][R][1], acc[si.cls], 0, 0, 0);
        acc[si.cls] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[WB][0], xin[XB][R][0], acc[si.cls], 0, 0, 0);
    };
    // one chunk: the column-shift groups alternate between the two input fragment buffers; the fragments of the next group / step are read
    // while the current MFMAs run; the ten LDS-DMA pieces of the next chunk (lk; into the other stage) go out one per step behind the MFMAs
    auto step = [&](auto s_, int stg, int lk) {
        constexpr int S = decltype(s_)::value;
        constexpr int GR = group_of(S);
        if constexpr (S == group_start(GR) && group_start(GR + 1) < NSTEP)      // first step of a group: prefetch the next group's rows
            load_x(std::integral_constant<int, (GR + 1) & 1>(), stg, step_info(group_start(GR + 1)).cs);
        if constexpr (S + 1 < NSTEP) load_w(std::integral_constant<int, (S + 1) & 1>(), stg, S + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(s_);
        if constexpr (S < NXP + NWP) lpiece(S, lk, stg ^ 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto chunk_body = [&](int stg, int lk) {
        load_x(I0(), stg, step_info(0).cs);
        load_w(I0(), stg, 0);
#define ST_(N_) step(std::integral_constant<int, N_>(), stg, lk);
        ST_(0) ST_(1) ST_(2) ST_(3) ST_(4) ST_(5) ST_(6) ST_(7) ST_(8) ST_(9) ST_(10) ST_(11) ST_(12) ST_(13) ST_(14) ST_(15) ST_(16) ST_(17)
        ST_(18) ST_(19) ST_(20) ST_(21) ST_(22) ST_(23) ST_(24)
#undef ST_
    };

    const int H4 = 4 * h, W4 = 4 * w;
    const unsigned Q16 = (unsigned)(H4 * W4) * 16u;                      // bytes of one output channel quad image
    int it = slot;
    lsetup(decode(it));
    lstage(0, 0);
    int stg = 0;                                                         // LDS stage of the chunk to compute next
    bool drained = false;                                                // this wave's pieces of that chunk are known to have landed
    for (; it < nitems; it += G) {
        const Item cur = decode(it);
        const int nxt = it + G;
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
#pragma unroll 1
        for (int k = 0; k < nchunk; ++k) {
            if (!drained) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            drained = false;
            __builtin_amdgcn_s_barrier();                                // chunk k is in stage stg; every wave is past its reads of stage stg^1
            int lk = k + 1;
            if (lk == nchunk) {
                lk = -1;
                if (nxt < nitems) { lsetup(decode(nxt)); lk = 0; }
            }
            __builtin_amdgcn_sched_barrier(0);
            chunk_body(stg, lk);
            stg ^= 1;
        }
        // ---- epilogue: y = acc * acc_scale + pre_add, fp32 quad-major.  Result layout of the 32x32 MFMA: lane (l31 = pixel, lhi),
        // register r = channel (r&3) + 8*(r>>2) + 4*lhi of the 32 -> registers 4i..4i+3 are channel quad 2i + lhi: one 16-byte access.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the DMA pieces of the next item's first chunk: ordinary loads follow
        drained = true;
        // descriptors of this item's 8 channel quads only (32-bit offsets)
        const long long qoff = (long long)cur.cg * 8 * (Q16 >> 2);
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y + (long long)cur.b * p.y_bs + qoff, 0, 8u * Q16, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pre_add ? p.pre_add + (long long)cur.b * p.pre_add_bs + qoff : p.y), 0,
                                                                              p.pre_add ? 8u * Q16 : 0u, 0x00020000);
        int lh = lhi, lx = l31;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(lh), "+v"(lx));                           // per-lane address arithmetic stays inside the item loop
#endif
        const int sx = cur.x0 + lx, sy = cur.y0 + wave;
        const bool ok = sy < h && sx < w;
        if constexpr (COMPACT) {
            // COMPACT output [Cout/4][h][9 classes][w][4]: the nine class values of this lane's source pixel, one ROW of w float4 per (source row, class) --
            // a wave's store for one class is 32 x 16 contiguous bytes (ABI 8; the [h][w][9][4] form of ABI 5 wrote 16 bytes at a 144-byte stride: WRITE_SIZE
            // 51.5 GB per config-4 launch for 21.7 GB of payload, and the consumer fetched 1.6x what it read, profiles/r06q_pmc_traffic_cfg4.json).
            // bfsr_conv3x3_h2x (`up4`) expands them while it writes the full-resolution tensor.  No pre_add in this form.
            const unsigned cqb = (unsigned)(h * w) * 144u;               // bytes of one channel quad's compact image
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(p.y + (long long)cur.b * p.y_bs + (long long)cur.cg * 8 * (cqb >> 2), 0, 8u * cqb, 0x00020000);
            const unsigned vc = ok ? (unsigned)lh * cqb + (unsigned)(sy * 9 * w + sx) * 16u : OOB;
            const unsigned crow = (unsigned)w * 16u;                     // bytes of one class row
#pragma unroll
            for (int c = 0; c < 9; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float4 o;
                    o.x = acc[c][4 * i + 0] * p.acc_scale; o.y = acc[c][4 * i + 1] * p.acc_scale;
                    o.z = acc[c][4 * i + 2] * p.acc_scale; o.w = acc[c][4 * i + 3] * p.acc_scale;
                    bfsr::store_b128_stream(rc, __builtin_bit_cast(u32x4, o), vc, (unsigned)(2 * i) * cqb + (unsigned)c * crow);
                }
            continue;
        }
        // eight rounds (output row phase py, channel-quad pair) of 8 accesses (4 column phases x 2 quads); the pre_add loads of round n+1 are
        // issued before round n's arithmetic and stores (the fragment registers are free here)
        unsigned vo[4];
#pragma unroll
        for (int py = 0; py < 4; ++py)
            vo[py] = ok ? (unsigned)lh * Q16 + (unsigned)((4 * sy + py) * W4 + 4 * sx) * 16u : OOB;          // pixel (4sy+py, 4sx); px adds 16 bytes
        float4 pre[2][4][2];
        auto load_pre = [&](auto b_, int n) {
            constexpr int BUF = decltype(b_)::value;
#pragma unroll
            for (int px = 0; px < 4; ++px)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    pre[BUF][px][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, vo[n >> 1] + 16u * px, (unsigned)(2 * (2 * (n & 1) + j)) * Q16, 0));
        };
        auto finish = [&](auto b_, auto n_) {
            constexpr int BUF = decltype(b_)::value, N = decltype(n_)::value;
            constexpr int PY = N >> 1;
#pragma unroll
            for (int px = 0; px < 4; ++px)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int i = 2 * (N & 1) + j;
                    const f32x16& a = acc[cls_of_phase(PY) * 3 + cls_of_phase(px)];
                    float4 o;
                    o.x = a[4 * i + 0] * p.acc_scale + pre[BUF][px][j].x;
                    o.y = a[4 * i + 1] * p.acc_scale + pre[BUF][px][j].y;
                    o.z = a[4 * i + 2] * p.acc_scale + pre[BUF][px][j].z;
                    o.w = a[4 * i + 3] * p.acc_scale + pre[BUF][px][j].w;
                    bfsr::store_b128_stream(rs_y, __builtin_bit_cast(u32x4, o), vo[PY] + 16u * px, (unsigned)(2 * i) * Q16);
                }
        };
        load_pre(I0(), 0);
#define FIN_(B_, NB_, N_) load_pre(NB_(), N_ + 1); __builtin_amdgcn_sched_barrier(0); finish(B_(), std::integral_constant<int, N_>()); __builtin_amdgcn_sched_barrier(0);
        FIN_(I0, I1, 0) FIN_(I1, I0, 1) FIN_(I0, I1, 2) FIN_(I1, I0, 3) FIN_(I0, I1, 4) FIN_(I1, I0, 5) FIN_(I0, I1, 6)
#undef FIN_
        finish(I1(), std::integral_constant<int, 7>());
    }
}

std::atomic<unsigned long long> g_lds_done{0}, g_lds_done_c{0};

}  // namespace

// ---- C ABI ------------------------------------------------------------------------------------------------------------------------
extern "C" long long bfsr_conv_up4_h2t_packed_size(int Cout, int Ct)
{
    if (Cout <= 0 || Ct <= 0 || Cout % 32 || Ct % 16) return -1;
    return (long long)(Cout / 32) * (Ct / 16) * (W_ST / 2);              // fp16 elements
}

// w_taps: OIHW 3x3 fp32 over the Ct upsampled channels.
// packed: [cout group of 32][chunk of 16 channels][plane hi, lo][step 25][k half][32 couts][8 channels] fp16 of scale * (pre-summed weight):
// step ci*5 + ri = the sum of the window taps rows [ax_lo(ri), ax_hi(ri)] x columns [ax_lo(ci), ax_hi(ci)]; formed in double, split once.
extern "C" int bfsr_pack_conv_up4_h2t(const float* w_taps, int Cout, int Ct, float scale, unsigned short* packed)
{
    if (!packed || !w_taps || Cout <= 0 || Ct <= 0 || Cout % 32 || Ct % 16 || !(scale > 0.f)) return -1;
    const int nchunk = Ct / 16;
    _Float16* out = reinterpret_cast<_Float16*>(packed);
    for (int cg = 0; cg < Cout / 32; ++cg)
        for (int k = 0; k < nchunk; ++k) {
            _Float16* blk = out + ((long long)cg * nchunk + k) * (W_ST / 2);
            for (int s = 0; s < NSTEP; ++s) {
                const int ci = s / 5, ri = s % 5;
                for (int kh = 0; kh < 2; ++kh)
                    for (int co = 0; co < 32; ++co)
                        for (int c = 0; c < 8; ++c) {
                            const float* wp = w_taps + ((long long)(cg * 32 + co) * Ct + (16 * k + 8 * kh + c)) * 9;
                            double sum = 0.0;
                            for (int dy = ax_lo(ri); dy <= ax_hi(ri); ++dy)
                                for (int dx = ax_lo(ci); dx <= ax_hi(ci); ++dx) sum += (double)wp[dy * 3 + dx];
                            const float v = (float)(sum * (double)scale);
                            const _Float16 hi = (_Float16)v;
                            const long long e = ((long long)(s * 2 + kh) * 32 + co) * 8 + c;
                            blk[e] = hi;
                            blk[W_PL / 2 + e] = (_Float16)(v - (float)hi);
                        }
            }
        }
    return 0;
}

extern "C" int bfsr_conv2d_up4_h2t(const BfsrUp2H2Args* a, void* stream)
{
    if (!a || !a->x || !a->w || !a->y || a->B <= 0 || a->h <= 0 || a->w_ <= 0 || a->Cin <= 0 || a->Cin % 16 || a->Cout <= 0 || a->Cout % 32) return -1;
    if (a->Ckey != 0) return -1;                                                                // channels at output resolution enter through pre_add
    if (a->y_fmt != 1 && a->y_fmt != 3) return -1;                                              // quad-major fp32, or the compact class form
    if (a->y_fmt == 3 && (a->pre_add || 8LL * a->h * a->w_ * 144 >= (1LL << 31))) return -1;   // (8 channel quads of the compact image of one sample)
    // 16-byte accesses on y / pre_add (quad-major) and LDS-DMA on x: misaligned views are refused, not faulted on
    if ((reinterpret_cast<unsigned long long>(a->x) & 15) || (a->x_bs & 7)) return -1;
    if ((reinterpret_cast<unsigned long long>(a->y) & 15) || (a->y_bs & 3)) return -1;
    if (a->pre_add && ((reinterpret_cast<unsigned long long>(a->pre_add) & 15) || (a->pre_add_bs & 3))) return -1;
    if ((long long)(a->Cin / 8) * 2 * a->h * a->w_ * 16 >= (1LL << 31)) return -1;              // 32-bit buffer offsets (source, per sample)
    if (a->y_fmt == 1 && 8LL * 16 * a->h * a->w_ * 16 >= (1LL << 31)) return -1;                 // (8 output channel quads of one sample)
    if ((long long)(a->Cout / 32) * (a->Cin / 16) * W_ST >= (1LL << 32)) return -1;              // (the packed weights)
    const int tiles_x = (a->w_ + 31) / 32, tiles_y = (a->h + TR - 1) / TR, groups = a->Cout / 32;
    const long long nitems = (long long)a->B * tiles_x * tiles_y * groups;
    if (nitems >= (1LL << 31)) return -1;
    const int cus = bfsr::cu_count();
    if (cus <= 0) return -2;
    const int grid = (int)(nitems < cus ? nitems : cus);
    if (a->y_fmt == 3) {
        if (bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_up4_h2t_kernel<true>), LDS_BYTES, g_lds_done_c) != 0) return -2;
        hipLaunchKernelGGL(conv_up4_h2t_kernel<true>, dim3(grid), dim3(NWV * 64), LDS_BYTES, static_cast<hipStream_t>(stream), *a, tiles_x, tiles_y, groups, (int)nitems);
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    if (bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_up4_h2t_kernel<false>), LDS_BYTES, g_lds_done) != 0) return -2;
    hipLaunchKernelGGL(conv_up4_h2t_kernel<false>, dim3(grid), dim3(NWV * 64), LDS_BYTES, static_cast<hipStream_t>(stream), *a, tiles_x, tiles_y, groups, (int)nitems);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
