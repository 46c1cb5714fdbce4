// linf_mlp.hip -- fused per-point conditioning of LINF-LP: local-ensemble Fourier features -> shared MLP -> affine_info
// (LINF-LP/models/linf.py:324-391: the K12 prologue + K13 MLP of SURVEY section 7 item 6), one kernel.
//
//   features[1024] = cat_k ( w_k * coef_k (.) [cos(pi f_k) | sin(pi f_k)] ),  f_k = freq_k . rel_k + phase(rel_cell)   (k = 4 neighbours)
//   affine_info    = W4 relu(W3 relu(W2 relu(W1 features + b1) + b2) + b3) + b4        (1024 -> 256 -> 256 -> 256 -> 2*D*L = 540)
//
// Before this kernel the 1024-channel feature tensor ([B,1024,Q,Q] fp32, 7.7 GB at BASELINE config 3 x4) and the three 256-channel
// hidden tensors went through HBM between five launches; here a workgroup keeps a tile of P = 64 query points on chip from the
// coef|freq gather to the 540 conditioning channels:
//   * 8 waves; wave w owns output rows 32w..32w+31 of every layer (M = 256 = 8 x 32; the 540-row last layer = 17 tiles, waves take
//     tiles w, w+8, w+16), N = 64 points = 2 MFMA column tiles, accumulators stay in registers across the whole K loop;
//   * activations live in LDS as MFMA B fragments ([16-channel chunk][plane][k half][64 points][8]): layer 1 consumes the features
//     in 8 intervals of 128 channels (wave w generates chunk w of an interval with lane = point: 32 gathers from the coef|freq map,
//     8 sincosf, split, 6 ds_write_b128), double-buffered so the VALU work of interval t+1 sits next to the MFMAs of interval t;
//     layers 2-4 read the previous layer's output, which the epilogue wrote back into the same LDS region (98 KB in the 3xBF16
//     arithmetic) after a v_permlane32_swap transposition to channel octets;
//   * weights are NOT staged: every wave needs different rows, so its A fragments (16 B per lane) are loaded straight from
//     global/L2 into registers, three chunks ahead (3.1 MB of packed weights, L2-resident, read once per workgroup);
//   * arithmetic: X3 = fp32-accurate 3xBF16 split (six v_mfma_f32_32x32x16_bf16 per operand pair, the default); otherwise operands
//     rounded to fp16 (precision='fp16', BASELINE config 5).  Features are computed in fp32 with exactly the operation order of
//     linf_features_kernel (linf_ops.hip), accumulation / bias / ReLU in fp32.
// affine_info [B, 540, Q, Q] fp32 is still written to HBM once: it is read twice (query_log_p and query_rgb, linf_flow_kernel).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/bfsr_hip.h"
#include "launch_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifndef BFSR_MLP_NT
#define BFSR_MLP_NT 0                           // measurement builds only: the affine_info rows leave with the nt cache policy
#endif
#ifndef BFSR_MLP_ABL
#define BFSR_MLP_ABL 0                          // ablation builds only (tools/exp/mlp_abl.sh): bit 0 no weight loads, 1 no cf gathers, 2 no output stores, 3 no MFMAs
#endif

namespace {

constexpr int NW = 8, HID = 256, KC = 16;
constexpr int K1 = 4 * HID;                    // 1024 feature channels
constexpr int NCH_HID = HID / KC;              // 16 chunks of 16 channels in a hidden activation

// arithmetic: 0 = operands rounded to fp16 (one product); 1 = exact three-term bf16 split (six products); 2 = two-term fp16 split
// (22 significant bits, three products; weights pre-scaled by a power of two per layer, accumulators multiplied by a.acc_scale[layer])
template <int MDE> struct Mode;
template <> struct Mode<1> {
    static constexpr int PL = 3;
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mode<0> {
    static constexpr int PL = 1;
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mode<2> {
    static constexpr int PL = 2;
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l)
{
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// sin(x), cos(x) for the Fourier features (x = fl(pi * f), the product the reference forms before torch.cos / torch.sin).  The
// libdevice sincosf costs ~250 VALU instructions per call (generic argument reduction, special cases) and made the fused kernel
// VALU-bound (512 calls per query point).  For |x| < 200 (always, unless a checkpoint has enormous frequencies) a three-term
// Cody-Waite reduction by pi/2 with FMAs and the classic single-precision minimax polynomials on [-pi/4, pi/4] give <= 2 ulp
// (checked against fp64 over the argument range in tests/test_linf_gpu.py::test_linf_mlp_fused); larger arguments take the
// library path.
__device__ __forceinline__ void sincos_feat(float x, float& s, float& c)
{
    if (!(fabsf(x) < 200.f)) { sincosf(x, &s, &c); return; }
    const float kf = rintf(x * 0.636619772367581343f);               // nearest multiple of pi/2
    float r = fmaf(-kf, 1.57079637050628662109375f, x);              // pi/2 = C1 + C2 + C3
    r = fmaf(-kf, -4.37113900018624283e-8f, r);
    r = fmaf(-kf, -1.71512451306013e-15f, r);
    const float r2 = r * r;
    float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(r2, ps, -1.6666654611e-1f);
    const float sr = fmaf(r * r2, ps, r);                            // sin(r)
    float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(r2, pc, 4.166664568298827e-2f);
    const float cr = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.f));        // cos(r)
    const int k = (int)kf;
    const float s0 = (k & 1) ? cr : sr, c0 = (k & 1) ? sr : cr;
    s = (k & 2) ? -s0 : s0;
    c = ((k + 1) & 2) ? -c0 : c0;
}

// 8 fp32 values of one point (8 consecutive channels of a chunk half) -> PL fragments of 16 B
template <int X3>
__device__ __forceinline__ void encode8(const float (&v)[8], typename Mode<X3>::frag (&out)[Mode<X3>::PL], float& amax)
{
    if constexpr (X3 == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { __bf16 h, m, l; split3(v[e], h, m, l); out[0][e] = h; out[1][e] = m; out[2][e] = l; }
    } else if constexpr (X3 == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // the rounded value is pinned in a register (launch_util.h, pin_f16): both the stored hi plane and the subtraction must use the SAME fp16 rounding of v
            const float hf = bfsr::pin_f16(v[e]);
            out[0][e] = (_Float16)hf; out[1][e] = (_Float16)(v[e] - hf);
            amax = fmaxf(amax, fabsf(v[e]));                             // range guard of the fp16 split (a.flag)
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) out[0][e] = (_Float16)v[e];
    }
}

// NT = 32-point column tiles per wave: 2 (64-point workgroup tiles) or 4 (round 6: 128-point tiles -- every weight fragment a wave pulls from L2 feeds four
// MFMAs instead of two, half the tiles, half the weight traffic per query point: each tile streams the whole 1.06 MB (fp16) of weights; same summation
// order per point, so the two widths give identical bits.  One workgroup per CU then: 64 accumulators + two gathers in flight per lane).
template <int X3, int NT>
__global__ __launch_bounds__(NW * 64, NT == 4 ? 1 : (X3 == 1 ? 2 : 4)) void linf_mlp_kernel(BfsrLinfMlpArgs a, int tiles_per_image)
{
    typedef Mode<X3> MD;
    typedef typename MD::frag frag;
    constexpr int PL = MD::PL;
    constexpr int P = 32 * NT, PG = NT / 2;            // points per tile; 64-lane point groups of the feature generation
    constexpr int CHUNK = PL * 2 * P * 16;            // bytes of one 16-channel activation chunk: [plane][k half][64 points][8]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // 16 chunks: stage A = chunks 0-7, stage B = 8-15
    float amax = 0.f;                                  // largest |value| this thread hands to the fp16 split (X3 == 2): range guard

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    // XCD-aware tile order: the 4-neighbour gathers of nearby tiles (same and adjacent query rows) hit the same cf lines, so one
    // XCD (= one L2) takes a contiguous range of tiles; with the default round-robin every L2 fetched every line (PMC: 20x the
    // algorithmic bytes of cf at BASELINE config 5)
    const int bid = (int)bfsr::xcd_order(blockIdx.x, gridDim.x);
    const int b = bid / tiles_per_image;
    const long long NQ = (long long)a.qh * a.qw;
    const long long q0 = (long long)(bid - b * tiles_per_image) * P;

    // ---- per-point geometry, identical arithmetic to linf_features_kernel (linf.py:332-383) ------------------------------
    const int h = a.h, w = a.w;
    const float fh = (float)h, fw = (float)w;
    float rel_y[PG][4], rel_x[PG][4], wk[PG][4];       // [point group]: this lane's query point q0 + 64 pg + lane during feature generation
    int off[PG][4];
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
        const long long q = q0 + 64 * pg + lane;
        const long long qq = q < NQ ? q : NQ - 1;
        const float cy = a.coord[((long long)b * NQ + qq) * 2 + 0];
        const float cx = a.coord[((long long)b * NQ + qq) * 2 + 1];
        float area[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float y = cy + ((j & 2) ? a.dy_pos : a.dy_neg);
            float x = cx + ((j & 1) ? a.dx_pos : a.dx_neg);
            y = fminf(fmaxf(y, a.clamp_lo), a.clamp_hi);
            x = fminf(fmaxf(x, a.clamp_lo), a.clamp_hi);
            int jy = (int)nearbyintf(((y + 1.f) * fh - 1.f) / 2.f);
            int jx = (int)nearbyintf(((x + 1.f) * fw - 1.f) / 2.f);
            jy = min(max(jy, 0), h - 1);
            jx = min(max(jx, 0), w - 1);
            const float qy = a.cy0 + a.cy1 * (float)jy;
            const float qx = a.cx0 + a.cx1 * (float)jx;
            rel_y[pg][j] = (cy - qy) * fh; rel_x[pg][j] = (cx - qx) * fw;
            area[j] = fabsf(rel_y[pg][j] * rel_x[pg][j]) + 1e-9f;
            off[pg][j] = jy * w + jx;
        }
        const float tot = ((area[0] + area[1]) + area[2]) + area[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) wk[pg][j] = area[3 - j] / tot;
    }
    const float cell_y = a.cell[b * 2 + 0] * fh, cell_x = a.cell[b * 2 + 1] * fw;
    const long long hw = (long long)h * w;
    const float* __restrict__ cfb = a.cf + (long long)b * a.cf_bs;
    const float PI = 3.14159274101257324f;

    // chunk kc (0..63) of the feature vector: neighbour k = kc / 16, pairs c0 = (kc % 16) * 8 .. +7;
    // k half 0 = cos features (channel k*256 + c), k half 1 = sin features (channel k*256 + 128 + c): W1 is packed in this order
    // k is wave-uniform but not a compile-time constant: the four per-neighbour values are picked with selects (an array
    // indexed by a run-time k would be demoted to scratch memory)
    struct Gather { float co0[8], co1[8], f0[8], f1[8]; };
    auto gen_load = [&](int k, int c0, Gather& g, auto pg_) {
        constexpr int pg = decltype(pg_)::value;
        const int ofs = k == 0 ? off[pg][0] : (k == 1 ? off[pg][1] : (k == 2 ? off[pg][2] : off[pg][3]));
        const float* cfp = cfb + ofs;
        if (a.cf_fmt == 1) {
            // cf as an h2 tensor [512/8][hi, lo][h*w][8] fp16: the 8 channels of a block are ONE 16-byte word per plane -- 8 loads instead of 32
            // (the raw words travel in the Gather's float slots: [0..3] = hi plane, [4..7] = lo plane; decoded in gen_finish)
            const unsigned short* cfh = reinterpret_cast<const unsigned short*>(a.cf) + (long long)b * a.cf_bs;
            auto ld = [&](int oct, float (&dst)[8]) {
                const unsigned short* p0 = cfh + ((long long)(oct * 2) * hw + ofs) * 8;
                const float4 vh = *reinterpret_cast<const float4*>(p0), vl = *reinterpret_cast<const float4*>(p0 + hw * 8);
                dst[0] = vh.x; dst[1] = vh.y; dst[2] = vh.z; dst[3] = vh.w; dst[4] = vl.x; dst[5] = vl.y; dst[6] = vl.z; dst[7] = vl.w;
            };
            const int o = c0 >> 3;
            ld(o, g.co0); ld(HID / 16 + o, g.co1); ld(HID / 8 + o, g.f0); ld(HID / 8 + HID / 16 + o, g.f1);
            return;
        }
        if (BFSR_MLP_ABL & 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { g.co0[e] = 0.5f + e; g.co1[e] = 0.25f * e; g.f0[e] = 0.1f * e + k; g.f1[e] = 0.3f + c0; }
            return;
        }
        // all 32 gathers of a chunk back to back; they are consumed by gen_finish AFTER the MFMAs of the current interval, so the
        // gather latency (PMC: the waves were 65 % parked in s_waitcnt when each pair's loads were waited for separately) is hidden
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            g.co0[e] = cfp[(long long)c * hw]; g.co1[e] = cfp[(long long)(HID / 2 + c) * hw];
            g.f0[e] = cfp[(long long)(HID + c) * hw]; g.f1[e] = cfp[(long long)(HID + HID / 2 + c) * hw];
        }
    };
    auto gen_finish = [&](int k, int c0, const Gather& g_, unsigned char* dst, auto pg_) {
        constexpr int pg = decltype(pg_)::value;
        const float ry = k == 0 ? rel_y[pg][0] : (k == 1 ? rel_y[pg][1] : (k == 2 ? rel_y[pg][2] : rel_y[pg][3]));
        const float rx = k == 0 ? rel_x[pg][0] : (k == 1 ? rel_x[pg][1] : (k == 2 ? rel_x[pg][2] : rel_x[pg][3]));
        const float wgt = k == 0 ? wk[pg][0] : (k == 1 ? wk[pg][1] : (k == 2 ? wk[pg][2] : wk[pg][3]));
        float vc[8], vs[8];
        Gather d;
        if (a.cf_fmt == 1) {
            auto dec = [&](const float (&raw)[8], float (&out)[8]) {
                const f16x8 hv = __builtin_bit_cast(f16x8, make_float4(raw[0], raw[1], raw[2], raw[3]));
                const f16x8 lv = __builtin_bit_cast(f16x8, make_float4(raw[4], raw[5], raw[6], raw[7]));
#pragma unroll
                for (int e = 0; e < 8; ++e) out[e] = (float)hv[e] + (float)lv[e];
            };
            dec(g_.co0, d.co0); dec(g_.co1, d.co1); dec(g_.f0, d.f0); dec(g_.f1, d.f1);
        } else {
            d = g_;
        }
        const Gather& g = d;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            float f = g.f0[e] * ry + g.f1[e] * rx;
            f = f + (cell_y * a.phase[c * 2 + 0] + cell_x * a.phase[c * 2 + 1]);
            float s, cs;
            if constexpr (X3 == 0) {
                // precision 'fp16': the features are rounded to fp16 (2^-11) right below, so the hardware sine / cosine (argument in
                // revolutions, |error| ~1e-6 absolute) are exact enough: sin(pi f) = v_sin(fract(f / 2)); two quarter-rate instructions
                // instead of the ~40 full-rate ones of sincos_feat -- the feature generation was this mode's VALU bound
                const float t = __builtin_amdgcn_fractf(f * 0.5f);
                s = __builtin_amdgcn_sinf(t);
                cs = __builtin_amdgcn_cosf(t);
            } else {
                sincos_feat(PI * f, s, cs);
            }
            vc[e] = (wgt * g.co0[e]) * cs;
            vs[e] = (wgt * g.co1[e]) * s;
        }
        frag fc[PL], fs[PL];
        encode8<X3>(vc, fc, amax);
        encode8<X3>(vs, fs, amax);
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) {
            *reinterpret_cast<frag*>(dst + ((pl * 2 + 0) * P + 64 * pg + lane) * 16) = fc[pl];
            *reinterpret_cast<frag*>(dst + ((pl * 2 + 1) * P + 64 * pg + lane) * 16) = fs[pl];
        }
    };

    // ---- weights: packed [layer][m tile][k chunk][plane][64 lanes][8]; this lane's fragment of (tile, chunk, plane) is 16 B
    const unsigned short* __restrict__ wbase = a.wts;
    auto load_a = [&](const unsigned short* wl, int nchunk, int mt, int kc, frag (&dst)[PL]) {
        const unsigned short* p = wl + (((long long)mt * nchunk + kc) * PL * 64 + lane) * 8;
        if (BFSR_MLP_ABL & 1) p = wl + lane * 8;                       // (ablation: one L1-resident fragment)
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) dst[pl] = *reinterpret_cast<const frag*>(p + pl * 64 * 8);
    };
    auto load_b = [&](const unsigned char* chunk, frag (&dst)[PL][NT]) {
#pragma unroll
        for (int pl = 0; pl < PL; ++pl)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                dst[pl][nt] = *reinterpret_cast<const frag*>(chunk + ((pl * 2 + lhi) * P + nt * 32 + l31) * 16);
    };
    // acc[nt] += A(tile) x B(chunk): X3 = six cross products, small terms first
    auto mma = [&](f32x16 (&acc)[NT], const frag (&af)[PL], const frag (&bf)[PL][NT]) {
        if constexpr (X3 == 1) {
#define BFSR_T(PA_, PB_) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) acc[nt] = MD::mfma(af[PA_], bf[PB_][nt], acc[nt]);
            BFSR_T(2, 0) BFSR_T(0, 2) BFSR_T(1, 1) BFSR_T(1, 0) BFSR_T(0, 1) BFSR_T(0, 0)
#undef BFSR_T
        } else if constexpr (X3 == 2) {
#define BFSR_T(PA_, PB_) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) acc[nt] = MD::mfma(af[PA_], bf[PB_][nt], acc[nt]);
            BFSR_T(1, 0) BFSR_T(0, 1) BFSR_T(0, 0)
#undef BFSR_T
        } else {
            if (BFSR_MLP_ABL & 8) { acc[0][0] += (float)af[0][0] * (float)bf[0][0][0]; return; }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = MD::mfma(af[0], bf[0][nt], acc[nt]);
        }
    };
    // K loop of one output tile over n chunks of 16 channels whose B operands sit at chunk0, chunk0 + CHUNK, ...: the A fragments
    // (weights, straight from global / L2: ~1 us away) run THREE chunks ahead in a ring of four -- with one chunk of lookahead
    // the loop was bound by the weight-load latency (12 MFMAs = 384 cycles per chunk vs > 1000 cycles of L2 latency)
    auto k_loop = [&](f32x16 (&acc_)[NT], const unsigned short* wl, int nchunk_total, int mt, int kc0, int n, const unsigned char* chunk0) {
        frag af[4][PL], bf[PL][NT];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < n) load_a(wl, nchunk_total, mt, kc0 + i, af[i]);
#pragma unroll 4
        for (int c = 0; c < n; ++c) {
            if (c + 3 < n) load_a(wl, nchunk_total, mt, kc0 + c + 3, af[(c + 3) & 3]);
            load_b(chunk0 + c * CHUNK, bf);
            mma(acc_, af[c & 3], bf);
        }
    };
    // one output tile of a hidden layer -> bias, ReLU, channel-octet transposition, re-encode, write as activation chunks
    auto store_hidden = [&](const f32x16 (&acc)[NT], const float* __restrict__ bias, int mt, float asc) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float v[2][8];
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");       // MFMA result -> VALU read inside the asm below: 20 wait states (>= 19 of a 16-pass XDL op), self-sufficient                 // MFMA result -> VALU read inside the asm below
#pragma unroll
            for (int qd = 0; qd < 2; ++qd)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float lo = acc[nt][8 * qd + i], hi = acc[nt][8 * qd + 4 + i];
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
                    v[qd][i] = lo; v[qd][4 + i] = hi;
                }
#pragma unroll
            for (int qd = 0; qd < 2; ++qd) {
                const int oct = qd * 2 + lhi;                        // channel octet inside the 32-row tile
                const int ch0 = mt * 32 + oct * 8;
                float u[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float t = (X3 == 2 ? v[qd][e] * asc : v[qd][e]) + bias[ch0 + e]; u[e] = t > 0.f ? t : 0.f; }
                frag fr[PL];
                encode8<X3>(u, fr, amax);
                unsigned char* dst = smem + (ch0 >> 4) * CHUNK;
#pragma unroll
                for (int pl = 0; pl < PL; ++pl)
                    *reinterpret_cast<frag*>(dst + ((pl * 2 + (oct & 1)) * P + nt * 32 + l31) * 16) = fr[pl];
            }
        }
    };

    // =============================== layer 1: features (K = 1024) -> 256, feature generation interleaved =================
    const unsigned short* w1 = wbase;
    constexpr long long WSZ1 = (long long)(HID / 32) * (K1 / KC) * PL * 64 * 8;
    constexpr long long WSZH = (long long)(HID / 32) * NCH_HID * PL * 64 * 8;
    typedef std::integral_constant<int, 0> G0;
    typedef std::integral_constant<int, 1> G1;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    {
        Gather g0[PG];
        gen_load(0, wave * 8, g0[0], G0());
        if constexpr (PG > 1) gen_load(0, wave * 8, g0[PG - 1], G1());
        gen_finish(0, wave * 8, g0[0], smem + wave * CHUNK, G0());    // interval 0 (chunk kc = wave) -> stage A
        if constexpr (PG > 1) gen_finish(0, wave * 8, g0[PG - 1], smem + wave * CHUNK, G1());
    }
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < 8; ++t) {
        // interval t+1: chunk kc = (t+1)*8 + wave -> neighbour (t+1)/2, pair block kc % 16: gathers issued, then the MFMAs of
        // interval t (which read stage t), then the arithmetic of interval t+1 into the other stage
        const int nk = (t + 1) >> 1, nc0 = ((((t + 1) & 1) * 8) + wave) * 8;
        Gather g[PG];
        if (t + 1 < 8) {
            gen_load(nk, nc0, g[0], G0());
            if constexpr (PG > 1) gen_load(nk, nc0, g[PG - 1], G1());
        }
        k_loop(acc, w1, K1 / KC, wave, t * 8, 8, smem + (t & 1) * 8 * CHUNK);
        if (t + 1 < 8) {
            gen_finish(nk, nc0, g[0], smem + (((t + 1) & 1) * 8 + wave) * CHUNK, G0());
            if constexpr (PG > 1) gen_finish(nk, nc0, g[PG - 1], smem + (((t + 1) & 1) * 8 + wave) * CHUNK, G1());
        }
        __syncthreads();
    }
    store_hidden(acc, a.bias, wave, a.acc_scale[0]);                 // all waves are past the last interval's reads (barrier above)
    __syncthreads();

    // =============================== layers 2, 3: 256 -> 256 ==============================================================
    for (int layer = 1; layer <= 2; ++layer) {
        const unsigned short* wl = wbase + WSZ1 + (layer - 1) * WSZH;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        k_loop(acc, wl, NCH_HID, wave, 0, NCH_HID, smem);
        __syncthreads();                                             // everybody has read the layer input
        store_hidden(acc, a.bias + layer * HID, wave, a.acc_scale[layer]);
        __syncthreads();
    }

    // =============================== layer 4: 256 -> Cout (540), fp32 to HBM ==============================================
    {
        const unsigned short* wl = wbase + WSZ1 + 2 * WSZH;
        const float* __restrict__ b4 = a.bias + 3 * HID;
        const int mtiles = (a.Cout + 31) / 32;
        float* __restrict__ outb = a.out + (long long)b * a.out_bs;
        for (int mt = wave; mt < mtiles; mt += NW) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
            k_loop(acc, wl, NCH_HID, mt, 0, NCH_HID, smem);
            if constexpr (X3 == 2) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[nt][r] *= a.acc_scale[3];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const long long qq = q0 + nt * 32 + l31;
                if (qq >= NQ) continue;
                if ((BFSR_MLP_ABL & 4) && acc[nt][0] != 1234.5f) continue;
                if (a.out_fmt == 1) {
                    // quad-major [Cout/4][NQ][4]: accumulator registers 4g .. 4g+3 are four CONSECUTIVE output rows of this lane's
                    // query point = one 16-byte store (the row-major form issued 16 four-byte stores per tile and half-wave-wide
                    // 128-byte segments: 1.5x write amplification and a store-issue-bound tail, profiles/r02_pmc_traffic.json)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int co = mt * 32 + 8 * gq + 4 * lhi;
                        if (co < a.Cout) {
                            const float4 bq = *reinterpret_cast<const float4*>(b4 + co);
                            const float4 ov = make_float4(acc[nt][4 * gq] + bq.x, acc[nt][4 * gq + 1] + bq.y, acc[nt][4 * gq + 2] + bq.z, acc[nt][4 * gq + 3] + bq.w);
                            float4* op = reinterpret_cast<float4*>(outb + ((long long)(co >> 2) * NQ + qq) * 4);
#if BFSR_MLP_NT
                            __builtin_nontemporal_store(ov.x, &op->x); __builtin_nontemporal_store(ov.y, &op->y);
                            __builtin_nontemporal_store(ov.z, &op->z); __builtin_nontemporal_store(ov.w, &op->w);
#else
                            *op = ov;
#endif
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (co < a.Cout) outb[(long long)co * NQ + qq] = acc[nt][r] + b4[co];
                }
            }
        }
    }
    if (X3 == 2 && a.flag && __any((int)!(amax < 65504.f))) { if (lane == 0) atomicOr(a.flag, 1u); }
}

template <int X3, int NT>
int launch_mlp(const BfsrLinfMlpArgs& a, hipStream_t st)
{
    constexpr int P = 32 * NT;
    constexpr int LDS = 16 * Mode<X3>::PL * 2 * P * 16;              // NT = 2: 98 304 B (x3) / 65 536 (f16x2) / 32 768 B (fp16); NT = 4: 131 072 (f16x2) / 65 536 (fp16)
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static std::atomic<unsigned long long> lds_done{0};
    if (LDS > 65536 && bfsr::ensure_dynamic_lds(reinterpret_cast<const void*>(&linf_mlp_kernel<X3, NT>), LDS, lds_done) != 0) return -1;
    const long long NQ = (long long)a.qh * a.qw;
    const long long tiles = (NQ + P - 1) / P;
    const long long nblk = tiles * a.B;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return -1;
    hipLaunchKernelGGL((linf_mlp_kernel<X3, NT>), dim3((unsigned)nblk), dim3(NW * 64), LDS, st, a, (int)tiles);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" long long bfsr_linf_mlp_packed_size(int hidden, int Cout, int x3)
{
    if (hidden != HID || Cout <= 0) return -1;
    const long long PL = x3 == 1 ? 3 : (x3 == 2 ? 2 : 1);
    const long long mt4 = (Cout + 31) / 32;
    return ((long long)(HID / 32) * (K1 / KC) + 2LL * (HID / 32) * NCH_HID + mt4 * NCH_HID) * PL * 64 * 8;      // 16-bit elements
}

// w1 [256][1024], w2, w3 [256][256], w4 [Cout][256] (row-major fp32, the nn.Conv2d 1x1 weights of linf.py:231-240) -> the
// per-lane fragment order of linf_mlp_kernel: [layer][m tile][k chunk][plane][lane = k half*32 + row][8].  Layer 1's K axis is
// regrouped per neighbour into (cos | sin) halves: chunk kc = k*16 + cb, half 0 -> channel k*256 + cb*8 + e, half 1 -> k*256 + 128 + cb*8 + e.
static int pack_linf_mlp_impl(const float* w1, const float* w2, const float* w3, const float* w4, int hidden, int Cout, int x3,
                              const float* scales, unsigned short* packed)
{
    if (hidden != HID || Cout <= 0 || !w1 || !w2 || !w3 || !w4 || !packed || x3 < 0 || x3 > 2 || (x3 == 2 && !scales)) return -1;
    const int PL = x3 == 1 ? 3 : (x3 == 2 ? 2 : 1);
    long long o = 0;
    float scale = 1.f;
    auto encode = [&](float v, unsigned short (&out)[3]) {
        if (x3 == 1) {
            float r = v;
            for (int i = 0; i < 3; ++i) { const __bf16 hb = (__bf16)r; __builtin_memcpy(&out[i], &hb, 2); r -= (float)hb; }
        } else if (x3 == 2) {
            float r = v * scale;
            for (int i = 0; i < 2; ++i) { const _Float16 hf = (_Float16)r; __builtin_memcpy(&out[i], &hf, 2); r -= (float)hf; }
        } else {
            const _Float16 hf = (_Float16)v;
            __builtin_memcpy(&out[0], &hf, 2);
        }
    };
    auto pack_layer = [&](const float* W, int rows, int K, bool feature_order) {
        const int mtiles = (rows + 31) / 32, nchunk = K / KC;
        for (int mt = 0; mt < mtiles; ++mt)
            for (int kc = 0; kc < nchunk; ++kc) {
                unsigned short* dst = packed + o + ((long long)mt * nchunk + kc) * PL * 64 * 8;
                for (int ln = 0; ln < 64; ++ln) {
                    const int row = mt * 32 + (ln & 31), half = ln >> 5;
                    for (int e = 0; e < 8; ++e) {
                        int kidx;
                        if (feature_order) { const int k = kc >> 4, cb = kc & 15; kidx = k * HID + half * (HID / 2) + cb * 8 + e; }
                        else kidx = kc * KC + half * 8 + e;
                        unsigned short s3[3] = {0, 0, 0};
                        if (row < rows) encode(W[(long long)row * K + kidx], s3);
                        for (int pl = 0; pl < PL; ++pl) dst[(pl * 64 + ln) * 8 + e] = s3[pl];
                    }
                }
            }
        o += (long long)mtiles * nchunk * PL * 64 * 8;
    };
    if (x3 == 2) scale = scales[0];
    pack_layer(w1, HID, K1, true);
    if (x3 == 2) scale = scales[1];
    pack_layer(w2, HID, HID, false);
    if (x3 == 2) scale = scales[2];
    pack_layer(w3, HID, HID, false);
    if (x3 == 2) scale = scales[3];
    pack_layer(w4, Cout, HID, false);
    return 0;
}

extern "C" int bfsr_pack_linf_mlp(const float* w1, const float* w2, const float* w3, const float* w4, int hidden, int Cout, int x3,
                                  unsigned short* packed)
{
    if (x3 == 2) return -1;                       // the two-term fp16 split needs per-layer scales: bfsr_pack_linf_mlp_f16x2
    return pack_linf_mlp_impl(w1, w2, w3, w4, hidden, Cout, x3 ? 1 : 0, nullptr, packed);
}

extern "C" int bfsr_pack_linf_mlp_f16x2(const float* w1, const float* w2, const float* w3, const float* w4, int hidden, int Cout,
                                        const float* scales4, unsigned short* packed)
{
    if (!scales4) return -1;
    for (int i = 0; i < 4; ++i) if (!(scales4[i] > 0.f)) return -1;
    return pack_linf_mlp_impl(w1, w2, w3, w4, hidden, Cout, 2, scales4, packed);
}

extern "C" int bfsr_linf_mlp(const BfsrLinfMlpArgs* a, int x3, void* stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!a || !a->cf || !a->coord || !a->cell || !a->phase || !a->wts || !a->bias || !a->out) return -1;
    if (a->hidden != HID || a->Cout <= 0 || a->Cout > 24 * 32 || a->B <= 0 || a->h <= 0 || a->w <= 0 || a->qh <= 0 || a->qw <= 0) return -1;
    if (a->out_fmt != 0 && a->out_fmt != 1) return -1;
    if (a->out_fmt == 1 && ((a->Cout & 3) || (reinterpret_cast<unsigned long long>(a->out) & 15) || (a->out_bs & 3) ||
                            (reinterpret_cast<unsigned long long>(a->bias) & 15))) return -1;
    if (a->cf_fmt != 0 && a->cf_fmt != 1) return -1;
    if (a->cf_fmt == 1 && ((reinterpret_cast<unsigned long long>(a->cf) & 15) || (a->cf_bs & 7))) return -1;       // 16-byte gathers
    if (a->tile != 0 && a->tile != 64 && a->tile != 128) return -1;
    if (a->tile == 128 && x3 == 1) return -1;                            // the bf16x3 activations of 128 points do not fit LDS (196 KB)
    BfsrLinfMlpArgs c = *a;
    if (x3 == 2) {
        for (int i = 0; i < 4; ++i) if (!(c.acc_scale[i] > 0.f)) return -1;
        return a->tile == 128 ? launch_mlp<2, 4>(c, st) : launch_mlp<2, 2>(c, st);
    }
    if (x3) return launch_mlp<1, 2>(c, st);
    return a->tile == 128 ? launch_mlp<0, 4>(c, st) : launch_mlp<0, 2>(c, st);
}
