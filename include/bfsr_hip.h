/*
 * bfsr_hip.h -- C ABI of libbfsr_hip.so: MI355X (gfx950) kernels for the BFSR hot path
 * (SRFlow-LP / LINF-LP latent-module forward + normalizing-flow inverse).
 *
 * The reference (liyuantsao/BFSR) is pure Python/PyTorch and exports no FFI; its plugin boundary
 * is the `models` registry (LINF-LP/models/models.py:7-23, SRFlow-LP/code/models/models.py:7-23).
 * This ABI sits *underneath* Python modules that keep the registry names; each entry point below
 * names the reference torch call sites it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - all tensors fp32, NCHW; a "view" is (ptr, batch_stride_in_floats, C, H, W) with contiguous
 *     H*W planes and channel stride H*W, so a channel slice of a larger buffer is a valid view;
 *   - the caller owns every buffer (inputs, outputs, scratch); nothing is retained after a call;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), no internal sync;
 *     the only process-global state are two idempotent per-DEVICE caches in the launchers (launch_util.h): the CU count
 *     and one "dynamic-LDS attribute already raised" bit per kernel and device (std::atomic, a racing second caller
 *     repeats the same hipFuncSetAttribute) -- nothing that depends on call order or arguments => re-entrant across
 *     host threads, streams and devices;
 *   - return value: 0 on success, otherwise a hipError_t (or -1 for an unsupported argument);
 *     no exceptions cross the ABI.
 */
#ifndef BFSR_HIP_H
#define BFSR_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define BFSR_ABI_VERSION 8      /* 8 (round 6, late): bfsr_linf_fold_skip / bfsr_linf_prep_down / bfsr_linf_prep_residual, bfsr_resize_h2 / bfsr_maxpool2_h2 / bfsr_h2_pack_pad added; the COMPACT output of bfsr_conv2d_up4_h2t (y_fmt 3) and BfsrConvX3Args.up4 are [Cout/4][h][9][w][4] (class rows: whole cache lines on both sides) instead of [Cout/4][h][w][9][4].  7 (round 6): bfsr_coupling_wide_head / _tail (the coupled FlowStep of the C = 96 level as two streaming kernels) and their pack functions added.  6 (round 6): bfsr_channel_range_check per sample + gain / ratio arguments, bfsr_channel_range_scratch(B, C); bfsr_conv_chain_progress_words counts the give-up word.  5 (round 5, last): BfsrConvX3Args.up4 / up4_bs appended, bfsr_conv2d_up4_h2t y_fmt 3 (compact output).  4 (round 5, late): BfsrLinfMlpArgs.cf_fmt appended; bfsr_conv2d_up4_h2t and its pack functions added.  3 (round 5): BfsrChainConv + the chain entry points, bfsr_channel_range_*, BfsrLinfMlpArgs.flag.  2 (round 4): BfsrConvArgs / BfsrConvX3Args grew y_fmt + flag, coupling head / tail structs redefined, bfsr_conv2d_up2_h2t, bfsr_h2_pack_s2d, bfsr_ssim_sum_w added, the step / up2_h2x entry points removed */

enum { BFSR_ACT_NONE = 0, BFSR_ACT_RELU = 1, BFSR_ACT_LRELU = 2 };

/* ---- dense conv (3x3 'same' or 1x1, stride 1) on fp32 MFMA, fused epilogue ----------------
 * replaces F.conv2d / nn.Conv2d call sites on the path:
 *   SRFlow-LP/code/models/modules/RRDBNet_arch.py:39-45,89-117 (RDB, trunk, upconv)
 *   SRFlow-LP/code/models/modules/flow.py:59-83 (flow.Conv2d + ActNorm, Conv2dZeros)
 *   SRFlow-LP/code/models/unet.py:23-52,101-107 and LINF-LP/models/unet.py (prior UNets)
 *   LINF-LP/models/rrdb.py:52-58, edsr.py:30-51, linf.py:231-240,250-251 (encoders, coef/freq, MLP)
 *
 * epilogue order per output element v (channel c):
 *   v = acc; v += bias[c]; v += pre_add[b,c,y,x]; v = (v + aff_shift[c]) * aff_scale[c] + aff_post[c];
 *   v = act(v); v *= post_scale[c]; v = alpha1*v + res1[b,c,y,x]; v = alpha2*v + res2[b,c,y,x]
 * The five per-channel vectors are passed packed: epi[c] = {bias, aff_shift, aff_scale, aff_post, post_scale, 0, 0, 0}
 * (8 floats per channel; neutral values 0,0,1,0,1 where a stage is unused; epi == NULL = all neutral), which keeps
 * the element math branch-free.  Tensor stages are skipped when their pointer is NULL.
 *
 * `w` is the packed weight produced by bfsr_pack_conv_weight (layout private to the library:
 * [cout_group][cin_pad][tap][mtile*32], zero padded).
 * `in_shift` s>0: the input view is (H>>s, W>>s) and is read through nearest-neighbour x2^s
 * upsampling (RRDBNet_arch.py:105-117 `F.interpolate(..., scale_factor=2, mode='nearest')`).
 */
typedef struct BfsrConvArgs {
    const float* x; long long x_bs; int Cin;
    const float* w;
    float* y; long long y_bs; int Cout;
    int B, H, W, KS, in_shift;
    int mtile;                       /* 32-wide cout tiles per workgroup the weight was packed for */
    const float* epi;                /* [Cout][8] packed per-channel epilogue parameters, or NULL */
    const float* pre_add; long long pre_add_bs;
    int act; float slope;
    const float* res1; long long res1_bs; float alpha1;
    const float* res2; long long res2_bs; float alpha2;
    int tune;                        /* 0 = auto; NR*100+CK forces a kernel variant (benchmarking) */
    /* optional fused second stage (w2 != NULL): a 1x1 conv over the <= 64 stage-1 channels applied in the same
     * kernel; y then has C2 <= 64 channels: y = epilogue2(W2 . stage1) with epi2/act2.  w2 is packed with
     * bfsr_pack_conv_weight(KS=1, mtile=2).  Stage 1 keeps bias / pre_add / aff_* / act. */
    const float* w2; int C2;
    const float* epi2; int act2;      /* stage-2 per-channel parameters, same packing ([C2][8]) */
    /* bfsr_conv2d_up2 only: optional extra input channels that live at OUTPUT resolution ([B,Cin2,H,W] view) with their
     * own ordinary 3x3 weights (bfsr_pack_conv_weight, same mtile): y = epilogue(conv3x3(cat[x2, nearest_up2(x)])) */
    const float* x2; long long x2_bs; int Cin2; const float* w_x2;
    /* bfsr_conv2d_bf16x3 / _up2_bf16x3 / _up4_bf16x3 only -- which split the kernel contracts in:
     * arith 0: exact three-term bf16 split, six products (weights from bfsr_pack_conv_weight[_taps]_bf16x3);
     * arith 1: two-term fp16 split of both operands (22 significant bits), three products, half the matrix instructions; weights
     *          from bfsr_pack_conv_weight_taps_f16x2(..., scale, ...) with scale = a power of two that puts the largest |w| into
     *          [2^9, 2^10) (their lo terms stay normal fp16 numbers); the accumulators are multiplied by acc_scale = 1/scale before
     *          the epilogue; activations must stay below 65504 in magnitude. */
    int arith; float acc_scale;
    /* y_fmt (bfsr_conv2d_bf16x3, bfsr_conv2d_up2_bf16x3 only) 0: y / pre_add are [B,Cout,H,W]; 1: both are QUAD-MAJOR [B][Cout/4][H][W][4] (16-byte
     * accesses per accumulator group; the private layout bfsr_coupling_head reads pre_aff in); needs Cout % 4 == 0, no residuals */
    int y_fmt;
    /* optional device word (arith 1 only): bit 0 is set when an activation handed to the fp16 split is >= 65504 in magnitude */
    unsigned* flag;
} BfsrConvArgs;

int bfsr_abi_version(void);
/* number of floats of the packed weight for (Cout, Cin, KS, mtile) */
long long bfsr_conv_packed_size(int Cout, int Cin, int KS, int mtile);
/* host-side packing: w_oihw [Cout][Cin][KS][KS] -> packed (both host pointers) */
int bfsr_pack_conv_weight(const float* w_oihw, int Cout, int Cin, int KS, int mtile, float* packed);
int bfsr_conv2d(const BfsrConvArgs* a, void* stream);

/* 3x3 'same' conv of nearest_up2(x) computed on the SOURCE resolution at 4/9 of the MACs: x is [B,Cin,H/2,W/2], y is
 * [B,Cout,H,W]; `w` holds the 16 pre-summed [Cout x Cin] matrices, one per (output parity a,b; source offset i,j),
 * tap index t = (a*2+b)*4 + i*2+j, packed with bfsr_pack_conv_weight_taps(T=16).  Row rule (same for columns):
 * a=0: i=0 <- w[dy=-1], i=1 <- w[0]+w[+1];  a=1: i=0 <- w[-1]+w[0], i=1 <- w[+1].  Same epilogue as bfsr_conv2d.
 * With x2/w_x2 set, channels that already live at the output resolution are convolved (plain 3x3) into the same
 * accumulators, i.e. the whole conv over cat[x2, nearest_up2(x)] in one kernel.
 * Replaces conv(F.interpolate(x, scale_factor=2, mode='nearest')) patterns: the x2-upsampled stacked RRDB taps in the
 * level conditionals (SRFlowNet_arch.py:137 + FlowAffineCouplingsAblation.py:108-119), RRDBNet_arch.py:105-117. */
int bfsr_conv2d_up2(const BfsrConvArgs* a, void* stream);
long long bfsr_conv_packed_size_taps(int Cout, int Cin, int T, int mtile);
int bfsr_pack_conv_weight_taps(const float* w_oit, int Cout, int Cin, int T, int mtile, float* packed);

/* Reduced-precision variant of bfsr_conv2d (BASELINE config 5, "fp16 MFMA path"): identical arguments and epilogue, but the
 * contraction runs on v_mfma_f32_32x32x16_f16 -- inputs and weights are rounded to fp16 (RNE) as they are staged,
 * accumulation / epilogue / HBM tensors stay fp32.  `w` = fp16 weights from bfsr_pack_conv_weight_f16; no fused second
 * stage.  Outside the 1e-4 fp32 tolerance by construction (tests report the deviation). */
int bfsr_conv2d_f16(const BfsrConvArgs* a, void* stream);
long long bfsr_conv_packed_size_f16(int Cout, int Cin, int KS, int mtile);      /* in fp16 elements */
int bfsr_pack_conv_weight_f16(const float* w_oihw, int Cout, int Cin, int KS, int mtile, unsigned short* packed);

/* fp32-accurate variant of bfsr_conv2d on the bf16 matrix pipe ("3xBF16" split): every operand is split exactly into three
 * bf16 terms and six of the nine cross products are accumulated in fp32 on v_mfma_f32_32x32x16_bf16; the dropped terms are
 * below one fp32 rounding of the product (error vs an fp64 conv = that of the native fp32 kernel, see tests).  Identical
 * arguments / epilogue; `w` from bfsr_pack_conv_weight_bf16x3; tune = NR*100 selects the tile height (0 = default). */
int bfsr_conv2d_bf16x3(const BfsrConvArgs* a, void* stream);
long long bfsr_conv_packed_size_bf16x3(int Cout, int Cin, int KS, int mtile);   /* in bf16 elements */
int bfsr_pack_conv_weight_bf16x3(const float* w_oihw, int Cout, int Cin, int KS, int mtile, unsigned short* packed);
/* bfsr_conv2d_up2 on the same 3xBF16 scheme: x [B,Cin,H/2,W/2] -> y [B,Cout,H,W] with the 16 parity-pre-summed matrices
 * (bfsr_pack_conv_weight_taps_bf16x3, T=16, mtile=1).  No x2/w_x2: channels at output resolution are convolved by
 * bfsr_conv2d_bf16x3 first and enter through `pre_add` (which may alias y).  tune = NW*100+NR (0 = default). */
int bfsr_conv2d_up2_bf16x3(const BfsrConvArgs* a, void* stream);
/* the same for a nearest-x4-upsampled input (level-1 conditional of the 8x model, SRFlowNet_arch.py:137 with scale 8):
 * x [B,Cin,H/4,W/4] -> y [B,Cout,H,W]; w = 25 pre-summed matrices, index = row_entry*5 + col_entry, entry e = 0..4 per axis:
 * (phase 0, offset -1) <- w[-1]; (phase 0, offset 0) <- w[0]+w[+1]; (phases 1,2, offset 0) <- w[-1]+w[0]+w[+1];
 * (phase 3, offset 0) <- w[-1]+w[0]; (phase 3, offset +1) <- w[+1]   (bfsr_pack_conv_weight_taps_bf16x3, T=25, mtile=1).
 * Epilogue: bias/affine/act + pre_add only (no residuals). */
int bfsr_conv2d_up4_bf16x3(const BfsrConvArgs* a, void* stream);
long long bfsr_conv_packed_size_taps_bf16x3(int Cout, int Cin, int T, int mtile);
int bfsr_pack_conv_weight_taps_bf16x3(const float* w_oit, int Cout, int Cin, int T, int mtile, unsigned short* packed);
long long bfsr_conv_packed_size_taps_f16x2(int Cout, int Cin, int T, int mtile);
int bfsr_pack_conv_weight_taps_f16x2(const float* w_oit, int Cout, int Cin, int T, int mtile, float scale, unsigned short* packed);

/* Wide 1x1 convolutions as a GEMM over the flattened pixel axis (the LINF shared MLP, linf.py:313-314): a workgroup owns
 * 128 pixels x 256 output channels, so activations are read once per 256 couts.  x3 != 0: exact 3-term bf16 split (fp32
 * accurate); x3 == 0: operands rounded to fp16 (LINF precision='fp16').  Same BfsrConvArgs / epilogue as bfsr_conv2d with
 * KS = 1, no in_shift, no fused second stage; `w` from bfsr_pack_conv1x1_weight (16-bit elements). */
int bfsr_conv1x1(const BfsrConvArgs* a, int x3, void* stream);
long long bfsr_conv1x1_packed_size(int Cout, int Cin, int x3);
int bfsr_pack_conv1x1_weight(const float* w_oi, int Cout, int Cin, int x3, unsigned short* packed);

/* ---- x3 tensors: activations stored as the exact 3-term bf16 split -----------------------------------------------------
 * Layout [B][C/8][3 planes h,m,l][H][W][8] bf16 with x = h + m + l exactly (round-to-nearest at each level, 8+8+8 significant
 * bits: a lossless 48-bit encoding of an fp32 value).  A view = (pointer, batch stride in bf16 elements, C); channel slices at
 * multiples of 8 are views.  bfsr_x3_pack / bfsr_x3_unpack convert from / to an fp32 NCHW view.
 *
 * bfsr_conv3x3_x3s: the 3x3 'same' conv of bfsr_conv2d_bf16x3 (same six-product arithmetic, same packed weights with mtile=1)
 * for inputs that are x3 tensors: tiles are staged by LDS-DMA straight from HBM (no split at staging time), the LDS stage is
 * double-buffered and the workgroups are persistent (one per CU).  Used for the dense blocks of the RRDB encoder
 * (SRFlow-LP/code/models/modules/RRDBNet_arch.py:25-65, LINF-LP/models/rrdb.py:38-76): conv1..4 write their 32-channel slice of
 * the block buffer as x3, conv5 applies `x5*0.2 + x` (and `*0.2 + x_rrdb`) with x3 residuals.
 * epilogue: v = acc + bias; v = (v + aff_shift)*aff_scale + aff_post; v = act(v); v *= post_scale;
 *           v = alpha1*v + res1; v = alpha2*v + res2;   y_fmt 0: fp32 NCHW view (y_bs in floats), 1: x3 view (y_bs in bf16 elements).
 * Cin must be a multiple of 16; Cout a multiple of 8 when the output or a residual is x3.  tune > 0 overrides the number of
 * persistent workgroups (default: one per CU). */
typedef struct BfsrConvX3Args {
    const unsigned short* x; long long x_bs; int Cin;
    const unsigned short* w;                       /* bfsr_pack_conv_weight_bf16x3(KS=3, mtile=1) */
    void* y; long long y_bs; int Cout; int y_fmt;
    int B, H, W;
    const float* epi; int act; float slope;        /* [Cout][8] packed per-channel parameters (see bfsr_conv2d), or NULL */
    const unsigned short* res1; long long res1_bs; float alpha1;
    const unsigned short* res2; long long res2_bs; float alpha2;
    int tune;
    float acc_scale;                               /* bfsr_conv3x3_h2x only: 1 / (the power of two the weights were packed with) */
    int mtile;                                     /* bfsr_conv3x3_h2x: 32-cout M tiles per workgroup the weights were packed for (0 or 1); bfsr_conv3x3_h2s: 0, 1 or 2 (| 0x100: keep the weights streamed instead of LDS-resident where they would fit) */
    unsigned* flag;                                /* bfsr_conv3x3_h2x only, optional device word: bit 0 is set when a value written to an h2 output is >= 65504 */
    const float* up4; long long up4_bs;            /* (ABI 5) bfsr_conv3x3_h2x with y_fmt 2 and no residuals only, optional: the COMPACT output of bfsr_conv2d_up4_h2t (its y_fmt 3:
                                                    * [B][Cout/4][H/4][9 phase classes][W/4][4] fp32 (ABI 8), batch stride in floats), added to the result after the
                                                    * epilogue -- the two convs of the x4 level (key channels at output resolution + nearest-x4 taps) meet here
                                                    * instead of through a full-resolution pre_add round trip.  H and W must be multiples of 4. */
} BfsrConvX3Args;
int bfsr_conv3x3_x3s(const BfsrConvX3Args* a, void* stream);
int bfsr_x3_pack(const float* x, long long x_bs, unsigned short* y, long long y_bs, int B, int C, int H, int W, void* stream);
int bfsr_x3_unpack(const unsigned short* x, long long x_bs, float* y, long long y_bs, int B, int C, int H, int W, void* stream);

/* ---- h2 tensors: activations stored in fp16 for the reduced-precision path (LINF precision='fp16', BASELINE config 5) ----------
 * Layout [B][C/8][2 planes hi,lo][H][W][8] fp16 with hi = fp16(x), lo = fp16(x - hi) (x ~ hi + lo to 22 significant bits).  A
 * view = (pointer, batch stride in fp16 elements, C); channel slices at multiples of 8 are views.  bfsr_h2_pack / bfsr_h2_unpack
 * convert from / to an fp32 NCHW view.
 *
 * bfsr_conv3x3_h2s: the 3x3 'same' conv of bfsr_conv2d_f16 (operands rounded to fp16, fp32 accumulation -- the hi plane IS that
 * rounding, so both kernels contract the same numbers) for inputs that are h2 tensors: LDS-DMA staging by dedicated loader waves
 * through a 4-stage LDS ring, persistent workgroups.  Same argument struct and epilogue as bfsr_conv3x3_x3s; residuals are h2
 * views read as hi + lo;  y_fmt 0: fp32 NCHW view, 1: h2 view (both planes), 2: h2 view, hi plane only (outputs that only feed
 * convs).  Weights from bfsr_pack_conv_weight_h2s (OIHW 3x3 fp32 -> fp16 [cout group of 32][16-channel chunk][tap][k half][32][8]).
 * Dense blocks of the RRDB encoder: LINF-LP/models/rrdb.py:38-76. */
int bfsr_conv3x3_h2s(const BfsrConvX3Args* a, void* stream);
long long bfsr_conv_packed_size_h2s(int Cout, int Cin);
int bfsr_pack_conv_weight_h2s(const float* w_oihw, int Cout, int Cin, unsigned short* packed);
/* (round 6) the same with 32 * mtile output channels per workgroup tile, mtile in {1, 2}: mtile 2 stages the input tile once for 64 output channels
 * (conv5 of a dense block, trunk convs); the caller passes the mtile the weights were packed with in BfsrConvX3Args.mtile (0 = 1).  Same summation
 * order per output element: bit-identical results. */
long long bfsr_conv_packed_size_h2s_mt(int Cout, int Cin, int mtile);
int bfsr_pack_conv_weight_h2s_mt(const float* w_oihw, int Cout, int Cin, int mtile, unsigned short* packed);
/* bfsr_conv3x3_h2x: the same conv at fp32-class accuracy on the fp16 matrix pipe -- both planes of the h2 input (22 significant
 * bits) against a two-term fp16 split of the weights, three products lo*hi + hi*lo + hi*hi in the fp32 accumulator (half the
 * matrix instructions and 2/3 of the operand bytes of the 3xBF16 scheme of bfsr_conv3x3_x3s; end to end indistinguishable from
 * fp32 on the SRFlow-LP pipeline, tests/test_srflow_gpu.py).  fp16 has a narrow exponent: bfsr_pack_conv_weight_h2x multiplies
 * the weights by `scale` (a power of two: the caller puts the largest |w| into [2^9, 2^10)) and the kernel multiplies the
 * accumulators by a->acc_scale = 1/scale before the epilogue of bfsr_conv3x3_x3s; activations must stay below 65504 in magnitude.
 * y_fmt 0: fp32 NCHW view, 1: h2 view (both planes).  Cin % 16 == 0.  Replaces the dense-block convs of
 * SRFlow-LP/code/models/modules/RRDBNet_arch.py:25-65 on the default (fp32-accurate) path. */
int bfsr_conv3x3_h2x(const BfsrConvX3Args* a, void* stream);
long long bfsr_conv_packed_size_h2x(int Cout, int Cin, int mtile);
int bfsr_pack_conv_weight_h2x(const float* w_oihw, int Cout, int Cin, int mtile, float scale, unsigned short* packed);
int bfsr_h2_pack(const float* x, long long x_bs, unsigned short* y, long long y_bs, int B, int C, int H, int W, unsigned* flag /* optional range guard */, void* stream);
/* (ABI 8) the same from a tensor with Cs <= C channels: channels >= Cs of the h2 tensor are zero (the K padding of a conv whose input channel count is not a multiple of the
 * kernel's chunk: the 6- and 27-channel latents entering DenseBlock_5C of the learned priors, models/unet.py:10-36) */
int bfsr_h2_pack_pad(const float* x, long long x_bs, unsigned short* y, long long y_bs, int B, int Cs, int C, int H, int W, unsigned* flag, void* stream);
int bfsr_h2_unpack(const unsigned short* x, long long x_bs, float* y, long long y_bs, int B, int C, int H, int W, void* stream);

/* ---- a CHAIN of bfsr_conv3x3_h2x convs in ONE persistent launch (round 5, conv_chain.hip) ------------------------------------------
 * replaces the per-conv launches of the dense blocks: SRFlow-LP/code/models/modules/RRDBNet_arch.py:25-65 (ResidualDenseBlock_5C.forward,
 * RRDB.forward) and LINF-LP/models/rrdb.py:38-74 -- conv i reads the h2 views earlier convs of the chain wrote (channel-slice views of the
 * block buffers), exactly as consecutive bfsr_conv3x3_h2x launches would -- and with exactly their results: same kernel arithmetic, same
 * summation order, bit for bit (tests/test_conv_chain.py); same epilogue.  An item (conv, 16 x 32 tile, 32-cout group) starts as soon as the
 * previous conv has finished the item's 3 x 3 tile neighbourhood (per-tile progress counters,
 * write-through stores + agent-scope atomics); there is no grid barrier, so tile quantisation and launch ramps are paid once per chain.
 * Restrictions: all convs share B, H, W; every conv but the last writes an h2 view (y_fmt 1); buffers may be reused along the chain the
 * way the dense-block ring does (each conv waits for its predecessor, so earlier readers of a region are complete before it is rewritten).
 * y2 (optional): a second, fp32 NCHW copy of the conv's result (tapped RRDB outputs), readable after the launch; for an h2 output it holds
 * hi + lo, i.e. exactly bfsr_h2_unpack(y).
 *   bfsr_conv_chain_prepare validates the descriptors and fills an opaque table (bfsr_conv_chain_table_size bytes, host memory); the
 *   caller keeps a device copy of the same bytes.  bfsr_conv_chain_launch zeroes `progress` (bfsr_conv_chain_progress_words unsigned
 *   words of device memory: one counter per tile + the launch's private give-up word) and launches.  `status` (device word, required):
 *   bit 0 = fp16-split range overflow (as BfsrConvX3Args.flag), bit 2 = a dependency wait of some launch timed out (~2 s; that launch's
 *   results are invalid).  Bit 2 is only ever WRITTEN by the launches: whether waiters give up is decided by the private word, so a sticky
 *   status bit does not make later launches skip their waits (round 6).  tune > 0 shrinks the persistent grid. */
typedef struct BfsrChainConv {
    const unsigned short* x; long long x_bs; int Cin;      /* h2 view */
    const unsigned short* w;                                /* bfsr_pack_conv_weight_h2x(mtile = 1) */
    void* y; long long y_bs; int Cout; int y_fmt;           /* 0 fp32 NCHW | 1 h2 | 2 fp32 quad-major (0 / 2: last conv only) */
    const float* epi; int act; float slope;                 /* as bfsr_conv2d */
    const unsigned short* res1; long long res1_bs; float alpha1;   /* h2 views */
    const unsigned short* res2; long long res2_bs; float alpha2;
    float acc_scale;                                        /* 1 / (the power of two the weights were packed with) */
    float* y2; long long y2_bs;                             /* optional fp32 NCHW copy of the result */
} BfsrChainConv;
long long bfsr_conv_chain_table_size(int nconv);
int bfsr_conv_chain_prepare(const BfsrChainConv* convs, int nconv, int B, int H, int W, void* table_host);
long long bfsr_conv_chain_progress_words(const void* table_host);
int bfsr_conv_chain_launch(const void* table_host, const void* table_dev, unsigned* progress, unsigned* status, int tune, void* stream);

/* ---- per-channel dynamic-range check of a tensor entering an fp16-split region (round 5, range_check.hip; bfsr_amd/guard.py) -----------
 * The two-term fp16 split of the default path (bfsr_conv3x3_h2x and friends) holds 22 significant bits for 2^-3 <= |x| < 65504.  Replaces
 * nothing in the reference: it is what lets the engine keep the reference's fp32 contract (RRDBNet_arch.py:39-45, LINF-LP/models/rrdb.py:52-58
 * contract in true fp32) -- a pass whose tensors leave that range is re-run under the bf16x3 split.  Per SAMPLE b and channel c, m = max |x|
 * over the sample's plane; raises in *flag: bit 0 when some m >= huge or not finite; bit 3 when some channel with 0 < m_c < tiny has
 * max_c'(m_c' * g_c') < g_c * ratio over the same sample -- with ratio = 2^-5 the channel's absolute split error, 2^-25 * g_c, then exceeds
 * 2^-20 of the largest per-channel contribution to the conv that reads the tensor.  gain (optional, device, C floats in [0, 1]): the consumer
 * convs' weight mass per input channel, normalised to its maximum; NULL = 1 for every channel (the rule then fires only when the whole sample
 * is tiny).  Round 6 (ABI 6): per sample instead of per batch, weight-aware; round 5 flagged every channel with max |x| < tiny over the batch.
 * scratch: bfsr_channel_range_scratch(B, C) floats, private to the stream of the call. */
long long bfsr_channel_range_scratch(int B, int C);
int bfsr_channel_range_check(const float* x, long long x_bs, int B, int C, int H, int W, float tiny, float huge, float ratio, const float* gain,
                             float* scratch, unsigned* flag, void* stream);

/* ---- fused flow-step pointwise chain -----------------------------------------------------------
 * One read of z / h_aff / h_ft, one write of z (the HBM-roofline "coupling inverse" kernel of
 * BASELINE.json).  replaces, per FlowStep (SRFlow-LP/code/models/modules/):
 *   FlowAffineCouplingsAblation.py:57-97  (cross split, sigmoid(h+2)+1e-4, affine apply)
 *   Permutations.py:44-58                 (1x1 invertible conv; `winv`/`w` precomputed by caller)
 *   FlowActNorms.py:61-113                (actnorm)
 * reverse (decode, FlowStep.py:113-129):
 *   if h_aff: z2 = z2/scale - shift            (shift,scale = h_aff[0::2], sigmoid(h_aff[1::2]+2)+eps)
 *   if h_ft : z  = z/scaleFt - shiftFt         (from h_ft, same split, all C channels)
 *   if w    : z  = w @ z  (w = fl32(inv(fl64(W))))    if an_bias: z = z*exp(-logs) - bias
 * forward (encode, FlowStep.py:88-111), h_aff belongs to the PREVIOUS step's self-conditional:
 *   if h_aff: z2 = (z2+shift)*scale
 *   if an_bias: z = (z+bias)*exp(logs)         if w: z = w @ z
 *   if h_ft : z = (z+shiftFt)*scaleFt
 * `an_escale` holds exp(-logs) (reverse) or exp(logs) (forward), precomputed by the caller.
 * z_in and z_out may alias (in-place).  C in {3..128}; h_aff has 2*(C - C/2) channels, h_ft 2*C.
 */
typedef struct BfsrFlowArgs {
    const float* z_in; long long z_in_bs;
    float* z_out; long long z_out_bs;
    const float* h_aff; long long h_aff_bs;
    const float* h_ft; long long h_ft_bs;
    const float* w;            /* [C][C] row-major, or NULL */
    const float* wt;           /* optional transpose of w ([k][i]); enables the MFMA path for C = 96 */
    const float* an_bias;      /* [C] or NULL */
    const float* an_escale;    /* [C] */
    int B, C, H, W;
    int reverse;
    float eps;                 /* affine_eps, 1e-4 */
} BfsrFlowArgs;
int bfsr_flow_pointwise(const BfsrFlowArgs* a, void* stream);

/* ---- the sequential part of a conditional-affine FlowStep in two kernels (coupling.hip, conv_h2s.hip) -------------------------
 * replaces, per coupled step of a level with C in {12, 24} flow channels (FlowAffineCouplingsAblation.py:57-135):
 *   bfsr_coupling_head: hid = relu(AN2(W2 . relu(AN0(conv3x3(z[:, :Cz]; W0z) + pre_aff))))     (fAffine.0 on the z1 rows + the hoisted
 *                       ft partial, fAffine.2; flow.Conv2d = conv without bias + ActNorm, flow.py:26-65); two-term fp16 split (three
 *                       products, fp32 accumulation), the 1x1 chained in registers; hid leaves as an h2 tensor.
 *   bfsr_coupling_tail: h_aff = (conv3x3(hid; W4) + b4) * exp(3*logs4)  (fAffine.4 = Conv2dZeros, flow.py:68-83) on the LDS-DMA kernel of
 *                       bfsr_conv3x3_h2x, then -- as that kernel's epilogue -- the pointwise chain of bfsr_flow_pointwise with that h_aff
 *                       (same argument meaning: h_ft, wmat, an_bias / an_escale, reverse, eps; z_in and z_out may alias). */
typedef struct BfsrCouplingHeadArgs {
    const float* z; long long z_bs; int Cz;
    const float* pre_aff; long long pre_aff_bs;
    int pre_fmt;                                      /* 0: pre_aff [B,64,H,W]; 1: quad-major [B][16][H][W][4] (a 16-byte load per channel quad) */
    const unsigned short* w;                          /* bfsr_pack_coupling_head */
    const float* epi0; const float* epi2;             /* [64] float4 {ActNorm shift, scale, 0, 0} of fAffine.0 / fAffine.2 */
    float acc_scale0, acc_scale2;                     /* 1 / the power-of-two scales the weights were packed with */
    unsigned short* hid; long long hid_bs;            /* OUT: h2 tensor [B][8][2 planes hi,lo][H][W][8] fp16; batch stride in fp16 elements */
    int B, H, W;
    unsigned* flag;                                   /* optional device word: bit 0 is set when a value handed to the fp16 split is >= 2^15 or NaN */
} BfsrCouplingHeadArgs;
typedef struct BfsrCouplingTailArgs {
    const unsigned short* hid; long long hid_bs; int Cin;            /* h2 tensor written by bfsr_coupling_head; Cin = 64 */
    const unsigned short* w; float acc_scale;                        /* bfsr_pack_conv_weight_h2x(fAffine.4, mtile 1, scale); acc_scale = 1/scale */
    const float* bias; const float* post_scale;                      /* [2*(C-C/2)] each: Conv2dZeros bias and exp(3*logs) */
    const float* z_in; long long z_in_bs;
    float* z_out; long long z_out_bs;                                /* may alias z_in (every lane reads and writes its own pixel) */
    const float* h_ft; long long h_ft_bs;
    int h_ft_fmt;                                                    /* 0: h_ft [B,2C,H,W]; 1: quad-major [B][2C/4][H][W][4] */
    const float* wmat; const float* an_bias; const float* an_escale;
    int B, C, H, W, reverse;
    float eps;
    unsigned* flag;                                                  /* optional device word: bit 1 is set when the flow state leaves the finite range */
} BfsrCouplingTailArgs;
int bfsr_coupling_head(const BfsrCouplingHeadArgs* a, void* stream);
int bfsr_coupling_tail(const BfsrCouplingTailArgs* a, void* stream);
long long bfsr_coupling_head_packed_size(int Cz);                                   /* fp16 elements */
int bfsr_pack_coupling_head(const float* w0_z1, const float* w2, int Cz, float scale0, float scale2, unsigned short* packed);
long long bfsr_coupling_tail_packed_size(int Cin, int Cout);                        /* fp16 elements; Cin = 64, Cout <= 32 */
int bfsr_pack_coupling_tail(const float* w4, int Cin, int Cout, float scale, unsigned short* packed);   /* w4 [Cout][64][3][3] * scale, fp16 hi/lo */
/* bfsr_conv3x3_h2r: the tail's conv kernel (one-octet chunks, four-stage LDS-DMA ring, resident weights) as a plain 3x3 'same' conv
 * 64 -> Cout <= 32 over an h2 tensor -- the Conv2dZeros of the hoisted fFeatures nets (flow.py:68-83, FlowAffineCouplingsAblation.py:127-135).
 * a->x h2 view (Cin = 64), a->w = bfsr_pack_coupling_tail(w, 64, Cout, scale), a->acc_scale = 1/scale, a->y fp32 [B,Cout,H,W] (y_fmt 0) or
 * quad-major [B][Cout/4][H][W][4] (y_fmt 2); a->epi / act / slope as for bfsr_conv2d; no residuals.
 * bfsr_coupling_head with Cz = 0 (w0 = NULL at pack time, z = NULL) is the matching producer: hid = relu(AN2(W2 . relu(AN0(pre_aff)))). */
int bfsr_conv3x3_h2r(const BfsrConvX3Args* a, void* stream);

/* ---- the coupled FlowStep of the WIDE level (C = 96 flow channels, level 3 of both SRFlow-LP models) as two streaming kernels (round 6,
 * coupling_wide.hip).  Same reference code as bfsr_coupling_head / _tail (FlowAffineCouplingsAblation.py:57-135, FlowStep.py:88-129,
 * flow.py:26-83, Permutations.py:37-58, FlowActNorms.py:61-113); what differs is the machine mapping: fAffine.0 on 48 z1 channels and fAffine.4
 * (64 -> 96) do not fit LDS, so both kernels stream their weights chunk by chunk the way bfsr_conv3x3_h2x does (same two-term fp16 split, same
 * summation order per output pixel: the conv results are bit-identical to that entry point's), with ALL output channels of an 8 x 32 tile in one
 * workgroup.
 *   bfsr_coupling_wide_head: hid = relu(AN2(W2 . relu(AN0(conv3x3(z1; W0z) + pre))))  -- z1: h2 view of the step's first Cz = 48 channels
 *       (written by the previous step's tail, or by bfsr_h2_pack), pre: h2 view (64 channels) of the hoisted partial, read as hi + lo;
 *       w0 = bfsr_pack_coupling_wide_conv(fAffine.0[:, :Cz], 64, Cz, scale0), w2 = bfsr_pack_coupling_wide_w2(fAffine.2, scale2), epi0 / epi2 as
 *       for bfsr_coupling_head; hid: h2 tensor (64 channels).
 *   bfsr_coupling_wide_tail: h_aff = (conv3x3(hid; W4) + bias) * post_scale, then the pointwise chain of bfsr_flow_pointwise with that h_aff
 *       (same argument meaning: z_in / z_out (may alias), h_ft (+ h_ft_fmt 1: quad-major [B][2C/4][H][W][4]), an_bias / an_escale, reverse, eps);
 *       the C x C matrix is passed as wperm = bfsr_pack_coupling_wide_wmat(W) (K axis in the order the kernel contracts it; NULL = no matvec,
 *       forward only: the last step of a level); w = bfsr_pack_coupling_wide_conv(fAffine.4, 96, 64, scale), acc_scale = 1/scale.
 *       z1h (optional): the first 48 channels of the result once more as an h2 view -- the next step's z1.
 *   flag: bit 0 = a value handed to the fp16 split is out of range, bit 1 = the flow state left the finite range. */
typedef struct BfsrWideHeadArgs {
    const unsigned short* z1; long long z1_bs; int Cz;
    const unsigned short* w0; float acc_scale0;
    const unsigned short* pre; long long pre_bs;
    const unsigned short* w2; float acc_scale2;
    const float* epi0; const float* epi2;             /* [64] float4 {ActNorm shift, scale, 0, 0} of fAffine.0 / fAffine.2 */
    unsigned short* hid; long long hid_bs;
    int B, H, W;
    unsigned* flag;
} BfsrWideHeadArgs;
typedef struct BfsrWideTailArgs {
    const unsigned short* hid; long long hid_bs;
    const unsigned short* w; float acc_scale;
    const float* bias; const float* post_scale;       /* [96] each: Conv2dZeros bias and exp(3*logs) */
    const float* z_in; long long z_in_bs;
    float* z_out; long long z_out_bs;
    const float* h_ft; long long h_ft_bs; int h_ft_fmt;
    const float* wperm; const float* an_bias; const float* an_escale;
    unsigned short* z1h; long long z1h_bs;
    int B, C, H, W, reverse;
    float eps;
    unsigned* flag;
} BfsrWideTailArgs;
int bfsr_coupling_wide_head(const BfsrWideHeadArgs* a, void* stream);
int bfsr_coupling_wide_tail(const BfsrWideTailArgs* a, void* stream);
long long bfsr_coupling_wide_conv_packed_size(int Cout, int Cin);                    /* fp16 elements; Cout in {32, 64, 96}, Cin % 16 == 0 */
int bfsr_pack_coupling_wide_conv(const float* w_oihw, int Cout, int Cin, float scale, unsigned short* packed);
long long bfsr_coupling_wide_w2_packed_size(void);                                   /* fp16 elements */
int bfsr_pack_coupling_wide_w2(const float* w2, float scale2, unsigned short* packed);   /* w2 [64][64] */
int bfsr_pack_coupling_wide_wmat(const float* w, float* wperm);                       /* w [96][96] row-major -> wperm [96][96] */

/* bfsr_conv2d_up2_h2t: the first 3x3 conv of a level's conditioning nets over torch.cat([key, F.interpolate(taps, mode='nearest')]) (key: Ckey
 * channels at the output resolution 2h x 2w; taps: Ct channels at h x w -- the stacked RRDB block outputs), evaluated at the SOURCE resolution by
 * output parity, fp32-class accuracy (two-term fp16 split, three products).  SRFlowNet_arch.py:122-137, RRDBNet_arch.py:105-109,
 * FlowAffineCouplingsAblation.py:108-135.  Taps: the 2x2 source pixels of each parity with pre-summed window weights (16 instead of 36 tap
 * products); key channels: as space-to-depth planes at source resolution (bfsr_h2_pack_s2d), 9 single window taps per plane.
 *   x: h2 view [B][Cin/8][2][h][w][8] (x_bs in fp16 elements) with Cin = Ct + 4*Ckey channels: the taps first, then plane q = qy*2+qx of key
 *      channel c at channel Ct + q*Ckey + c; Ct % 16 == 0, Ckey % 16 == 0 (Ckey = 0: no key channels -- they may instead be convolved separately
 *      and enter through pre_add);  w = bfsr_pack_conv_up2_h2t(w_taps, w_key, Cout, Ct, Ckey, scale), acc_scale = 1/scale (scale: the power of
 *      two that puts the largest packed |w| into [2^9, 2^10));
 *   y, pre_add: fp32 QUAD-MAJOR [B][Cout/4][2h][2w][4] (y_fmt must be 1; batch strides in floats; pre_add may be NULL or alias y), Cout % 32 == 0:
 *   y = conv / scale + pre_add.  One persistent workgroup per CU; item = 16 x 32 source pixels x 32 output channels x all four parities. */
typedef struct BfsrUp2H2Args {
    const unsigned short* x; long long x_bs; int Cin; int Ckey;
    const unsigned short* w; float acc_scale;
    float* y; long long y_bs; int Cout; int y_fmt;
    const float* pre_add; long long pre_add_bs;
    int B, h, w_;                                  /* SOURCE height / width (w_ : `w` is the weight pointer) */
} BfsrUp2H2Args;
int bfsr_conv2d_up2_h2t(const BfsrUp2H2Args* a, void* stream);
long long bfsr_conv_up2_h2t_packed_size(int Cout, int Ct, int Ckey);                 /* fp16 elements */
int bfsr_pack_conv_up2_h2t(const float* w_taps_oihw, const float* w_key_oihw, int Cout, int Ct, int Ckey, float scale, unsigned short* packed);
/* bfsr_conv2d_up4_h2t (round 5): the same for a NEAREST-x4 upsampling (the level-1 conditioning of the 8x model, BASELINE config 4;
 * RRDBNet_arch.py:105-112 fea_up4 + SRFlowNet_arch.py:122-137): x = h2 tensor of the Ct tap channels at SOURCE resolution (Cin = Ct, Ckey must be 0:
 * channels at output resolution enter through pre_add); w = bfsr_pack_conv_up4_h2t(w_taps, Cout, Ct, scale) -- per axis the phases {0}, {1, 2}, {3}
 * of an output pixel see 2, 1, 2 source pixels: 25 pre-summed weight blocks per 16-channel chunk instead of 16 x 9 tap products;
 * y, pre_add: fp32 QUAD-MAJOR [B][Cout/4][4h][4w][4] (y_fmt 1).  y_fmt 3 (ABI 5; layout of ABI 8): COMPACT output [B][Cout/4][h][9][w][4] fp32 -- the nine phase-class
 * values per source pixel and channel quad, one row of w float4 per (source row, class) (class = rc*3 + cc, rc / cc = 0, 1, 2 for output phases {0}, {1, 2}, {3}), no pre_add: 9/16 of the bytes, and the
 * consumer (bfsr_conv3x3_h2x with `up4`) adds it while writing the full-resolution tensor.  Item = 8 x 32 source pixels x 32 output channels x all nine classes. */
int bfsr_conv2d_up4_h2t(const BfsrUp2H2Args* a, void* stream);
long long bfsr_conv_up4_h2t_packed_size(int Cout, int Ct);                             /* fp16 elements */
int bfsr_pack_conv_up4_h2t(const float* w_taps_oihw, int Cout, int Ct, float scale, unsigned short* packed);
/* fp32 [B][C][2h][2w] view -> h2 view with 4C channels at h x w (space to depth): channel q*C + c = pixels (2y+qy, 2x+qx) of channel c, q = qy*2+qx.
 * flag as for bfsr_h2_pack. */
int bfsr_h2_pack_s2d(const float* x, long long x_bs, unsigned short* y, long long y_bs, int B, int C, int h, int w, unsigned* flag, void* stream);

/* squeeze2d / unsqueeze2d, factor 2 (flow.py:122-152): x [B,C,H,W] <-> y [B,4C,H/2,W/2] */
int bfsr_squeeze2d(const float* x, long long x_bs, float* y, long long y_bs, int B, int C, int H, int W,
                   void* stream);
int bfsr_unsqueeze2d(const float* x, long long x_bs, float* y, long long y_bs, int B, int C, int H, int W,
                     void* stream);

/* Split2d (Split.py:48-77) given h = Conv2dZeros(z1) [B,2*Cc,H,W]: mean,logs = h[0::2], h[1::2]
 *   forward: eps = (z2 - mean) / exp(logs)
 *   reverse: z2  = mean + exp(logs) * eps      */
int bfsr_split2d(const float* h, long long h_bs, const float* src, long long src_bs, float* dst,
                 long long dst_bs, int B, int Cc, int H, int W, int reverse, void* stream);

/* likelihood terms of the flow (parity with the `logdet` / `nll` return values, SRFlowNet_arch.py:83-116,145-158).
 * Both add coef * (per-sample sum) into out[b] (double, atomically; zero it before the first call):
 *   bfsr_logscale_sum : sum over j < Cs and pixels of log(sigmoid(h[2j+1] + 2) + eps)   -- the affine couplings'
 *                       get_logdet(scale) (FlowAffineCouplingsAblation.py:66,75,86,92); h = Conv2dZeros output, cross split
 *   bfsr_gaussian_logp: GaussianDiag.logp (flow.py:86-107): h == NULL: sum -0.5*(x^2 + log 2pi); else mean,logs =
 *                       h[2c], h[2c+1]: sum -0.5*(2*logs + (x-mean)^2/exp(2*logs) + log 2pi)   (Split.py:56,74,77-80) */
int bfsr_logscale_sum(const float* h, long long h_bs, int B, int Cs, long long HW, float eps, double coef, double* out,
                      void* stream);
int bfsr_gaussian_logp(const float* x, long long x_bs, const float* h, long long h_bs, int B, int C, long long HW,
                       double coef, double* out, void* stream);

/* per-pixel channel standardisation (SRFlow-LP/code/test.py:141-145): (e-mean_c)/(std_c(unbiased)+1e-8) */
int bfsr_standardize(const float* x, long long x_bs, float* y, long long y_bs, int B, int C, int H, int W,
                     void* stream);

/* resampling.  mode: 0 nearest (src = floor(dst*in/out), F.interpolate default),
 * 1 bilinear align_corners=False with ratio r_h,r_w (src=(dst+0.5)*r-0.5 clamped at 0),
 * 2 bilinear align_corners=True (src = dst*(in-1)/(out-1)).
 * call sites: SRFlow-LP/code/test.py:137, RRDBNet_arch.py:138, SRFlowNet_arch.py:137,
 * models/unet.py:80 (both), LINF-LP/test.py:149,171.  The output may be a padded sub-window:
 * (oy0, ox0) is where the resized image starts inside the (OH, OW) output view, the rest is
 * zero-filled (F.pad in unet.py:86-92). */
int bfsr_resize(const float* x, long long x_bs, int IH, int IW, float* y, long long y_bs, int OH, int OW,
                int RH, int RW, int oy0, int ox0, int B, int C, int mode, float r_h, float r_w, void* stream);

/* MaxPool2d(2) (models/unet.py:63), floor mode */
int bfsr_maxpool2(const float* x, long long x_bs, float* y, long long y_bs, int B, int C, int H, int W,
                  void* stream);

/* y = clamp(a*x + b + (r ? r : 0), lo, hi) elementwise over a view (test.py:150, LINF-LP/test.py:217) */
int bfsr_axpb_clamp(const float* x, long long x_bs, const float* r, long long r_bs, float* y, long long y_bs,
                    int B, int C, int H, int W, float a, float b, float lo, float hi, void* stream);

/* ---- LINF-LP ------------------------------------------------------------------------------------------
 * local-ensemble Fourier features (LINF-LP/models/linf.py:344-388): for each query point and each of the 4
 * neighbouring LR cells (coord shifted by +-half a cell + 1e-6, clamped, nearest lookup) emit
 * w * coef (.) [cos(pi f) | sin(pi f)], f = freq . rel_coord + phase(rel_cell); ensemble weights are the
 * diagonally swapped areas.  cf = [coef | freq] ([B, 2*hidden, h, w], one conv), coord [B,qh,qw,2] (y,x),
 * cell [B,2], phase [hidden/2, 2]; out [B, 4*hidden, qh, qw].  The four shifts (vx*rx+1e-6 as float) and the
 * clamp bounds are passed in so the host computes them exactly like the reference (double -> float). */
typedef struct BfsrLinfFeatArgs {
    const float* cf; long long cf_bs;
    const float* coord; const float* cell; const float* phase;
    float* out; long long out_bs;
    int B, hidden, h, w, qh, qw;
    float dy_neg, dy_pos, dx_neg, dx_pos, clamp_lo, clamp_hi;
    float cy0, cy1, cx0, cx1;      /* LR cell centres: cy0 + cy1*i = float(-1+1/h) + float(2/h)*i (utils.py:113-115) */
} BfsrLinfFeatArgs;
int bfsr_linf_features(const BfsrLinfFeatArgs* a, void* stream);

/* Fused per-point conditioning (LINF-LP/models/linf.py:324-391): the Fourier features of bfsr_linf_features feed the shared MLP
 * (`self.layers`: 1x1 convs 4*hidden -> hidden -> hidden -> hidden -> Cout with ReLU, linf.py:231-240,313-314) inside one kernel;
 * the 4*hidden-channel feature tensor and the hidden activations never leave the chip.  out = affine_info [B, Cout, qh, qw].
 * hidden must be 256; Cout = 2*D*layers (540).  `wts` from bfsr_pack_linf_mlp (x3 != 0: exact 3-term bf16 split of the weights and
 * activations = fp32-accurate; x3 == 0: operands rounded to fp16, LINF precision='fp16'); bias = [b1 | b2 | b3 | b4] (3*hidden + Cout).
 * Geometry fields as in BfsrLinfFeatArgs. */
typedef struct BfsrLinfMlpArgs {
    const float* cf; long long cf_bs;
    const float* coord; const float* cell; const float* phase;
    const unsigned short* wts; const float* bias;
    float* out; long long out_bs;
    int B, hidden, Cout, h, w, qh, qw;
    float dy_neg, dy_pos, dx_neg, dx_pos, clamp_lo, clamp_hi;
    float cy0, cy1, cx0, cx1;
    int out_fmt;                   /* 0: out [B,Cout,qh,qw].  1: quad-major [B][Cout/4][qh*qw][4] (Cout % 4 == 0): the lane that holds
                                    * four consecutive output rows of a query point stores them as ONE 16-byte word (4x fewer store
                                    * instructions, full 64-byte sectors); read by bfsr_linf_flow with ai_fmt = 1 */
    float acc_scale[4];            /* x3 == 2 only: 1 / (the power of two layer i's weights were packed with, bfsr_pack_linf_mlp_f16x2) */
    unsigned* flag;                /* x3 == 2 only, optional device word: bit 0 is set when a feature or hidden activation handed to the fp16 split is >= 65504 (ABI 3) */
    int cf_fmt;                    /* (ABI 4) 0: cf = fp32 [B,2*hidden,h,w], cf_bs in floats.  1: cf = h2 tensor [B][2*hidden/8][hi, lo][h][w][8] fp16 (what
                                    * bfsr_conv3x3_h2s / _h2x write with y_fmt = 1), cf_bs in fp16 elements: the 8 channels of a block are one 16-byte
                                    * gather per plane instead of eight 4-byte ones (value = hi + lo: 22 bits) */
    int tile;                      /* (ABI 7) query points per workgroup tile: 0 or 64, or 128 (x3 = 0 / 2 only): every weight fragment pulled from L2 feeds four
                                    * MFMAs instead of two; identical results */
} BfsrLinfMlpArgs;
/* x3: 0 = operands rounded to fp16 (LINF precision='fp16'); 1 = exact three-term bf16 split, six products; 2 = two-term fp16 split of
 * both operands, three products (fp32-class accuracy at half the matrix instructions of 1; weights from bfsr_pack_linf_mlp_f16x2) */
int bfsr_linf_mlp(const BfsrLinfMlpArgs* a, int x3, void* stream);
long long bfsr_linf_mlp_packed_size(int hidden, int Cout, int x3);          /* in 16-bit elements */
int bfsr_pack_linf_mlp(const float* w1, const float* w2, const float* w3, const float* w4, int hidden, int Cout, int x3,
                       unsigned short* packed);
int bfsr_pack_linf_mlp_f16x2(const float* w1, const float* w2, const float* w3, const float* w4, int hidden, int Cout,
                             const float* scales4, unsigned short* packed);   /* size: bfsr_linf_mlp_packed_size(hidden, Cout, 2) */

/* local implicit coupling flow (LINF-LP/models/flow.py:44-63) over D = 3*ps*ps vectors on the query grid:
 * x,y [B,D,qh,qw]; ai = affine_info [B, 2*D*layers, qh, qw]; lin_w [layers+1][D][D] holds W (forward) or
 * inv(W) (reverse, precomputed by the caller), lin_b [layers+1][D]; last entry = `last` linear.
 * reverse: 0 = forward, 1 = inverse, 2 = vector-Jacobian product of the inverse w.r.t. its input (x = upstream gradient
 * [B,D,qh,qw], lin_w[i] = inv(W_i)^T, lin_b unused): the backward of query_rgb into the latent (LINF-LP/train.py:143). */
typedef struct BfsrLinfFlowArgs {
    const float* x; long long x_bs;
    const float* ai; long long ai_bs;
    float* y; long long y_bs;
    const float* lin_w; const float* lin_b;
    int B, D, layers, qh, qw, reverse;
    float eps;
    float* log_p;              /* optional (forward only): [B][qh*qw] total log-det + base log-prob per query point
                                * (flow.py:44-55); logdet_const = sum over the layers+1 linears of slogdet(W)[1] */
    float logdet_const;
    int ai_fmt;                /* 0: ai [B, 2*D*layers, qh, qw].  1: quad-major [B][layers][2*S/4][qh*qw][4] with S = D rounded up to a multiple of
                                * four: per layer S raw scales (D used) then S shifts (D used), so both halves start on a quad (round 6: before, the
                                * shifts followed the scales directly); the producer's rows are laid out in that padded order.  With D = 27 the
                                * forward (log_p == NULL) and inverse passes run on the fp32 matrix pipe in this format. */
} BfsrLinfFlowArgs;
int bfsr_linf_flow(const BfsrLinfFlowArgs* a, void* stream);

/* F.fold of ps x ps patches (linf.py:401-406) + crop: p [B,C*ps*ps,qh,qw] -> img [B,C,H,W], H <= ps*qh */
int bfsr_patch_fold(const float* p, long long p_bs, float* img, long long img_bs, int B, int C, int qh, int qw,
                    int H, int W, int ps, void* stream);
/* zero-pad + unfold (datasets/wrappers.py:224-228): img [B,C,H,W] -> p [B,C*ps*ps,qh,qw] */
int bfsr_patch_unfold(const float* img, long long img_bs, float* p, long long p_bs, int B, int C, int qh, int qw,
                      int H, int W, int ps, void* stream);
/* ---- h2 forms of the learned priors' glue (ABI 8, round 6; resample.hip): bfsr_resize writing an h2 tensor directly (= bfsr_resize + bfsr_h2_pack, `flag` as in
 * bfsr_h2_pack) and the 2 x 2 max-pool of an h2 tensor into an h2 tensor and / or an fp32 NCHW tensor (= bfsr_h2_unpack + bfsr_maxpool2 [+ bfsr_h2_pack]); the same bits
 * as those launches.  models/unet.py:58-98 (`Down`: MaxPool2d(2), `Up`: Upsample(bilinear, align_corners=True)) of both learned priors. */
int bfsr_resize_h2(const float* x, long long x_bs, int IH, int IW, unsigned short* y, long long y_bs, int OH, int OW, int RH, int RW, int oy0, int ox0,
                   int B, int C, int mode, float r_h, float r_w, unsigned* flag, void* stream);
int bfsr_maxpool2_h2(const unsigned short* x, long long x_bs, unsigned short* y_h2, long long yh_bs, float* y_f32, long long yf_bs, int B, int C, int H, int W,
                     unsigned* flag, void* stream);
/* ---- LINF-LP harness glue, fused (ABI 8, round 6; resample.hip).  Bit-identical to the sequences of bfsr_resize / bfsr_axpb_clamp / bfsr_patch_fold /
 * bfsr_patch_unfold launches they replace (the same float operations in the same order).
 * bfsr_linf_fold_skip: the tail of LINF-LP/test.py:168-171, 217 -- raw = fold(p)[.., :H, :W] + F.interpolate(inp, (H, W), bilinear) and
 *   out01 = clamp(0.5 raw + 0.5, 0, 1); p [B,C*ps*ps,qh,qw] (the inverse flow's output), inp [B,C,h,w] (normalised LR), raw / out01 [B,C,H,W] (either may be
 *   NULL), r_h = h / H, r_w = w / W as floats (the ratios bfsr_resize takes).
 * bfsr_linf_prep_down / bfsr_linf_prep_residual: the input prep of datasets/wrappers.py:203-228 (`SRImplicitPairedFastPatch`) from inp01 [B,C,h,w] in [0,1] alone:
 *   lr_up = bilinear(2 inp01 - 1 -> H x W) is never stored; down [B,C,h,w] = bilinear(lr_up -> h x w) (ratios rd_*: H / h, W / w);
 *   gt [B,C*ps*ps,qh,qw] = unfold(zero-pad(lr_up - bilinear(down -> H x W))) (ratios ru_*: h / H, w / W; ps = 1: the pixel-wise wrapper's residual image). */
int bfsr_linf_fold_skip(const float* p, long long p_bs, const float* inp, long long inp_bs, float* raw, long long raw_bs, float* out01, long long out_bs,
                        int B, int C, int qh, int qw, int H, int W, int ps, int h, int w, float r_h, float r_w, void* stream);
int bfsr_linf_prep_down(const float* inp01, long long in_bs, float* down, long long down_bs, int B, int C, int h, int w, int H, int W,
                        float ru_h, float ru_w, float rd_h, float rd_w, void* stream);
int bfsr_linf_prep_residual(const float* inp01, long long in_bs, const float* down, long long down_bs, float* gt, long long gt_bs, int B, int C, int h, int w,
                            int H, int W, int qh, int qw, int ps, float ru_h, float ru_w, void* stream);
/* out = acc + F.grid_sample(x, coord.flip(-1), bilinear, padding_mode='border', align_corners=False): the in-method
 * LR skip of the pixel-wise LINF (LINF-LP/models/linf.py:193-194); x [B,C,h,w], coord [B,qh,qw,2] (y,x), acc/out [B,C,qh,qw] */
int bfsr_grid_sample_add(const float* x, long long x_bs, const float* coord, const float* acc, long long acc_bs, float* out,
                         long long out_bs, int B, int C, int h, int w, int qh, int qw, void* stream);
/* small direct strided conv (+bias, +activation): LINF prior `lr_proj.0` (LINF-LP/models/unet.py:118) */
int bfsr_conv2d_direct(const float* x, long long x_bs, const float* w, const float* bias, float* y, long long y_bs,
                       int B, int Cin, int Cout, int H, int W, int KS, int stride, int pad, int act, float slope,
                       void* stream);

/* ---- evaluation metrics / output formatting on the device (LINF-LP/test.py:172-225) -------------
 * bfsr_resample_taps: one pass of a separable resampler, y = sum_p w[o][p] * x[..idx[o][p]..] along dim (0 = rows, 1 = cols);
 *   idx/w [O][P] are the host-built tables of MATLAB-style `imresize` (imresize.py:64-88 `contributions`, :110-121).
 * bfsr_sqdiff_sum: out[b] += sum over the window shaved by `shave` of ((a-b)/rgb_range)^2; luma != 0 first reduces the
 *   3 channels with [65.738,129.057,25.064]/256 (calc_psnr 'benchmark', utils.py:132-149).  Zero `out` before the call.
 * bfsr_ssim_sum: out[b*C+c] += sum of the SSIM map over the 'valid' (H-10)x(W-10) region, 11x11 window `window121`
 *   (outer product of the Gaussian sigma 1.5), images multiplied by `scale` first (255 for [0,1] data), fp64 (utils.py:152-171).
 * bfsr_to_uint8: y[b][i] = uint8(rint(clamp(x,0,1)*255)) for i < n (test.py:210-212). */
int bfsr_resample_taps(const float* x, long long x_bs, float* y, long long y_bs, const int* idx, const float* w,
                       int B, int C, int H, int W, int O, int P, int dim, void* stream);
int bfsr_sqdiff_sum(const float* a, long long a_bs, const float* b, long long b_bs, int B, int C, int H, int W, int shave,
                    int luma, float rgb_range, double* out, void* stream);
int bfsr_ssim_sum(const float* a, long long a_bs, const float* b, long long b_bs, int B, int C, int H, int W, double scale,
                  const double* window121, double* out, void* stream);
/* bfsr_ssim_sum_w: bfsr_ssim_sum with a caller-given ws x ws window (ws <= 11, 'valid' region (H-ws+1) x (W-ws+1)) and the variances /
 * covariance multiplied by cov_norm -- ws = 7, window = 1/49, cov_norm = 49/48 on 0..255 images (scale 1) is
 * skimage.metrics.structural_similarity(imgA, imgB, multichannel=True) for uint8 inputs (uniform 7x7 filter, sample covariance, data_range 255,
 * mean over the image cropped by 3), the call of SRFlow-LP/code/Measure.py:45-48. */
int bfsr_ssim_sum_w(const float* a, long long a_bs, const float* b, long long b_bs, int B, int C, int H, int W, double scale,
                    int ws, const double* window, double cov_norm, double* out, void* stream);
int bfsr_to_uint8(const float* x, long long x_bs, unsigned char* y, int B, long long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
