"""ctypes binding of libbfsr_hip.so (the C ABI declared in include/bfsr_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a RuntimeError
is raised.  Build it with `bfsr_amd/csrc/build.sh` (or `__graft_entry__.build()`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BFSR_HIP_LIB") or os.path.join(_HERE, "lib", "libbfsr_hip.so")

c_float_p = C.POINTER(C.c_float)


class BfsrConvArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_bs", C.c_longlong), ("Cin", C.c_int),
        ("w", C.c_void_p),
        ("y", C.c_void_p), ("y_bs", C.c_longlong), ("Cout", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("KS", C.c_int), ("in_shift", C.c_int),
        ("mtile", C.c_int),
        ("epi", C.c_void_p),
        ("pre_add", C.c_void_p), ("pre_add_bs", C.c_longlong),
        ("act", C.c_int), ("slope", C.c_float),
        ("res1", C.c_void_p), ("res1_bs", C.c_longlong), ("alpha1", C.c_float),
        ("res2", C.c_void_p), ("res2_bs", C.c_longlong), ("alpha2", C.c_float),
        ("tune", C.c_int),
        ("w2", C.c_void_p), ("C2", C.c_int), ("epi2", C.c_void_p), ("act2", C.c_int),
        ("x2", C.c_void_p), ("x2_bs", C.c_longlong), ("Cin2", C.c_int), ("w_x2", C.c_void_p),
        ("arith", C.c_int), ("acc_scale", C.c_float), ("y_fmt", C.c_int), ("flag", C.c_void_p),
    ]


class BfsrConvX3Args(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_bs", C.c_longlong), ("Cin", C.c_int),
        ("w", C.c_void_p),
        ("y", C.c_void_p), ("y_bs", C.c_longlong), ("Cout", C.c_int), ("y_fmt", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("epi", C.c_void_p), ("act", C.c_int), ("slope", C.c_float),
        ("res1", C.c_void_p), ("res1_bs", C.c_longlong), ("alpha1", C.c_float),
        ("res2", C.c_void_p), ("res2_bs", C.c_longlong), ("alpha2", C.c_float),
        ("tune", C.c_int), ("acc_scale", C.c_float), ("mtile", C.c_int), ("flag", C.c_void_p),
        ("up4", C.c_void_p), ("up4_bs", C.c_longlong),
    ]


class BfsrUp2H2Args(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_bs", C.c_longlong), ("Cin", C.c_int), ("Ckey", C.c_int),
        ("w", C.c_void_p), ("acc_scale", C.c_float),
        ("y", C.c_void_p), ("y_bs", C.c_longlong), ("Cout", C.c_int), ("y_fmt", C.c_int),
        ("pre_add", C.c_void_p), ("pre_add_bs", C.c_longlong),
        ("B", C.c_int), ("h", C.c_int), ("w_", C.c_int),
    ]


class BfsrFlowArgs(C.Structure):
    _fields_ = [
        ("z_in", C.c_void_p), ("z_in_bs", C.c_longlong),
        ("z_out", C.c_void_p), ("z_out_bs", C.c_longlong),
        ("h_aff", C.c_void_p), ("h_aff_bs", C.c_longlong),
        ("h_ft", C.c_void_p), ("h_ft_bs", C.c_longlong),
        ("w", C.c_void_p), ("wt", C.c_void_p), ("an_bias", C.c_void_p), ("an_escale", C.c_void_p),
        ("B", C.c_int), ("C", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("reverse", C.c_int), ("eps", C.c_float),
    ]


class BfsrCouplingHeadArgs(C.Structure):
    _fields_ = [
        ("z", C.c_void_p), ("z_bs", C.c_longlong), ("Cz", C.c_int),
        ("pre_aff", C.c_void_p), ("pre_aff_bs", C.c_longlong), ("pre_fmt", C.c_int),
        ("w", C.c_void_p), ("epi0", C.c_void_p), ("epi2", C.c_void_p),
        ("acc_scale0", C.c_float), ("acc_scale2", C.c_float),
        ("hid", C.c_void_p), ("hid_bs", C.c_longlong),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("flag", C.c_void_p),
    ]


class BfsrCouplingTailArgs(C.Structure):
    _fields_ = [
        ("hid", C.c_void_p), ("hid_bs", C.c_longlong), ("Cin", C.c_int),
        ("w", C.c_void_p), ("acc_scale", C.c_float),
        ("bias", C.c_void_p), ("post_scale", C.c_void_p),
        ("z_in", C.c_void_p), ("z_in_bs", C.c_longlong),
        ("z_out", C.c_void_p), ("z_out_bs", C.c_longlong),
        ("h_ft", C.c_void_p), ("h_ft_bs", C.c_longlong), ("h_ft_fmt", C.c_int),
        ("wmat", C.c_void_p), ("an_bias", C.c_void_p), ("an_escale", C.c_void_p),
        ("B", C.c_int), ("C", C.c_int), ("H", C.c_int), ("W", C.c_int), ("reverse", C.c_int),
        ("eps", C.c_float),
        ("flag", C.c_void_p),
    ]


class BfsrWideHeadArgs(C.Structure):
    _fields_ = [
        ("z1", C.c_void_p), ("z1_bs", C.c_longlong), ("Cz", C.c_int),
        ("w0", C.c_void_p), ("acc_scale0", C.c_float),
        ("pre", C.c_void_p), ("pre_bs", C.c_longlong),
        ("w2", C.c_void_p), ("acc_scale2", C.c_float),
        ("epi0", C.c_void_p), ("epi2", C.c_void_p),
        ("hid", C.c_void_p), ("hid_bs", C.c_longlong),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("flag", C.c_void_p),
    ]


class BfsrWideTailArgs(C.Structure):
    _fields_ = [
        ("hid", C.c_void_p), ("hid_bs", C.c_longlong),
        ("w", C.c_void_p), ("acc_scale", C.c_float),
        ("bias", C.c_void_p), ("post_scale", C.c_void_p),
        ("z_in", C.c_void_p), ("z_in_bs", C.c_longlong),
        ("z_out", C.c_void_p), ("z_out_bs", C.c_longlong),
        ("h_ft", C.c_void_p), ("h_ft_bs", C.c_longlong), ("h_ft_fmt", C.c_int),
        ("wperm", C.c_void_p), ("an_bias", C.c_void_p), ("an_escale", C.c_void_p),
        ("z1h", C.c_void_p), ("z1h_bs", C.c_longlong),
        ("B", C.c_int), ("C", C.c_int), ("H", C.c_int), ("W", C.c_int), ("reverse", C.c_int),
        ("eps", C.c_float),
        ("flag", C.c_void_p),
    ]


class BfsrLinfFeatArgs(C.Structure):
    _fields_ = [
        ("cf", C.c_void_p), ("cf_bs", C.c_longlong),
        ("coord", C.c_void_p), ("cell", C.c_void_p), ("phase", C.c_void_p),
        ("out", C.c_void_p), ("out_bs", C.c_longlong),
        ("B", C.c_int), ("hidden", C.c_int), ("h", C.c_int), ("w", C.c_int), ("qh", C.c_int), ("qw", C.c_int),
        ("dy_neg", C.c_float), ("dy_pos", C.c_float), ("dx_neg", C.c_float), ("dx_pos", C.c_float),
        ("clamp_lo", C.c_float), ("clamp_hi", C.c_float),
        ("cy0", C.c_float), ("cy1", C.c_float), ("cx0", C.c_float), ("cx1", C.c_float),
    ]


class BfsrLinfMlpArgs(C.Structure):
    _fields_ = [
        ("cf", C.c_void_p), ("cf_bs", C.c_longlong),
        ("coord", C.c_void_p), ("cell", C.c_void_p), ("phase", C.c_void_p),
        ("wts", C.c_void_p), ("bias", C.c_void_p),
        ("out", C.c_void_p), ("out_bs", C.c_longlong),
        ("B", C.c_int), ("hidden", C.c_int), ("Cout", C.c_int), ("h", C.c_int), ("w", C.c_int), ("qh", C.c_int), ("qw", C.c_int),
        ("dy_neg", C.c_float), ("dy_pos", C.c_float), ("dx_neg", C.c_float), ("dx_pos", C.c_float),
        ("clamp_lo", C.c_float), ("clamp_hi", C.c_float),
        ("cy0", C.c_float), ("cy1", C.c_float), ("cx0", C.c_float), ("cx1", C.c_float),
        ("out_fmt", C.c_int), ("acc_scale", C.c_float * 4), ("flag", C.c_void_p), ("cf_fmt", C.c_int), ("tile", C.c_int),
    ]


class BfsrLinfFlowArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_bs", C.c_longlong),
        ("ai", C.c_void_p), ("ai_bs", C.c_longlong),
        ("y", C.c_void_p), ("y_bs", C.c_longlong),
        ("lin_w", C.c_void_p), ("lin_b", C.c_void_p),
        ("B", C.c_int), ("D", C.c_int), ("layers", C.c_int), ("qh", C.c_int), ("qw", C.c_int), ("reverse", C.c_int),
        ("eps", C.c_float),
        ("log_p", C.c_void_p), ("logdet_const", C.c_float), ("ai_fmt", C.c_int),
    ]


class BfsrChainConv(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_bs", C.c_longlong), ("Cin", C.c_int),
        ("w", C.c_void_p),
        ("y", C.c_void_p), ("y_bs", C.c_longlong), ("Cout", C.c_int), ("y_fmt", C.c_int),
        ("epi", C.c_void_p), ("act", C.c_int), ("slope", C.c_float),
        ("res1", C.c_void_p), ("res1_bs", C.c_longlong), ("alpha1", C.c_float),
        ("res2", C.c_void_p), ("res2_bs", C.c_longlong), ("alpha2", C.c_float),
        ("acc_scale", C.c_float),
        ("y2", C.c_void_p), ("y2_bs", C.c_longlong),
    ]


# every symbol include/bfsr_hip.h declares: name -> (restype, argtypes)
_LL, _I, _F, _VP = C.c_longlong, C.c_int, C.c_float, C.c_void_p
SYMBOLS = {
    "bfsr_abi_version": (_I, []),
    "bfsr_conv_packed_size": (_LL, [_I, _I, _I, _I]),
    "bfsr_pack_conv_weight": (_I, [_VP, _I, _I, _I, _I, _VP]),
    "bfsr_conv2d": (_I, [C.POINTER(BfsrConvArgs), _VP]),
    "bfsr_conv2d_f16": (_I, [C.POINTER(BfsrConvArgs), _VP]),
    "bfsr_conv_packed_size_f16": (_LL, [_I, _I, _I, _I]),
    "bfsr_pack_conv_weight_f16": (_I, [_VP, _I, _I, _I, _I, _VP]),
    "bfsr_conv2d_bf16x3": (_I, [C.POINTER(BfsrConvArgs), _VP]),
    "bfsr_conv_packed_size_bf16x3": (_LL, [_I, _I, _I, _I]),
    "bfsr_pack_conv_weight_bf16x3": (_I, [_VP, _I, _I, _I, _I, _VP]),
    "bfsr_conv3x3_x3s": (_I, [C.POINTER(BfsrConvX3Args), _VP]),
    "bfsr_x3_pack": (_I, [_VP, _LL, _VP, _LL, _I, _I, _I, _I, _VP]),
    "bfsr_x3_unpack": (_I, [_VP, _LL, _VP, _LL, _I, _I, _I, _I, _VP]),
    "bfsr_conv3x3_h2s": (_I, [C.POINTER(BfsrConvX3Args), _VP]),
    "bfsr_conv_packed_size_h2s": (_LL, [_I, _I]),
    "bfsr_pack_conv_weight_h2s": (_I, [_VP, _I, _I, _VP]),
    "bfsr_conv_packed_size_h2s_mt": (_LL, [_I, _I, _I]),
    "bfsr_pack_conv_weight_h2s_mt": (_I, [_VP, _I, _I, _I, _VP]),
    "bfsr_conv3x3_h2x": (_I, [C.POINTER(BfsrConvX3Args), _VP]),
    "bfsr_conv_packed_size_h2x": (_LL, [_I, _I, _I]),
    "bfsr_pack_conv_weight_h2x": (_I, [_VP, _I, _I, _I, C.c_float, _VP]),
    "bfsr_h2_pack": (_I, [_VP, _LL, _VP, _LL, _I, _I, _I, _I, _VP, _VP]),
    "bfsr_h2_pack_pad": (_I, [_VP, _LL, _VP, _LL, _I, _I, _I, _I, _I, _VP, _VP]),
    "bfsr_h2_unpack": (_I, [_VP, _LL, _VP, _LL, _I, _I, _I, _I, _VP]),
    "bfsr_conv_chain_table_size": (_LL, [_I]),
    "bfsr_conv_chain_prepare": (_I, [C.POINTER(BfsrChainConv), _I, _I, _I, _I, _VP]),
    "bfsr_conv_chain_progress_words": (_LL, [_VP]),
    "bfsr_conv_chain_launch": (_I, [_VP, _VP, _VP, _VP, _I, _VP]),
    "bfsr_conv1x1": (_I, [C.POINTER(BfsrConvArgs), _I, _VP]),
    "bfsr_conv1x1_packed_size": (_LL, [_I, _I, _I]),
    "bfsr_pack_conv1x1_weight": (_I, [_VP, _I, _I, _I, _VP]),
    "bfsr_conv2d_up2_bf16x3": (_I, [C.POINTER(BfsrConvArgs), _VP]),
    "bfsr_conv2d_up4_bf16x3": (_I, [C.POINTER(BfsrConvArgs), _VP]),
    "bfsr_conv_packed_size_taps_bf16x3": (_LL, [_I, _I, _I, _I]),
    "bfsr_pack_conv_weight_taps_bf16x3": (_I, [_VP, _I, _I, _I, _I, _VP]),
    "bfsr_conv_packed_size_taps_f16x2": (_LL, [_I, _I, _I, _I]),
    "bfsr_pack_conv_weight_taps_f16x2": (_I, [_VP, _I, _I, _I, _I, C.c_float, _VP]),
    "bfsr_conv2d_up2": (_I, [C.POINTER(BfsrConvArgs), _VP]),
    "bfsr_conv_packed_size_taps": (_LL, [_I, _I, _I, _I]),
    "bfsr_pack_conv_weight_taps": (_I, [_VP, _I, _I, _I, _I, _VP]),
    "bfsr_flow_pointwise": (_I, [C.POINTER(BfsrFlowArgs), _VP]),
    "bfsr_coupling_head": (_I, [C.POINTER(BfsrCouplingHeadArgs), _VP]),
    "bfsr_coupling_tail": (_I, [C.POINTER(BfsrCouplingTailArgs), _VP]),
    "bfsr_coupling_head_packed_size": (_LL, [_I]),
    "bfsr_pack_coupling_head": (_I, [_VP, _VP, _I, C.c_float, C.c_float, _VP]),
    "bfsr_coupling_tail_packed_size": (_LL, [_I, _I]),
    "bfsr_pack_coupling_tail": (_I, [_VP, _I, _I, C.c_float, _VP]),
    "bfsr_conv3x3_h2r": (_I, [C.POINTER(BfsrConvX3Args), _VP]),
    "bfsr_coupling_wide_head": (_I, [C.POINTER(BfsrWideHeadArgs), _VP]),
    "bfsr_coupling_wide_tail": (_I, [C.POINTER(BfsrWideTailArgs), _VP]),
    "bfsr_coupling_wide_conv_packed_size": (_LL, [_I, _I]),
    "bfsr_pack_coupling_wide_conv": (_I, [_VP, _I, _I, C.c_float, _VP]),
    "bfsr_coupling_wide_w2_packed_size": (_LL, []),
    "bfsr_pack_coupling_wide_w2": (_I, [_VP, C.c_float, _VP]),
    "bfsr_pack_coupling_wide_wmat": (_I, [_VP, _VP]),
    "bfsr_conv2d_up2_h2t": (_I, [C.POINTER(BfsrUp2H2Args), _VP]),
    "bfsr_conv_up2_h2t_packed_size": (_LL, [_I, _I, _I]),
    "bfsr_pack_conv_up2_h2t": (_I, [_VP, _VP, _I, _I, _I, C.c_float, _VP]),
    "bfsr_conv2d_up4_h2t": (_I, [C.POINTER(BfsrUp2H2Args), _VP]),
    "bfsr_conv_up4_h2t_packed_size": (_LL, [_I, _I]),
    "bfsr_pack_conv_up4_h2t": (_I, [_VP, _I, _I, C.c_float, _VP]),
    "bfsr_h2_pack_s2d": (_I, [_VP, _LL, _VP, _LL, _I, _I, _I, _I, _VP, _VP]),
    "bfsr_squeeze2d": (_I, [_VP, _LL, _VP, _LL, _I, _I, _I, _I, _VP]),
    "bfsr_unsqueeze2d": (_I, [_VP, _LL, _VP, _LL, _I, _I, _I, _I, _VP]),
    "bfsr_split2d": (_I, [_VP, _LL, _VP, _LL, _VP, _LL, _I, _I, _I, _I, _I, _VP]),
    "bfsr_standardize": (_I, [_VP, _LL, _VP, _LL, _I, _I, _I, _I, _VP]),
    "bfsr_resize": (_I, [_VP, _LL, _I, _I, _VP, _LL, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _VP]),
    "bfsr_maxpool2": (_I, [_VP, _LL, _VP, _LL, _I, _I, _I, _I, _VP]),
    "bfsr_axpb_clamp": (_I, [_VP, _LL, _VP, _LL, _VP, _LL, _I, _I, _I, _I, _F, _F, _F, _F, _VP]),
    "bfsr_linf_features": (_I, [C.POINTER(BfsrLinfFeatArgs), _VP]),
    "bfsr_linf_mlp": (_I, [C.POINTER(BfsrLinfMlpArgs), _I, _VP]),
    "bfsr_linf_mlp_packed_size": (_LL, [_I, _I, _I]),
    "bfsr_pack_linf_mlp": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "bfsr_pack_linf_mlp_f16x2": (_I, [_VP, _VP, _VP, _VP, _I, _I, _VP, _VP]),
    "bfsr_logscale_sum": (_I, [_VP, _LL, _I, _I, _LL, _F, C.c_double, _VP, _VP]),
    "bfsr_gaussian_logp": (_I, [_VP, _LL, _VP, _LL, _I, _I, _LL, C.c_double, _VP, _VP]),
    "bfsr_linf_flow": (_I, [C.POINTER(BfsrLinfFlowArgs), _VP]),
    "bfsr_patch_fold": (_I, [_VP, _LL, _VP, _LL, _I, _I, _I, _I, _I, _I, _I, _VP]),
    "bfsr_resize_h2": (_I, [_VP, _LL, _I, _I, _VP, _LL, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _VP, _VP]),
    "bfsr_maxpool2_h2": (_I, [_VP, _LL, _VP, _LL, _VP, _LL, _I, _I, _I, _I, _VP, _VP]),
    "bfsr_linf_fold_skip": (_I, [_VP, _LL, _VP, _LL, _VP, _LL, _VP, _LL, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _VP]),
    "bfsr_linf_prep_down": (_I, [_VP, _LL, _VP, _LL, _I, _I, _I, _I, _I, _I, _F, _F, _F, _F, _VP]),
    "bfsr_linf_prep_residual": (_I, [_VP, _LL, _VP, _LL, _VP, _LL, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _VP]),
    "bfsr_patch_unfold": (_I, [_VP, _LL, _VP, _LL, _I, _I, _I, _I, _I, _I, _I, _VP]),
    "bfsr_grid_sample_add": (_I, [_VP, _LL, _VP, _VP, _LL, _VP, _LL, _I, _I, _I, _I, _I, _I, _VP]),
    "bfsr_resample_taps": (_I, [_VP, _LL, _VP, _LL, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _VP]),
    "bfsr_sqdiff_sum": (_I, [_VP, _LL, _VP, _LL, _I, _I, _I, _I, _I, _I, _F, _VP, _VP]),
    "bfsr_ssim_sum": (_I, [_VP, _LL, _VP, _LL, _I, _I, _I, _I, C.c_double, _VP, _VP, _VP]),
    "bfsr_ssim_sum_w": (_I, [_VP, _LL, _VP, _LL, _I, _I, _I, _I, C.c_double, _I, _VP, C.c_double, _VP, _VP]),
    "bfsr_to_uint8": (_I, [_VP, _LL, _VP, _I, _LL, _VP]),
    "bfsr_channel_range_scratch": (_LL, [_I, _I]),
    "bfsr_channel_range_check": (_I, [_VP, _LL, _I, _I, _I, _I, _F, _F, _F, _VP, _VP, _VP, _VP]),
    "bfsr_conv2d_direct": (_I, [_VP, _LL, _VP, _VP, _VP, _LL, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _VP]),
}

_lib = None


def load():
    """Load libbfsr_hip.so once; raise loudly if it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "bfsr_amd: %s not found -- the HIP extension is required (build with "
            "bfsr_amd/csrc/build.sh or __graft_entry__.build()); there is no CPU fallback." % LIB_PATH)
    # torch first: PyTorch-ROCm ships its own libamdhip64 and the streams / allocations this library is handed come from THAT runtime.  Loaded before
    # torch, libbfsr_hip.so binds to /opt/rocm's copy instead -- a second runtime instance without torch's device context: every launch then fails
    # with hipErrorNoDevice (seen with `python __graft_entry__.py smoke`: build() loads the library before smoke() imports torch)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.bfsr_abi_version() != 8:
        raise RuntimeError("bfsr_amd: ABI version mismatch")
    _lib = lib
    return lib


def check(err, what):
    if err != 0:
        raise RuntimeError("bfsr_amd: %s failed with code %d" % (what, err))
