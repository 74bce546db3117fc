"""ctypes binding of libadaptpoint_amd.so (the C ABI in include/adaptpoint_amd.h).

The library is the product: if it is missing or fails to load this module
raises -- there is no CPU or PyTorch fallback behind it.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# APN_LIB_PATH selects another build of the library for A/B measurements and the diagnostic variants of
# scripts/asm_variants.py -- instead of overwriting the shipped file -- and only together with APN_ALLOW_UNSAFE_LIB=1.
LIB_PATH = os.path.join(_HERE, "libadaptpoint_amd.so")
if os.environ.get("APN_LIB_PATH"):
    if os.environ.get("APN_ALLOW_UNSAFE_LIB") != "1":
        raise RuntimeError("APN_LIB_PATH is for diagnostic builds: set APN_ALLOW_UNSAFE_LIB=1 with it")
    LIB_PATH = os.path.abspath(os.environ["APN_LIB_PATH"])
REQUIRED_BUILD_FLAGS = ("-ffp-contract=off", "-fno-slp-vectorize")

_c_int, _c_float, _c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p
_c_double = ctypes.c_double
_c_longlong = ctypes.c_longlong

# name -> argtypes, exactly the prototypes of include/adaptpoint_amd.h
SIGNATURES = {
    "apn_version": [],
    "apn_furthest_point_sampling": [_c_int] * 3 + [_c_void_p] * 4,
    "apn_ball_query": [_c_int] * 3 + [_c_float, _c_int] + [_c_void_p] * 4,
    "apn_group_points": [_c_int] * 5 + [_c_void_p] * 4,
    "apn_group_points_grad": [_c_int] * 5 + [_c_void_p] * 4,
    "apn_gather_points": [_c_int] * 4 + [_c_void_p] * 4,
    "apn_gather_points_grad": [_c_int] * 4 + [_c_void_p] * 4,
    "apn_resample_points": [_c_int] * 6 + [_c_void_p] * 6,
    "apn_three_nn": [_c_int] * 3 + [_c_void_p] * 5,
    "apn_three_interpolate": [_c_int] * 4 + [_c_void_p] * 5,
    "apn_three_interpolate_grad": [_c_int] * 4 + [_c_void_p] * 5,
    "apn_furthest_point_sampling_tuned": [_c_int] * 3 + [_c_void_p] * 3 + [_c_int] * 2 + [_c_void_p],
    "apn_fps_debug_stamps": [_c_int] * 3 + [_c_void_p] * 5,
    "apn_furthest_point_sampling_xyz": [_c_int] * 3 + [_c_void_p] * 5,
    "apn_ball_query_zero": [_c_int] * 3 + [_c_float, _c_int] + [_c_void_p] * 4,
    "apn_sa_grid_blocks": [_c_int] * 2,
    "apn_sa_grid_rows": [_c_int] * 3,
    "apn_sa_bwd_main_rows": [_c_int] * 2,
    "apn_sa_acc_words": [_c_int],
    "apn_sa_debug_stamps": [_c_void_p],
    "apn_zero_fill": [_c_void_p, _c_longlong, _c_void_p],
    "apn_debug_stamp": [_c_void_p, _c_int, _c_void_p],
    "apn_debug_vgpr_hold": [_c_int, _c_int, _c_void_p, _c_void_p],
    "apn_debug_vpk_probe": [_c_int, _c_int, _c_int, _c_void_p, _c_void_p],
    "apn_sa_geo_dd_doubles": [_c_int],
    "apn_sa_point_geo": [_c_int] * 4 + [_c_float] + [_c_void_p] * 6,
    "apn_sa_prep_rows": [_c_int] * 2,
    "apn_sa_prep_stats": [_c_int] * 2 + [_c_void_p] * 4 + [_c_int] * 2 + [_c_void_p] * 3 + [_c_longlong, _c_void_p],
    "apn_sa_reduce_rows": [_c_void_p, _c_int, _c_int, _c_double, _c_void_p, _c_void_p],
    "apn_sa_reduce_acc": [_c_void_p, _c_int, _c_double, _c_void_p, _c_void_p],
    "apn_sa_bn_fold": [_c_void_p, _c_int, _c_void_p, _c_int, _c_double, _c_void_p, _c_void_p,
                       _c_float, _c_float, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p,
                       _c_void_p, _c_int, _c_void_p, _c_void_p],
    "apn_sa_fwd_main": [_c_int] * 4 + [_c_float] + [_c_void_p] * 12 + [_c_float, _c_float, _c_int, _c_double,
                                                                        _c_void_p, _c_int] + [_c_void_p] * 7,
    "apn_sa_fwd_out": [_c_int] * 3 + [_c_void_p] * 8 + [_c_float, _c_float, _c_int, _c_double] + [_c_void_p] * 2
                      + [_c_int] + [_c_void_p] * 3 + [_c_int] + [_c_void_p] * 2 + [_c_longlong, _c_void_p],
    "apn_sa_bwd_prep_rows": [_c_int] * 2,
    "apn_sa_bwd_prep": [_c_int] * 3 + [_c_void_p] + [_c_longlong] * 3 + [_c_void_p] + [_c_int]
                       + [_c_void_p] * 3 + [_c_int] + [_c_void_p] * 8,
    "apn_sa_bwd_main": [_c_int] * 4 + [_c_float] + [_c_void_p] * 11 + [_c_double, _c_int] + [_c_void_p] * 9,
    "apn_sa_bwd_weight_rows": [_c_int] * 2,
    "apn_sa_bwd_point_grads": [_c_int] * 3 + [_c_void_p] * 7 + [_c_double, _c_int] + [_c_void_p] * 2 + [_c_int]
                              + [_c_void_p] * 4 + [_c_float] + [_c_void_p] * 5,
    "apn_sa_bwd_finalize": [_c_void_p, _c_int, _c_float, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int]
                           + [_c_void_p] * 11,
    "apn_sa_forward_seq": ([_c_int] * 5 + [_c_float] + [_c_void_p] * 12
                           + [_c_void_p] * 5 + [_c_float, _c_float, _c_int]
                           + [_c_void_p] * 5 + [_c_float, _c_float, _c_int]
                           + [_c_double, _c_int] + [_c_void_p] * 11 + [_c_longlong, _c_void_p]),
    "apn_sa_backward_seq": ([_c_int] * 5 + [_c_float] + [_c_void_p] * 15 + [_c_int] * 3
                            + [_c_double] + [_c_void_p] + [_c_longlong] * 3 + [_c_void_p]
                            + [_c_longlong]
                            + [_c_void_p] * 26),
    "apn_attention_small_max": [],
    "apn_attention_small_fwd": [_c_int] * 3 + [_c_void_p] * 5,
    "apn_attention_small_bwd": [_c_int] * 3 + [_c_void_p] * 8,
    "apn_attention_prep": [_c_int] * 3 + [_c_void_p] * 4 + [_c_int, _c_void_p],
    "apn_attention_fwd": [_c_int] * 3 + [_c_void_p] * 4,
    "apn_attention_bwd": [_c_int] * 3 + [_c_void_p] * 9,
    "apn_pointset_group_rows": [_c_int] * 3,
    "apn_pointset_group_max": [_c_int] * 5 + [_c_void_p] * 8,
    "apn_pointset_group_max_grad": [_c_int] * 5 + [_c_void_p] * 9,
    "apn_sa_wide_grid": [_c_int] * 2,
    "apn_sa_wide_colsum_chunks": [_c_int] * 2,
    "apn_sa_wide_colsum": [_c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p],
    "apn_sa_wide_tilemap_ints": [_c_int] * 2,
    "apn_sa_wide_tilemap": [_c_int] * 3 + [_c_void_p] * 3,
    "apn_sa_wide_tilemap_many": [_c_int] * 4 + [_c_void_p] * 3,
    "apn_sa_rowmap_many": [_c_int] * 4 + [_c_void_p] * 6,
    "apn_sa_rowmap_ints": [_c_int] * 2,
    "apn_sa_rowmap_places": [_c_int] * 3,
    "apn_sa_wide_stats1": [_c_int] * 4 + [_c_void_p] * 6,
    "apn_sa_wide_fwd_main": [_c_int] * 5 + [_c_void_p] * 11,
    "apn_sa_wide_bwd_main": [_c_int] * 5 + [_c_void_p] * 15,
    "apn_sa_wide_wgrad_fused": [_c_int],
    "apn_sa_wide_image": [_c_void_p, _c_int, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p],
    "apn_sa_wide_bwd_prep_rows": [_c_int] * 2,
    "apn_sa_wide_bwd_prep": [_c_int] * 3 + [_c_void_p] + [_c_longlong] * 3 + [_c_void_p] * 7,
    "apn_sa_wide_consts2": [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_double, _c_int] + [_c_void_p] * 4,
    "apn_sa_wide_consts1": [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_double, _c_int] + [_c_void_p] * 4,
    "apn_sa_wide_point_terms": [_c_int] * 4 + [_c_void_p] * 5 + [_c_int, _c_float] + [_c_void_p] * 7,
    "apn_sa_wide_wgrad_splits": [_c_int] * 3,
    "apn_sa_wide_csr": [_c_int] * 3 + [_c_void_p] * 9,
    "apn_sa_wide_fwd_prep": [_c_int] * 6 + [_c_float] + [_c_void_p] * 13,
    "apn_sa_wide_fwd_prep_rows": [_c_int] * 3,
    "apn_sa_wide_out": [_c_int] * 3 + [_c_void_p] * 2 + [_c_int] + [_c_void_p] * 3 + [_c_int] + [_c_void_p] * 2,
    "apn_sa_wide_bwd_mid": [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_double, _c_int] + [_c_void_p] * 7,
    "apn_sa_wide_bwd_fin": [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_double, _c_int] + [_c_void_p] * 8,
    "apn_sa_wide_point_grads_rows": [_c_int] * 2,
    "apn_sa_wide_point_grads_cols": [_c_int] * 3,
    "apn_sa_wide_point_grads": [_c_int] * 5 + [_c_float] + [_c_void_p] * 13 + [_c_int] + [_c_void_p] * 9,
    "apn_sa_wide_colsum_f32": [_c_void_p, _c_int, _c_int, _c_void_p, _c_void_p],
    "apn_sa_wide_wgrad": [_c_int] * 5 + [_c_void_p] * 7 + [_c_int] + [_c_void_p] * 2,
    "apn_pw_conv_tiles": [_c_int] * 2,
    "apn_pw_conv_forward": [_c_int] * 5 + [_c_void_p] * 5,
    "apn_pw_bn_act": [_c_int] * 3 + [_c_void_p] * 2 + [_c_int] + [_c_void_p] * 2 + [_c_float] * 2 + [_c_int] * 2
                     + [_c_void_p] * 6,
    "apn_pw_bn_act_grad_splits": [_c_int] * 2,
    "apn_pw_bn_act_grad": [_c_int] * 3 + [_c_void_p] * 3 + [_c_int] * 2 + [_c_void_p] * 5,
    "apn_pw_conv_grad_input": [_c_int] * 5 + [_c_void_p] * 4,
    "apn_pw_conv_grad_weight_splits": [_c_int] * 4,
    "apn_pw_conv_grad_weight": [_c_int] * 5 + [_c_void_p] * 5,
    "apn_deform_forward": [_c_int] * 3 + [_c_void_p] * 6 + [_c_float] + [_c_void_p] * 4,
    "apn_deform_backward": [_c_int] * 3 + [_c_void_p] * 4 + [_c_float] + [_c_void_p] * 7,
    "apn_pw_contract_splits": [_c_int] * 4,
    "apn_pw_contract": [_c_int] * 4 + [_c_void_p, _c_longlong, _c_int, _c_int] * 2 + [_c_void_p, _c_longlong, _c_int, _c_int,
                                                                                         _c_void_p, _c_int, _c_void_p],
    "apn_pw_transpose": [_c_int] * 3 + [_c_void_p] * 3,
    "apn_three_nn_weights": [_c_longlong] + [_c_void_p] * 3,
    "apn_anchor_transforms": [_c_int] + [_c_void_p] * 3 + [_c_float] * 3 + [_c_void_p] * 3,
    "apn_anchor_transforms_grad": [_c_int] + [_c_void_p] * 3 + [_c_float] * 3 + [_c_void_p] * 4,
    "apn_pw_conv_max_tiles": [_c_int],
    "apn_pw_conv_max_forward": [_c_int] * 5 + [_c_void_p] * 3 + [_c_int] + [_c_void_p] * 5,
    "apn_pw_conv_max_backward": [_c_int] * 4 + [_c_void_p] * 5 + [_c_int] + [_c_void_p] * 5,
    "apn_spectral_norm_blocks": [_c_int] * 2,
    "apn_spectral_norm": [_c_int] * 2 + [_c_void_p, _c_int, _c_float] + [_c_void_p] * 8,
    "apn_spectral_norm_grad": [_c_int] * 2 + [_c_void_p] * 8,
    "apn_spectral_norm_many": [_c_int] + [_c_void_p] * 3 + [_c_int, _c_float] + [_c_void_p] * 8,
    "apn_spectral_norm_grad_many": [_c_int] + [_c_void_p] * 10,
    "apn_sa_sample_overlap": [_c_int] * 3 + [_c_float, _c_int] + [_c_void_p] * 7,
    "apn_sa_sample_seq": [_c_int] * 3 + [_c_float, _c_int] + [_c_void_p] * 8,
    "apn_furthest_point_sampling_nested": [_c_int] * 3 + [_c_void_p] * 6,
    "apn_sa_sample_seq_nested": [_c_int] * 3 + [_c_float, _c_int] + [_c_void_p] * 9,
}

_lib = None


class ExtensionMissing(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raise loudly if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ExtensionMissing(
            f"{LIB_PATH} not found: build it with `python -m adaptpoint_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a symbol is missing
        fn.argtypes = argtypes
        fn.restype = _c_int
    lib.apn_error_string.argtypes = [_c_int]
    lib.apn_error_string.restype = ctypes.c_char_p
    lib.apn_build_flags.argtypes = []
    lib.apn_build_flags.restype = ctypes.c_char_p
    # a build without the two correctness flags gives wrong results beside other streams' kernels: refuse it
    flags = lib.apn_build_flags().decode("utf-8", "replace").split()
    missing = [f for f in REQUIRED_BUILD_FLAGS if f not in flags]
    if missing and os.environ.get("APN_ALLOW_UNSAFE_LIB") != "1":
        raise ExtensionMissing(f"{LIB_PATH} was built without {missing} (its flags: {' '.join(flags)}): rebuild with "
                               "`python -m adaptpoint_amd.build --force`")
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().apn_error_string(code).decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed: {msg} (code {code})")
