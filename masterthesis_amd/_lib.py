"""ctypes binding of ``libmt_hip.so`` (C ABI declared in ``include/mt_api.h``).

The library is the only arithmetic backend of this package: there is no CPU or eager
PyTorch fallback.  If the shared object is missing or a symbol is absent the import of
the op layer fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmt_hip.so")
if os.environ.get("MT_LIB_PATH"):      # development only: A/B a differently built library (tools/ab_env.sh, tools/diag_build.sh)
    LIB_PATH = os.path.abspath(os.environ["MT_LIB_PATH"])

MT_F32, MT_BF16 = 0, 1
PAD_ZERO, PAD_REFLECT = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
PACK_FWD, PACK_BWD_DATA = 0, 1
NORM_INSTANCE, NORM_ADAIN, NORM_LAYER, NORM_BATCH = 0, 1, 2, 3
GAN_LSGAN, GAN_HINGE_D, GAN_NEG_MEAN = 1, 2, 3


class ConvDesc(C.Structure):
    """mirror of ``mt_conv_desc``"""
    _fields_ = [
        ("dtype", C.c_int), ("transposed", C.c_int),
        ("N", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("Ci", C.c_int), ("Co", C.c_int),
        ("kh", C.c_int), ("kw", C.c_int),
        ("stride", C.c_int), ("pad", C.c_int), ("pad_mode", C.c_int), ("out_pad", C.c_int),
        ("act", C.c_int), ("slope", C.c_float),
    ]


class BwdStats(C.Structure):
    """mirror of ``mt_bwd_stats``"""
    _fields_ = [("x", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("sums", C.c_void_p),
                ("act", C.c_int), ("slope", C.c_float)]


_p, _i, _f, _z = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_i64, _u64 = C.c_int64, C.c_uint64
_dp = C.POINTER(ConvDesc)

# name -> (restype, argtypes).  Every entry is a symbol declared in include/mt_api.h.
SIGNATURES = {
    "mt_last_error": (C.c_char_p, []),
    "mt_version": (_i, []),
    "mt_kernel_variant_launches": (C.c_long, [_i]),
    "mt_kernel_variant_enable": (_i, [_i, _i]),
    "mt_kernel_variant_epoch": (C.c_long, []),
    "mt_conv_out_hw": (_i, [_dp, C.POINTER(_i), C.POINTER(_i)]),
    "mt_conv_pack_bytes": (_z, [_dp, _i]),
    "mt_conv_pack": (_i, [_dp, _i, _p, _p, _p]),
    "mt_conv_pack_multi_table_bytes": (_z, [_i]),
    "mt_conv_pack_multi_build": (_i, [_i, _p, _p, _p, _p, _p, C.POINTER(_i), C.POINTER(_i)]),
    "mt_conv_pack_multi_run": (_i, [_p, _i, _i, _p]),
    "mt_conv_fwd": (_i, [_dp, _p, _p, _p, _p, _p]),
    "mt_conv_fwd_ws_bytes": (_z, [_dp]),
    "mt_conv_fwd_ex": (_i, [_dp, _p, _p, _p, _p, _p, _z, _p]),
    "mt_conv_fwd_stats": (_i, [_dp, _p, _p, _p, _p, _p, _p]),
    "mt_conv_bwd_data_ws_bytes": (_z, [_dp]),
    "mt_conv_bwd_data": (_i, [_dp, _p, _p, _p, _p, _z, _p]),
    "mt_conv_bwd_data_add": (_i, [_dp, _p, _p, _p, _p, _p, _z, _p]),
    "mt_conv_bwd_data_ex": (_i, [_dp, _p, _p, _p, _p, C.POINTER(BwdStats), C.POINTER(C.c_int), _p, _z, _p]),
    "mt_conv_bwd_weight_ws_bytes": (_z, [_dp]),
    "mt_conv_bwd_weight": (_i, [_dp, _p, _p, _p, _p, _p, _z, _i, _p]),
    "mt_conv_bwd_weight_group_max": (_i, [_dp]),
    "mt_conv_bwd_weight_group_ws_bytes": (_z, [_dp, _i]),
    "mt_conv_bwd_weight_group": (_i, [_dp, _i, _p, _p, _p, _p, _z, _i, _p]),
    "mt_conv_bwd_weight_partial": (_i, [_dp, _p, _p, _p, _p, _z, _i, _i, C.POINTER(C.c_int), _p]),
    "mt_conv_bwd_weight_finish": (_i, [_dp, _p, _i, _p, _i, _p]),
    "mt_conv_bwd_weight_finish_multi": (_i, [_i, _p, _p, _p, _p, _i, _p]),
    "mt_conv_bwd_weight_slab_bytes": (_z, [_dp]),
    "mt_conv_bwd_weight_rows_ok": (_i, [_dp]),
    "mt_conv_bwd_weight_rows_multi_ws_bytes": (_z, [_i, _p]),
    "mt_conv_bwd_weight_rows_multi": (_i, [_i, _p, _p, _p, _p, _p, _z, _i, _p]),
    "mt_linear_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "mt_linear_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "mt_conv_fwd_stats_fused": (_i, [_dp]),
    "mt_nc_stats_parts": (_i, [_i, _i, _i, _i]),
    "mt_norm_apply_fused": (_i, [_i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _p]),
    "mt_act_bwd_bias_ws_bytes": (_z, [_i]),
    "mt_act_bwd_bias": (_i, [_i, _p, _p, _p, _z, _i, _i, _i, _f, _p, _i, _p, _z, _p]),
    "mt_nc_stats": (_i, [_i, _p, _p, _i, _i, _i, _p]),
    "mt_norm_finalize": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p]),
    "mt_scale_shift_act": (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "mt_nc_stats_bwd": (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "mt_norm_bwd_finalize": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "mt_norm_bwd_apply": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "mt_norm_bwd_onepass_capacity": (_i, []),
    "mt_norm_bwd_onepass_ok": (_i, [_i, _i, _i, _i, _i, _i, _i, _p]),
    "mt_norm_bwd_onepass": (_i, [_i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p]),
    "mt_act_fwd": (_i, [_i, _p, _p, _z, _i, _f, _p]),
    "mt_act_bwd": (_i, [_i, _p, _p, _p, _z, _i, _f, _p]),
    "mt_add": (_i, [_i, _p, _p, _p, _z, _p]),
    "mt_gaussian_noise_add": (_i, [_i, _p, _p, _z, _u64, _u64, _p]),
    "mt_bernoulli_mask": (_i, [_i, _p, _z, _i, _i, _f, _u64, _u64, _p]),
    "mt_gaussian_noise_add_dev": (_i, [_i, _p, _p, _z, _p, _p]),
    "mt_rng_advance": (_i, [_p, _p]),
    "mt_bernoulli_mask_dev": (_i, [_i, _p, _z, _i, _i, _f, _p, _p]),
    "mt_mul_scale": (_i, [_i, _p, _p, _p, _z, _f, _p]),
    "mt_avgpool2_fwd": (_i, [_i, _p, _p, _i, _i, _i, _i, _p]),
    "mt_avgpool2_bwd": (_i, [_i, _p, _p, _i, _i, _i, _i, _p]),
    "mt_upsample2_fwd": (_i, [_i, _p, _p, _i, _i, _i, _i, _p]),
    "mt_upsample2_bwd": (_i, [_i, _p, _p, _i, _i, _i, _i, _p]),
    "mt_bn_finalize": (_i, [_p, _p, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "mt_bn_bwd_finalize": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mt_sn_ws_bytes": (_z, [_i, _i]),
    "mt_sn_power_iter": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _p, _z, _p]),
    "mt_sn_scale_fwd": (_i, [_p, _p, _p, C.c_long, _p]),
    "mt_sn_scale_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _p, _z, _p]),
    "mt_avgpool3s2_fwd": (_i, [_i, _p, _p, _i, _i, _i, _i, _p]),
    "mt_avgpool3s2_bwd": (_i, [_i, _p, _p, _i, _i, _i, _i, _p]),
    "mt_patch4s2_fwd": (_i, [_i, _p, _p, _i, _i, _i, _i, _p]),
    "mt_patch4s2_bwd": (_i, [_i, _p, _p, _i, _i, _i, _i, _p]),
    "mt_patch4s2_multi": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _i, _p]),
    "mt_gap_fwd": (_i, [_i, _p, _p, _i, _i, _i, _i, _p]),
    "mt_gap_bwd": (_i, [_i, _p, _p, _i, _i, _i, _i, _p]),
    "mt_to_nhwc": (_i, [_i, _p, _i64, _i64, _i64, _i64, _i, _p, _i, _i, _i, _i, _p]),
    "mt_to_nchw_f32": (_i, [_i, _p, _p, _i, _i, _i, _i, _p]),
    "mt_cat_class_planes": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _p]),
    "mt_slice_channels": (_i, [_i, _p, _p, _i, _i, _i, _i, _p]),
    "mt_bce_const_fwd": (_i, [_i, _p, _f, _p, _z, _i, _i, _p]),
    "mt_bce_const_bwd": (_i, [_i, _p, _f, _p, _p, _z, _i, _i, _p]),
    "mt_gan_const_fwd": (_i, [_i, _i, _p, _f, _p, _z, _i, _i, _p]),
    "mt_gan_const_bwd": (_i, [_i, _i, _p, _f, _p, _p, _z, _i, _i, _p]),
    "mt_bce_target_fwd": (_i, [_p, _p, _p, _z, _p]),
    "mt_bce_target_bwd": (_i, [_p, _p, _p, _p, _z, _p]),
    "mt_l1_fwd": (_i, [_i, _p, _p, _p, _z, _z, _p]),
    "mt_l1_bwd": (_i, [_i, _p, _p, _p, _p, _p, _z, _z, _p]),
    "mt_l2mean_fwd": (_i, [_i, _p, _p, _z, _z, _p]),
    "mt_l2mean_bwd": (_i, [_i, _p, _p, _p, _z, _z, _p]),
    "mt_reparam_fwd": (_i, [_p, _p, _p, _p, _z, _p]),
    "mt_reparam_bwd": (_i, [_p, _p, _p, _p, _p, _z, _p]),
    "mt_kl_fwd": (_i, [_p, _p, _p, _z, _p]),
    "mt_kl_bwd": (_i, [_p, _p, _p, _p, _p, _z, _p]),
    "mt_adam_multi": (_i, [_p, _p, _i, _i64, _f, _f, _f, _f, _f, _i, _p]),
    "mt_adam_multi_dev": (_i, [_p, _p, _i, _i64, _f, _f, _f, _f, _p, _i, _p]),
    "mt_loss_sum_fwd": (_i, [_p, _p, _p, _i, _p, _p, _i, _p, _p]),
    "mt_loss_sum_bwd": (_i, [_p, _p, _p, _i, _p, _i, _p, _p]),
    "mt_linear_group_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "mt_linear_group_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "mt_comm_unique_id": (_i, [_p]),
    "mt_comm_init": (_i, [C.POINTER(_p), _i, _i, _p, _i]),
    "mt_comm_allreduce_async": (_i, [_p, _p, _z, _p]),
    "mt_comm_wait": (_i, [_p, _i, _p]),
    "mt_comm_rank": (_i, [_p]),
    "mt_comm_world": (_i, [_p]),
    "mt_comm_destroy": (_i, [_p]),
}

_lib = None


def load():
    """Load the shared library once; raise if it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C masterthesis_amd/csrc`). There is no fallback backend.")
    # torch first: its bundled HIP runtime must be the one already in the process when the library's
    # libamdhip64.so.7 dependency is resolved -- loading the library before torch pulls in /opt/rocm's copy as a
    # second runtime, which then finds no device once torch's has claimed it
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().mt_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
