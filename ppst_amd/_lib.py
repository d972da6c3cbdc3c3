"""ctypes binding of libppst_hip.so (the C ABI declared in include/ppst_hip.h).

The product path has NO fallback: if the library is missing or a symbol cannot
be resolved, importing this module raises, and every op that reaches a kernel
raises on non-CUDA tensors (like the reference's CHECK_CUDA,
stylegan2_op/upfirdn2d.cpp:9, fused_bias_act.cpp:8).
"""
import ctypes
import os

# PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64; it must be in the
# process BEFORE libppst_hip.so is dlopen'ed so that both resolve to ONE HIP runtime (the
# dynamic loader matches the SONAME).  Loading ours first pulls /opt/rocm's runtime in
# and torch's allocations then live in a different runtime: launches fail with
# hipErrorNoDevice.  Hosts without torch (cgo/JNI) use the system runtime.
import torch  # noqa: F401  (import order matters)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PPST_HIP_LIB") or os.path.join(HERE, "libppst_hip.so")  # env override: diagnostic builds only

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "ppst_amd: %s is missing -- build it with `python -m ppst_amd.build` "
        "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)

vp, i32, i64, f32, f64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double


class PackJob(ctypes.Structure):
    """ppst_pack_job (include/ppst_hip.h)."""
    _fields_ = [("w", vp), ("sn", i64), ("sc", i64), ("sy", i64), ("sx", i64), ("src_c", vp), ("src_ky", vp), ("src_kx", vp),
                ("out", vp), ("total", i64), ("block0", i64), ("scale", f32), ("cout", i32), ("bn", i32), ("nsteps", i32),
                ("n_groups", i32), ("x3", i32), ("f16", i32), ("nblocks", i32), ("dual", i32)]


class UpscaleJob(ctypes.Structure):
    """ppst_upscale_job (include/ppst_hip.h)."""
    _fields_ = [("w", vp), ("out", vp), ("total", i64), ("block0", i64), ("scale", f32), ("cout", i32), ("cin", i32), ("nblocks", i32)]


class ConvArgs(ctypes.Structure):
    """ppst_conv_args (include/ppst_hip.h)."""
    _fields_ = [
        ("x", vp), ("wpack", vp), ("steps", vp), ("y", vp), ("bias", vp), ("noise", vp),
        ("prelu", vp), ("stats", vp), ("residual", vp),
        ("noise_weight", f32), ("out_scale", f32),
        ("B", i32), ("in_h", i32), ("in_w", i32), ("in_ld", i32),
        ("out_h", i32), ("out_w", i32), ("out_ld", i32), ("cout", i32),
        ("nsteps", i32), ("n_groups", i32), ("pad_mode", i32),
        ("in_off_y", i32), ("in_off_x", i32), ("out_sy", i32), ("out_sx", i32),
        ("act", i32), ("precision", i32), ("res_ld", i32), ("tile_h", i32), ("tile_w", i32),
        ("halo", i32), ("bn", i32), ("in_scale_shift", vp), ("in_prelu", vp), ("in_c", i32), ("in_act", i32),
        ("flop_steps", i32), ("tile_rows", i32), ("a_slots", i32), ("early_a", i32), ("variant", i32), ("in_presplit", i32), ("dual_b", i32), ("io_st", i32), ("k64", i32),
        ("in_res", vp), ("in_res_ld", i32), ("ksplit", i32), ("ksplit_starts", vp),
    ]


_SIGS = {
    "ppst_version": (i32, []),
    "ppst_guided_filter_tune": (i32, [i32, i32]),
    "ppst_upfirdn2d": (i32, [vp, vp, vp] + [i32] * 14 + [i32, vp]),
    "ppst_blur_nhwc": (i32, [vp, vp, vp] + [i32] * 10 + [vp, i32, vp]),
    "ppst_blur_nhwc_st": (i32, [vp, vp, vp] + [i32] * 10 + [vp, i32, i32, vp]),
    "ppst_fused_bias_act": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, f32, i32, vp]),
    "ppst_nchw_to_nhwc": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ppst_nhwc_to_nchw": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ppst_conv_pack": (i32, [vp, i64, i64, i64, i64, f32, i32, i32, vp, vp, vp, i32, i32, i32, vp, vp]),
    "ppst_conv_pack_k64": (i32, [vp, i64, i64, i64, i64, f32, i32, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "ppst_conv_pack_up9_bytes": (i64, [i32, i32]),
    "ppst_conv_pack_up9": (i32, [vp, i64, i64, i64, i64, f32, i32, i32, vp, vp]),
    "ppst_conv_pack_wino_bytes": (i64, [i32, i32]),
    "ppst_conv_pack_wino": (i32, [vp, i64, i64, i64, i64, f32, i32, i32, vp, vp]),
    "ppst_conv_pack_dual": (i32, [vp, i64, i64, i64, i64, f32, i32, vp, vp, vp, i32, i32, i32, vp, vp]),
    "ppst_upscale_weight": (i32, [vp, vp, i32, i32, f32, vp]),
    "ppst_dgrad_s2d_stack_weight": (i32, [vp, vp, i32, i32, vp]),
    "ppst_depth_to_space_st": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ppst_pack_job_blocks": (i32, [i64]),
    "ppst_conv_pack_batch": (i32, [vp, i32, i32, vp]),
    "ppst_upscale_weight_batch": (i32, [vp, i32, i32, vp]),
    "ppst_conv2d_mfma": (i32, [ctypes.POINTER(ConvArgs), vp]),
    "ppst_conv_ksplit_check": (i32, [vp]),
    "ppst_has_experiments": (i32, []),
    "ppst_presplit": (i32, [vp, vp, i64, i32, i32, i32, vp]),
    "ppst_conv2d_f32": (i32, [ctypes.POINTER(ConvArgs), vp, i64, i64, i64, i64, f32, vp, vp, vp, vp]),
    "ppst_conv_tiles": (i32, [i32, i32, i32]),
    "ppst_conv1x1_small_cin": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, f32, i32, vp]),
    "ppst_conv1x1_small_cin_st": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, f32, i32, i32, vp]),
    "ppst_conv1x1_small_cout": (i32, [vp, vp, vp, vp, i64, i32, i32, f32, vp]),
    "ppst_conv1x1_small_cout_st": (i32, [vp, vp, vp, vp, i64, i32, i32, f32, i32, vp]),
    "ppst_torgb_apply_st": (i32, [vp, vp, vp, i32, f32, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp]),
    "ppst_in_stats": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, ctypes.POINTER(i32), vp]),
    "ppst_in_finalize": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, f64, f32, vp]),
    "ppst_affine_act": (i32, [vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, i32, i32, vp, f32, i32, vp]),
    "ppst_affine_act_st": (i32, [vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, i32, i32, vp, f32, i32, i32, i32, vp]),
    "ppst_affine_act_stats": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, f32, i32, i32, vp]),
    "ppst_upsample_nearest2": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ppst_upsample_nearest2_st": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "ppst_gap_gmp_ws": (i64, [i32, i64, i32]),
    "ppst_gap_gmp": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ppst_gap_gmp_st": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ppst_avgpool": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ppst_bilinear": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ppst_head_tail": (i32, [vp, vp, vp, vp, vp] + [i32] * 10 + [vp]),
    "ppst_maxpool2": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ppst_linear": (i32, [vp, vp, vp, vp, i32, i32, i32, f32, f32, i32, i32, vp]),
    "ppst_l2norm_rows": (i32, [vp, vp, i32, i32, f32, i32, vp]),
    "ppst_lerp": (i32, [vp, vp, vp, i64, f32, vp]),
    "ppst_spatial_modulation": (i32, [vp, vp, vp, vp, i32, i64, i32, vp]),
    "ppst_spatial_modulation_st": (i32, [vp, vp, vp, vp, i32, i64, i32, i32, vp]),
    "ppst_rselfcorr": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "ppst_corr_prep": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ppst_gemm_nt_f32": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, vp]),
    "ppst_gemm_nn_f32": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ppst_unfold_rows": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "ppst_unfold_rows_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "ppst_gemm_nt_split": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, i32, vp]),
    "ppst_gemm_nn_split": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ppst_softmax_rows": (i32, [vp, i64, i32, f32, vp]),
    "ppst_unfold_patches": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "ppst_fold_patches": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "ppst_tensor2im_u8": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ppst_resample_u8": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp]),
    "ppst_u8_to_tensor": (i32, [vp, vp, i32, i32, i32, i32, f32, f32, vp]),
    "ppst_guided_filter_ws": (i64, [i32, i32, i32]),
    "ppst_guided_filter": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp]),
    "ppst_smooth_local_affine_ws": (i64, [i32, i32, i32]),
    "ppst_smooth_local_affine": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, f32, vp]),
    "ppst_conv_wgrad_f32": (i32, [vp, vp, vp, vp, vp] + [i32] * 11 + [vp]),
    "ppst_conv_wgrad_bf16x3": (i32, [vp, vp, vp, vp, vp] + [i32] * 11 + [vp]),
    "ppst_conv_wgrad_tr": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ppst_conv_wgrad_tr2": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ppst_wgrad_scatter": (i32, [vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, i32, i32, f32, i32, vp, vp, i32, i32, vp]),
    "ppst_wgrad_small_cin_ws": (i64, [i64, i32, i32]),
    "ppst_wgrad_small_cin": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, f32, i32, vp]),
    "ppst_colsum_ws": (i64, [i64, i32]),
    "ppst_colsum": (i32, [vp, vp, vp, i64, i32, i32, f32, i32, vp]),
    "ppst_linear_wgrad": (i32, [vp, vp, vp, i32, i32, i32, f32, i32, vp]),
    "ppst_linear_dgrad_ws": (i64, [i32, i32, i32]),
    "ppst_linear_dgrad": (i32, [vp, vp, vp, vp, i32, i32, i32, f32, vp]),
    "ppst_linear_wgrad_fused": (i32, [vp, vp, vp, vp, i32, i32, i32, f32, f32, i32, i32, i32, vp]),
    "ppst_linear_dgrad_gate": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, f32, vp]),
    "ppst_lsgan": (i32, [vp, vp, vp, i32, f32, f32, vp]),
    "ppst_l1_mean_ws": (i64, [i64]),
    "ppst_l1_mean": (i32, [vp, vp, vp, vp, i64, f32, vp]),
    "ppst_rscl_loss": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]),
    "ppst_adam_step": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, i32, vp]),
    "ppst_in_finalize_train": (i32, [vp, i32, vp, i32, vp, vp, vp, i32, i32, f64, f32, vp]),
    "ppst_dual_stats": (i32, [vp, vp, vp, vp, i32, i64, i32, i32, i32, i32, ctypes.POINTER(i32), vp]),
    "ppst_in_bwd_finalize": (i32, [vp, i32, vp, vp, i32, vp, vp, i32, i32, f64, vp]),
    "ppst_in_bwd_apply": (i32, [vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, i32, i32, i32, vp]),
    "ppst_prelu_bwd_ws": (i64, [i64]),
    "ppst_prelu_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, i32, vp]),
    "ppst_sum_partials": (i32, [vp, vp, i32, f32, vp]),
    "ppst_pad2d": (i32, [vp, vp] + [i32] * 10 + [vp]),
    "ppst_pad2d_bwd": (i32, [vp, vp] + [i32] * 9 + [vp]),
    "ppst_bilinear_bwd": (i32, [vp, vp] + [i32] * 8 + [vp]),
    "ppst_avgpool_bwd": (i32, [vp, vp] + [i32] * 7 + [vp]),
    "ppst_gap_gmp_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, vp]),
    "ppst_gap_gmp_multi_ws": (i64, [i32, i64, i32, i32]),
    "ppst_gap_gmp_multi": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ppst_gap_gmp_multi_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, i32, i32, i32, vp]),
    "ppst_gap_gmp_bwd_st": (i32, [vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, i32, vp]),
    "ppst_dual_stats_st": (i32, [vp, vp, vp, vp, i32, i64, i32, i32, i32, i32, ctypes.POINTER(i32), i32, vp]),
    "ppst_in_bwd_apply_st": (i32, [vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ppst_pad2d_st": (i32, [vp, vp] + [i32] * 11 + [vp]),
    "ppst_pad2d_bwd_st": (i32, [vp, vp] + [i32] * 10 + [vp]),
    "ppst_bilinear_bwd_st": (i32, [vp, vp] + [i32] * 9 + [vp]),
    "ppst_noise_wgrad_st": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]),
    "ppst_space_to_depth_st": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ppst_colsum_st": (i32, [vp, vp, vp, i64, i32, i32, f32, i32, i32, vp]),
    "ppst_wgrad_small_cin_st": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, f32, i32, i32, vp]),
    "ppst_conv_wgrad_tr2_st": (i32, [vp, vp, vp, vp, vp, vp] + [i32] * 16 + [vp]),
    "ppst_l2norm_rows_bwd": (i32, [vp, vp, vp, i32, i32, f32, i32, vp]),
    "ppst_softmax_rows_bwd": (i32, [vp, vp, i64, i32, f32, vp]),
    "ppst_corr_prep_bwd": (i32, [vp, vp, vp, i64, i32, i32, f32, vp]),
    "ppst_l1_grad": (i32, [vp, vp, vp, i64, f32, vp]),
    "ppst_scale_by": (i32, [vp, vp, vp, i64, vp]),
    "ppst_rscl_loss_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]),
    "ppst_rselfcorr_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ppst_noise_wgrad_ws": (i64, [i64]),
    "ppst_noise_wgrad": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, vp]),
    "ppst_upscale_weight_bwd": (i32, [vp, vp, i32, i32, f32, i32, vp]),
    "ppst_space_to_depth": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "ppst_prof_enable": (i32, [i32]),
    "ppst_prof_dropped": (i32, []),
    "ppst_wgrad_flop_steps": (i32, [i32]),
    "ppst_prof_collect": (i32, [ctypes.POINTER(f64), ctypes.POINTER(i64), ctypes.POINTER(f64)]),
    "ppst_prof_detail": (i32, [i32, ctypes.POINTER(f64), ctypes.POINTER(f64), ctypes.POINTER(i32)]),
}

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here == ABI mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args

ABI_VERSION = 2          # PPST_ABI_VERSION of include/ppst_hip.h: the struct layouts above are that revision's
if lib.ppst_version() != ABI_VERSION:
    raise ImportError("ppst_amd: %s reports ABI revision %d, this binding is written for %d -- rebuild the library "
                      "(python -m ppst_amd.build)" % (LIB_PATH, lib.ppst_version(), ABI_VERSION))

_ERR = {-1: "PPST_EINVAL (bad size / flag combination)", -2: "PPST_EUNSUPPORTED", -3: "PPST_ENULL"}


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, _ERR.get(rc, "hipError_t %d" % rc)))


def exported_symbols():
    return sorted(_SIGS)
