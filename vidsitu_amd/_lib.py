"""ctypes binding of libvidsitu_hip.so (the C-ABI in include/vidsitu_hip.h).

The product path has NO CPU fallback: if the shared object is missing or a
symbol is absent this module raises at first use, loudly.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# VS_LIB_PATH: A/B runs of two builds of the same ABI in one GPU session (tools/); never a fallback.
LIB_PATH = os.environ.get("VS_LIB_PATH") or os.path.join(_HERE, "libvidsitu_hip.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

VS_CONV_AFFINE = 1
VS_CONV_RESIDUAL = 2
VS_CONV_RELU = 4
VS_CONV_STATS = 8
VS_CONV_NAIVE = 16


class ConvDesc(C.Structure):
    """vs_conv_desc (include/vidsitu_hip.h)."""

    _fields_ = [
        (n, C.c_int32)
        for n in (
            "N Ti Hi Wi Cin To Ho Wo Cout kT kH kW sT sH sW pT pH pW x_ld y_ld res_ld flags"
        ).split()
    ]


class DgradEpilogue(C.Structure):
    """vs_dgrad_epilogue (include/vidsitu_hip.h)."""

    _fields_ = [("residual", C.c_void_p), ("residual_bits", C.c_void_p), ("bn_y", C.c_void_p),
                ("bn_y_ld", C.c_int32), ("relu_bits", C.c_void_p), ("mean", C.c_void_p),
                ("invstd", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("stats_partial", C.c_void_p), ("bn_y2", C.c_void_p), ("bn_y2_ld", C.c_int32),
                ("mean2", C.c_void_p), ("invstd2", C.c_void_p), ("stats_partial2", C.c_void_p)]


class WgradItem(C.Structure):
    """vs_wgrad_item (include/vidsitu_hip.h)."""

    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("d", ConvDesc)]


_p, _i, _i64, _f, _d, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t
_dp = C.POINTER(ConvDesc)

# name -> (restype, argtypes); mirrors include/vidsitu_hip.h one to one
SIGNATURES = {
    "vs_last_error_string": (C.c_char_p, []),
    "vs_version": (_i, []),
    "vs_launch_count": (_i64, []),
    "vs_pack_input": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "vs_frames_u8_pack": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p]),
    "vs_stem_conv_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "vs_stem_stats_rows": (_i, [_i, _i, _i, _i]),
    "vs_stem_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "vs_stem_conv_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "vs_conv_fwd": (_i, [_p, _p, _p, _dp, _p, _p, _p, _p, _p, _sz, _p]),
    "vs_conv_stats_rows": (_i, [_dp]),
    "vs_conv_workspace_bytes": (_sz, [_dp, _i]),
    "vs_conv_plan": (_i, [_dp, _i, _p]),
    "vs_conv_aol_ok": (_i, [_dp]),
    "vs_conv_fwd_aol": (_i, [_p, _p, _p, _dp, _p, _p, _p, _p]),
    "vs_conv_wgrad_aol_ok": (_i, [_dp]),
    "vs_conv_wgrad_aol": (_i, [_p, _p, _p, _dp, _p, _p, _p, _sz, _p]),
    "vs_conv_fwd_bc_fusable": (_i, [_dp, _i]),
    "vs_conv_fwd_bc": (_i, [_p, _p, _dp, _p, _p, _p, _i, _p, _p, _p, _i, _p, _i, _i, _p]),
    "vs_conv_dgrad": (_i, [_p, _p, _p, _dp, _p, _p, _sz, _p]),
    "vs_conv_dgrad_bnstats_rows": (_i, [_dp]),
    "vs_conv_dgrad_bnstats": (_i, [_p, _p, _p, _dp, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "vs_conv_dgrad_ex": (_i, [_p, _p, _p, _dp, C.POINTER(DgradEpilogue), _p, _sz, _p]),
    "vs_weight_transpose": (_i, [_p, _p, _i, _i, _i, _p]),
    "vs_weight_transpose_batched": (_i, [_p, _p, _p, _i, _i64, _p]),
    "vs_transpose_f32_batched": (_i, [_p, _p, _p, _i, _i64, _p]),
    "vs_weight_transpose_tiled": (_i, [_p, _p, _p, _p, _i, _i64, _i, _p]),
    "vs_conv_wgrad_workspace_bytes": (_sz, [_dp]),
    "vs_conv_wgrad": (_i, [_p, _p, _p, _dp, _p, _sz, _p]),
    "vs_conv_wgrad_group_ok": (_i, [C.POINTER(WgradItem), _i]),
    "vs_conv_wgrad_group_workspace_bytes": (_sz, [C.POINTER(WgradItem), _i]),
    "vs_conv_wgrad_group": (_i, [C.POINTER(WgradItem), _i, _p, _sz, _p]),
    "vs_conv_wgrad_partial": (_i, [_p, _p, _p, _dp, _p, _sz, C.POINTER(C.c_int), _p]),
    "vs_wgrad_reduce_blocks": (_i64, [_i64]),
    "vs_wgrad_reduce": (_i, [_p, _p, _i64, _i, _p]),
    "vs_wgrad_reduce_batched": (_i, [_p, _i, _i64, _p]),
    "vs_wgrad_reduce_defer": (_i, [_i]),
    "vs_wgrad_reduce_flush": (_i, []),
    "vs_bn_finalize": (_i, [_p, _i, _d, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _i, _p]),
    "vs_bn_finalize_workspace_bytes": (_sz, []),
    "vs_bn_finalize_ws": (_i, [_p, _i, _d, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _i, _p, _sz, _p]),
    "vs_bn_bwd_finalize_ws": (_i, [_p, _i, _p, _p, _i, _p, _sz, _p]),
    "vs_bn_partials_reduce": (_i, [_p, _i, _p, _i, _i, _p]),
    "vs_bn_apply": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _p]),
    "vs_bn_apply_mask": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _p]),
    "vs_bn_apply2": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _p]),
    "vs_bn_bwd_reduce": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _p]),
    "vs_bn_bwd_reduce_rows": (_i, [_i64, _i]),
    "vs_bn_bwd_finalize": (_i, [_p, _i, _p, _p, _i, _p]),
    "vs_bn_bwd_apply": (
        _i,
        [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _p],
    ),
    "vs_bn_bwd_apply2": (_i, [_p] * 16 + [_i64, _i, _i, _i, _i, _i, _i, _p]),
    "vs_residual_add_f32": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _i, _p]),
    "vs_maxpool_hw3s2_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "vs_maxpool_hw3s2_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "vs_bn_apply_maxpool": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "vs_bn_bwd_reduce_pool": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "vs_bn_bwd_apply_pool": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "vs_maxpool_t_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vs_maxpool_t_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vs_avgpool_fwd": (_i, [_p, _p, _i, _i64, _i, _i, _i, _i, _p]),
    "vs_avgpool_bwd": (_i, [_p, _p, _i, _i64, _i, _i, _i, _i, _p]),
    "vs_linear_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vs_transpose_f32": (_i, [_p, _p, _i, _i, _p]),
    "vs_linear_bwd_data": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "vs_linear_bwd_weight": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "vs_linear_bwd_fused": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "vs_conv_pair_begin": (_i, []),
    "vs_conv_pair_end": (_i, []),
    "vs_conv_pair_count": (_i64, []),
    "vs_linear_bwd_fused_res": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "vs_attn_small_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "vs_attn_small_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "vs_add_layernorm_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "vs_add_layernorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "vs_softmax_xent": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "vs_softmax_topk": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "vs_adam_step": (_i, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _i, _f, _p]),
    "vs_adam_step_dev": (_i, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _p, _f, _p]),
    "vs_adam_step_dev_cast": (_i, [_p, _p, _p, _p, _p, _i64, _f, _f, _f, _f, _p, _f, _p]),
    "vs_adam_step_dev_cast_g16": (_i, [_p, _p, _p, _p, _p, _i64, _f, _f, _f, _f, _p, _f, _p]),
    "vs_adam_tick": (_i, [_p, _p]),
    "vs_adam_step_dev_range": (_i, [_p, _p, _i, _p, _p, _p, _i64, _f, _f, _f, _f, _p, _f, _p]),
    "vs_cast_f32_to_bf16": (_i, [_p, _p, _i64, _p]),
    "vs_gemm_nt_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vs_gpt2_embed": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vs_attn_causal_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "vs_attn_decode": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "vs_pack_rows_f32": (_i, [_p, _p, _i, _i, _p]),
    "vs_gemm_nt_f32_workspace_bytes": (_sz, [_i, _i, _i]),
    "vs_gemm_nt_f32_ws": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "vs_maxpool_hw2_fwd": (_i, [_p, _p, _p, _i64, _i, _i, _i, _p]),
    "vs_maxpool_hw2_bwd": (_i, [_p, _p, _p, _i64, _i, _i, _i, _p]),
    "vs_softmax_rows_bf16": (_i, [_p, _p, _i64, _i, _p]),
    "vs_softmax_rows_bwd_bf16": (_i, [_p, _p, _p, _i64, _i, _f, _p]),
    "vs_colsum_bf16": (_i, [_p, _p, _i64, _i, _i, _p]),
    "vs_embed_pos_fwd": (_i, [_p, _p, _p, _p, _p, _i64, _i, _f, _p]),
    "vs_embed_scatter_bwd": (_i, [_p, _p, _p, _i64, _i, _f, _i64, _p]),
    "vs_relu_bwd": (_i, [_p, _p, _p, _i64, _p]),
    "vs_resize_ksize": (_i, [_i, _i]),
    "vs_resize_coeffs": (_i, [_i, _i, _p, _p]),
    "vs_resize_tmp_bytes": (_sz, [_i64, _i, _i]),
    "vs_resize_bicubic_u8": (_i, [_p, _p, _p, _i64, _i, _i, _i, _i, _p, _p, _i, _p, _p, _i, _i, _i, _p]),
    "vs_gemm_nt_f32_packed": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vs_layernorm_fwd_packed": (_i, [_p, _p, _p, _p, _i, _i, _f, _p]),
    "vs_beam_topk_workspace_bytes": (_sz, [_i, _i, _i]),
    "vs_kv_gather": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vs_beam_topk": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _i, _p, _sz, _p]),
    "vs_xent_ignore": (_i, [_p, _p, _p, _p, _i, _i, _i64, _i, _p]),
    "vs_beam_step": (_i, [_p] * 17 + [_i] * 9 + [_f, _p]),
    "vs_gelu_new_fwd": (_i, [_p, _p, _i64, _p]),
    "vs_gelu_new_bwd": (_i, [_p, _p, _p, _i64, _p]),
    "vs_add_f32": (_i, [_p, _p, _p, _i64, _p]),
    "vs_colsum_f32": (_i, [_p, _p, _i, _i, _p]),
    "vs_attn_causal_bwd_scratch_bytes": (_sz, [_i, _i, _i]),
    "vs_attn_causal_bwd": (_i, [_p, _p, _p, _p, _p, _sz, _i, _i, _i, _i, _p]),
    "vs_gpt2_embed_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vs_xent_ignore_grad": (_i, [_p, _p, _p, _p, _i, _i, _i64, _i, _f, _p]),
    "vs_xent_ignore_grad_dev": (_i, [_p, _p, _p, _p, _i, _i, _i64, _i, _p, _p]),
}


class VsError(RuntimeError):
    pass


_lib = None


def build_library(force=False):
    """Compile csrc/*.hip for gfx950 into libvidsitu_hip.so (hipcc cross-compiles
    without a GPU)."""
    args = ["make", "-C", CSRC_DIR, "-j4"]
    if force:
        subprocess.check_call(["make", "-C", CSRC_DIR, "clean"])
    subprocess.check_call(args)
    return LIB_PATH


def load():
    """dlopen the library and bind every symbol the header declares."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64; it must be in the process first so that this
    # library binds to the same HIP runtime (two runtimes => "no ROCm-capable device").
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise VsError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the HIP path)"
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().vs_last_error_string()
        raise VsError(f"{what} failed with vs_status {rc}: {msg.decode() if msg else ''}")


# bench.py installs a callable here to bracket every entry point with HIP events on the launch
# stream (profiling only; None = straight call)
_probe = None


def call(name, *args):
    lib = load()
    if _probe is not None:
        rc = _probe(name, args, getattr(lib, name))
    else:
        rc = getattr(lib, name)(*args)
    check(rc, name)
