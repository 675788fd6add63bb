"""ctypes binding of libhavatar_hip.so (the C ABI declared in include/havatar.h).

Fails loudly: there is no CPU or PyTorch fallback behind these entry points.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libhavatar_hip.so")

HAV_F32, HAV_F16, HAV_BF16, HAV_F64 = 0, 1, 2, 3
HAV_MLP_SPLIT_BF16, HAV_MLP_F32, HAV_MLP_SPLIT_F16, HAV_MLP_SPLIT_F16_MX = 0, 1, 2, 3
ABI_VERSION = 6
HAV_FLAG_PAIR_KERNEL, HAV_FLAG_FINE_CACHE, HAV_FLAG_FINE_RECOMPUTE, HAV_FLAG_NO_FP16_GUARD = 1, 2, 4, 8
HAV_STATUS_FP16_FALLBACK = 1


class HavRenderParams(C.Structure):
    _fields_ = [("B", C.c_int32), ("R", C.c_int32), ("ray_stride", C.c_int32), ("S_c", C.c_int32),
                ("S_f", C.c_int32), ("perturb", C.c_int32), ("noise_std", C.c_float),
                ("plane_res", C.c_int32), ("plane_ch", C.c_int32), ("vol_res", C.c_int32),
                ("nerf_scale", C.c_float * 3), ("nerf_trans", C.c_float * 3),
                ("skin_scale", C.c_float * 3), ("skin_trans", C.c_float * 3),
                ("seed", C.c_uint64), ("rng_offset", C.c_uint64), ("mlp_mode", C.c_int32), ("flags", C.c_int32),
                ("rng_counter", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_uint64),
                ("dbg_zfine", C.c_void_p), ("status", C.c_void_p), ("grid_blocks", C.c_int32), ("reserved0", C.c_int32)]


class HavFieldParams(C.Structure):
    _fields_ = [("n", C.c_int64), ("n_per_b", C.c_int64), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("D", C.c_int32), ("nerf_scale", C.c_float * 3), ("nerf_trans", C.c_float * 3),
                ("skin_scale", C.c_float * 3), ("skin_trans", C.c_float * 3)]


class HavMlpWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")]


class HavMlpGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")]


class HavRenderOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("rgb_coarse", "depth_coarse", "acc_coarse", "weights_max",
                                          "rgb_fine", "depth_fine", "acc_fine")]


class HavatarLibraryError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the library (once). Raises HavatarLibraryError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("HAVATAR_LIB", LIB_PATH)      # kernel experiments: an alternative build of the SAME library
    if not os.path.exists(path):
        raise HavatarLibraryError(
            f"{path} is missing: build it with `python -m havatar_amd.build` "
            "(or __graft_entry__.build()). There is no fallback path.")
    # PyTorch-ROCm ships its own libamdhip64; the library must bind to THAT runtime (device memory and streams come from torch).
    # Loaded before torch, it would pull in /opt/rocm's copy first and the process would hold two HIP runtimes: every call from
    # here then fails with hipErrorNoDevice (100) -- seen when __graft_entry__.build() ran before the first `import torch`.
    import torch  # noqa: F401
    L = C.CDLL(path)
    L.hav_abi_version.restype = C.c_int
    if L.hav_abi_version() != ABI_VERSION:
        raise HavatarLibraryError(f"ABI mismatch: library {L.hav_abi_version()} vs binding {ABI_VERSION}")
    i64, i32, f32, vp = C.c_int64, C.c_int, C.c_float, C.c_void_p
    L.hav_fused_bias_act.argtypes = [vp, vp, vp, vp, i32, i32, i32, f32, f32, i64, i64, i64, vp]
    L.hav_fused_bias_act.restype = i32
    L.hav_upfirdn2d.argtypes = [vp, vp, vp, i32, i64] + [i32] * 13 + [vp]
    L.hav_upfirdn2d.restype = i32
    L.hav_upfirdn2d_out_size.argtypes = [i32] * 12 + [C.POINTER(i32), C.POINTER(i32)]
    L.hav_upfirdn2d_out_size.restype = i32
    L.hav_style_demod.argtypes = [vp, vp, vp, vp, vp, vp, f32, i32, i32, i32, i32, vp]
    L.hav_style_demod.restype = i32
    L.hav_style_demod_blocks.argtypes = [i32, i32]
    L.hav_style_demod_blocks.restype = i32
    L.hav_style_demod_batched.argtypes = [vp, i32, i32, i32, vp, f32, i32, i32, i32, vp]
    L.hav_style_demod_batched.restype = i32
    L.hav_styled_epilogue.argtypes = [vp, vp, vp, vp, vp, vp, f32, f32, i32, i32, i64, i32, vp]
    L.hav_styled_epilogue.restype = i32
    L.hav_torgb.argtypes = [vp, vp, vp, vp, vp, vp, f32, i32, i32, i32, i64, vp]
    L.hav_torgb.restype = i32
    L.hav_demod_fwd.argtypes = [vp, vp, vp, vp, f32, f32, i32, i32, i32, i32, vp]
    L.hav_demod_fwd.restype = i32
    L.hav_demod_bwd.argtypes = [vp] * 8 + [f32, i32, i32, i32, i32, vp]
    L.hav_demod_bwd.restype = i32
    L.hav_triplane_gather_fwd.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, i32, vp]
    L.hav_triplane_gather_fwd.restype = i32
    L.hav_triplane_gather_bwd.argtypes = [vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, i32, vp]
    L.hav_triplane_gather_bwd.restype = i32
    L.hav_field_inputs_fwd.argtypes = [vp, C.POINTER(HavFieldParams), vp, vp, vp, vp, vp]
    L.hav_field_inputs_fwd.restype = i32
    L.hav_field_inputs_fwd_bf16.argtypes = [vp, C.POINTER(HavFieldParams), vp, vp, vp, vp, vp]
    L.hav_field_inputs_fwd_bf16.restype = i32
    L.hav_field_inputs_bwd.argtypes = [vp, vp, vp, C.POINTER(HavFieldParams), vp, vp, vp, vp, vp]
    L.hav_field_inputs_bwd.restype = i32
    L.hav_field_inputs_bwd_rows.argtypes = [vp, vp, vp, C.POINTER(HavFieldParams), vp, vp, vp, vp, i32, vp]
    L.hav_field_inputs_bwd_rows.restype = i32
    L.hav_field_inputs_bwd_fixed.argtypes = [vp, vp, vp, vp, vp, C.POINTER(HavFieldParams), vp, vp, vp, vp, vp]
    L.hav_field_inputs_bwd_fixed.restype = i32
    L.hav_field_inputs_bwd_fixed_scratch_bytes.argtypes = [C.POINTER(HavFieldParams)]
    L.hav_field_inputs_bwd_fixed_scratch_bytes.restype = i64
    L.hav_composite_fwd.argtypes = [vp] * 9 + [i64, i32, i32, i32, vp]
    L.hav_composite_fwd.restype = i32
    L.hav_composite_bwd.argtypes = [vp] * 10 + [i64, i32, i32, i32, vp]
    L.hav_composite_bwd.restype = i32
    L.hav_resample_depths.argtypes = [vp] * 5 + [i64, i32, i32, vp]
    L.hav_resample_depths.restype = i32
    L.hav_equal_linear_fwd.argtypes = [vp] * 4 + [C.c_float, C.c_float, i32, i32, i32, vp]
    L.hav_equal_linear_fwd.restype = i32
    L.hav_equal_linear_bwd.argtypes = [vp] * 6 + [C.c_float, C.c_float, i32, i32, i32, vp]
    L.hav_equal_linear_bwd.restype = i32
    L.hav_mlp_blob_bytes.restype = i64
    L.hav_mlp_pack.argtypes = [vp, C.POINTER(HavMlpWeights), vp]
    L.hav_mlp_pack.restype = i32
    for fn in (L.hav_haar_dwt, L.hav_haar_idwt):
        fn.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
        fn.restype = i32
    L.hav_style_mlp.argtypes = [vp, vp, vp, i32, i32, i32, i32, f32, f32, vp]
    L.hav_style_mlp.restype = i32
    L.hav_haar_up2.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.hav_haar_up2.restype = i32
    L.hav_conv3x3_wgrad_scratch_bytes.argtypes = [i32] * 5
    L.hav_conv3x3_wgrad_scratch_bytes.restype = i64
    L.hav_conv3x3_wgrad.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.hav_conv3x3_wgrad.restype = i32
    L.hav_gemm_packed_bytes.argtypes = [i32, i32]
    L.hav_gemm_packed_bytes.restype = i64
    L.hav_gemm_pack.argtypes = [vp, vp, i32, i32, f32, vp]
    L.hav_gemm_pack.restype = i32
    L.hav_gemm_split.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.hav_gemm_split.restype = i32
    L.hav_upconv_finish.argtypes = [vp] * 7 + [f32, f32, i32, i32, i32, i32, i32, i32, vp]
    L.hav_upconv_finish.restype = i32
    L.hav_conv3x3_packed_bytes.argtypes = [i32, i32]
    L.hav_conv3x3_packed_bytes.restype = i64
    L.hav_conv3x3_pack.argtypes = [vp, vp, i32, i32, f32, vp]
    L.hav_conv3x3_pack.restype = i32
    L.hav_conv3x3_scratch_bytes.argtypes = [i32] * 5
    L.hav_conv3x3_scratch_bytes.restype = i64
    L.hav_conv3x3_split.argtypes = [vp] * 8 + [f32, f32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    L.hav_conv3x3s2_split.argtypes = [vp] * 8 + [f32, f32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    L.hav_conv3x3s2_scratch_bytes.argtypes = [i32] * 6
    L.hav_conv3x3s2_scratch_bytes.restype = C.c_int64
    L.hav_conv3x3s2_split.restype = i32
    L.hav_conv3x3_pack_t.argtypes = [vp, vp, i32, i32, f32, vp]
    L.hav_conv3x3_pack_t.restype = i32
    L.hav_conv3x3_wgrad_mod.argtypes = [vp, vp, vp, vp, f32, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.hav_conv3x3_wgrad_mod.restype = i32
    L.hav_conv3x3s2_wgrad.argtypes = [vp, vp, vp, vp, f32, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.hav_conv3x3s2_wgrad.restype = i32
    L.hav_conv3x3s2_wgrad_scratch_bytes.argtypes = [i32] * 5
    L.hav_conv3x3s2_wgrad_scratch_bytes.restype = i64
    L.hav_conv_block_bwd.argtypes = [vp] * 11 + [f32, f32, i32, i32, i32, i32, i64, vp]
    L.hav_conv_block_bwd.restype = i32
    L.hav_mod_input_bwd.argtypes = [vp, vp, vp, vp, i32, i32, i64, vp]
    L.hav_mod_input_bwd.restype = i32
    L.hav_absmax.argtypes = [vp, vp, i64, vp]
    L.hav_absmax.restype = i32
    L.hav_debug_nonfinite.argtypes = [vp, vp, i64, vp]
    L.hav_debug_nonfinite.restype = i32
    L.hav_conv3x3_split.restype = i32
    L.hav_upsample3d_2x_fwd.argtypes = [vp, vp, i64, i32, i32, i32, vp]
    L.hav_upsample3d_2x_fwd.restype = i32
    L.hav_upsample3d_2x_bwd.argtypes = [vp, vp, i64, i32, i32, i32, vp]
    for fn in (L.hav_im2col3d, L.hav_col2im3d):
        fn.argtypes = [vp, vp, i32, i32, vp]
        fn.restype = i32
    L.hav_upsample3d_2x_bwd.restype = i32
    L.hav_mlp_train_blob_bytes.restype = i64
    L.hav_mlp_train_ops_bytes.argtypes = [i64]
    L.hav_mlp_train_ops_bytes.restype = i64
    L.hav_mlp_train_partial_bytes.argtypes = [i64]
    L.hav_mlp_train_partial_bytes.restype = i64
    L.hav_mlp_train_pack.argtypes = [vp, C.POINTER(HavMlpWeights), vp]
    L.hav_mlp_train_pack.restype = i32
    L.hav_mlp_train_fwd.argtypes = [vp, vp, vp, i64, vp]
    L.hav_mlp_train_fwd.restype = i32
    L.hav_mlp_train_bwd.argtypes = [vp, C.POINTER(HavMlpGrads), i32, vp, vp, vp, vp, vp, i64, vp]
    L.hav_mlp_train_bwd.restype = i32
    L.hav_mlp_train_fwd_xbf16.argtypes = [vp, vp, vp, i64, vp]
    L.hav_mlp_train_fwd_xbf16.restype = i32
    L.hav_mlp_train_bwd_xbf16.argtypes = [vp, C.POINTER(HavMlpGrads), i32, vp, vp, vp, vp, vp, i64, vp]
    L.hav_mlp_train_bwd_xbf16.restype = i32
    L.hav_triplane_prepare.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    L.hav_triplane_prepare.restype = i32
    L.hav_triplane_prepared_bytes.argtypes = [i32, i32, i32]
    L.hav_triplane_prepared_bytes.restype = i64
    L.hav_render_rays.argtypes = [C.POINTER(HavRenderParams)] + [vp] * 10 + [C.POINTER(HavRenderOut), vp]
    L.hav_render_rays.restype = i32
    L.hav_render_workspace_bytes.argtypes = [C.POINTER(HavRenderParams)]
    L.hav_render_workspace_bytes.restype = i64
    L.hav_render_variant_name.argtypes = [C.POINTER(HavRenderParams), i32, i32, C.c_char_p, i32]
    L.hav_render_variant_name.restype = i32
    L.hav_gen_rays.argtypes = [vp, i32, i32, C.POINTER(f32), C.POINTER(f32), f32, f32, i32, i32, vp]
    L.hav_gen_rays.restype = i32
    L.hav_debug_mlp_layer.argtypes = [vp, vp, vp, i32, i32, i64, vp]
    L.hav_debug_mlp_layer.restype = i32
    _lib = L
    return L


_ERR = {-1: "HAV_EINVAL (bad size / null pointer / inconsistent arguments)",
        -2: "HAV_EUNSUP (valid for the reference, not supported by this build)"}


def check(rc, what):
    """0 -> ok; negative -> library refusal; positive -> hipError_t from the launch."""
    if rc == 0:
        return
    if rc < 0:
        raise RuntimeError(f"{what}: {_ERR.get(rc, rc)}")
    raise RuntimeError(f"{what}: HIP error {rc}")
