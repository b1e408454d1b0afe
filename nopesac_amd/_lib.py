"""ctypes binding of libnopesac_hip.so (the C ABI declared in include/nopesac_hip.h).

There is NO fallback: if the shared library is missing or a symbol cannot be resolved, importing
the ops raises.  Build it with `python -m nopesac_amd.build` (or `__graft_entry__.build()`).
"""
from __future__ import annotations

import ctypes
import os
import re
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libnopesac_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "nopesac_hip.h")

P, I, L, F = c_void_p, c_int, c_int64, c_float

# name -> argtypes (return type is int unless listed in _RESTYPE)
SIGNATURES = {
    "nopesac_version": [],
    "nopesac_last_error": [],
    "nopesac_conv2d_nhwc": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, L, L, L, L, I, I, I, P],
    "nopesac_conv2d_nhwc_bfrag": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, L, L, L, I, I, I, P],
    "nopesac_conv2d_nhwc_p8": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, L, L, L, I, I, I, P],
    "nopesac_conv2d_nhwc_p8_sk": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, L, L, L, I, I, I, P, L, P],
    "nopesac_conv2d_p8_sk_workspace_bytes": [],
    "nopesac_conv2d_nhwc_p8n": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, L, L, I, I, P],
    "nopesac_conv2d_nhwc_p8n_splitk": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, L, L, I, I, I, P, L, P],
    "nopesac_conv2d_nhwc_ex": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, L, L, L, L, I, I, I, I, P],
    "nopesac_stem_fused_bf16": [P, P, P, P, P, I, I, I, P],
    "nopesac_stem_fused_raw_bf16": [P, P, P, P, P, P, P, I, I, I, P],
    "nopesac_stem_fused_raw_shifted_bf16": [P, P, P, P, P, P, I, I, I, P],
    "nopesac_bottleneck_tail_bf16": [P] * 9 + [I] * 9 + [P] * 4 + [I, P, P],
    "nopesac_bottleneck_tail_bf16_ex": [P] * 9 + [I] * 9 + [P] * 4 + [I, P, I, P],
    "nopesac_conv2d_nhwc_fp8": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, L, L, L, I, I, I, P],
    "nopesac_gnn_layer_bf16": [P, I, P, I, P, I, I, I, P, P] + [P] * 10 + [P],
    "nopesac_gnn_layer_bf16_pf": [P, I, P, I, P, I, I, I, P, P] + [P] * 10 + [P, I, P],
    "nopesac_encoder_tail_bf16": [P] * 13 + [I] + [P] * 3 + [I, P],
    "nopesac_resize_bilinear_u8": [P, I, I, I, P, I, I, P],
    "nopesac_resize_bilinear_u8_batch": [P, I, L, I, I, I, P, I, I, I, P],
    "nopesac_mask_head_bf16": [P] * 9 + [I] * 5 + [P],
    "nopesac_mask_operands": [P, I, P, P, I, I, I, P],
    "nopesac_decoder_tail_bf16": [P] * 13 + [I] + [P] * 4 + [I, P],
    "nopesac_transformer_tail_bf16": [P] * 13 + [I] + [P] * 4 + [I, I, P, P, P, I, P, P, P, I, I, P],
    "nopesac_transformer_tail_bf16_pf": [P] * 13 + [I] + [P] * 4 + [I, I, P, P, P, I, P, P, P, I, I, P, P, I, I, P],
    "nopesac_conv3x3_c64_bf16": [P, P, P, P, P, I, I, I, I, P],
    "nopesac_conv3x3_halo_bf16": [P, P, P, P, P, I, I, I, I, I, I, I, P],
    "nopesac_rle_labels": [P, P, P, P, P, I, I, I, I, P],
    "nopesac_rle_transitions": [P, P, P, P, P, I, I, I, P],
    "nopesac_rle_compress_host": [P, I, I, I, P, I, P],
    "nopesac_rle_compress_device": [P, P, P, I, I, I, P, P, P, P, P],
    "nopesac_rle_compress_device_capped": [P, P, P, I, I, I, P, P, P, L, P],
    "nopesac_decode_masks": [P, P, P, P, P, P, I, I, I, I, P],
    "nopesac_rle_compress_batch_host": [P, P, P, I, I, I, P, L, P, P],
    "nopesac_preprocess_nchw_to_nhwc": [P, P, P, P, I, I, I, I, I, I, P],
    "nopesac_maxpool_nhwc": [P, P, I, I, I, I, I, I, I, I, P],
    "nopesac_upsample2x_bilinear_nhwc": [P, P, P, I, I, I, I, I, I, P],
    "nopesac_upsample2x_nearest_add_nhwc": [P, P, P, I, I, I, I, I, P],
    "nopesac_groupnorm_nhwc": [P, P, P, P, I, I, I, I, F, I, I, P, P],
    "nopesac_layernorm": [P, P, P, P, P, P, I, P, I, I, F, P],
    "nopesac_layernorm_ex": [P, P, P, P, P, P, I, P, P, P, I, I, F, P],
    "nopesac_add_rows": [P, P, P, I, I, I, P],
    "nopesac_softmax_rows": [P, P, I, I, P],
    "nopesac_softmax_rows_pad": [P, P, I, I, I, I, P],
    "nopesac_add_rows_bf16": [P, P, P, P, I, I, I, P],
    "nopesac_concat_cols": [P, I, P, I, P, I, P],
    "nopesac_metric_rows": [P, P, P, P, P, P, P, P, I, P, I, P],
    "nopesac_attention_small": [P, L, P, L, P, L, P, L, I, I, I, I, F, P, P, P],
    "nopesac_attention_small_bf16": [P, L, P, L, P, L, P, L, I, I, I, I, F, P, P, P],
    "nopesac_attention_small_bf16io": [P, L, P, L, P, L, P, L, I, I, I, I, F, P, P, P],
    "nopesac_transpose_hw_rows": [P, P, I, I, I, I, P],
    "nopesac_postselect_planes": [P, P, P, P, I, I, I, I, I, I, I, F, F, F, P, P, P, P, P, P, P, P, P, P, P],
    "nopesac_postselect_planes_ex": [P, P, P, P, I, I, I, I, I, I, I, F, F, F, P, P, P, P, P, P, P, P, P, P, I, P],
    "nopesac_matcher_sinkhorn": [P, P, P, P, P, P, P, F, F, I, F, I, I, P, P, P],
    "nopesac_geo_sequence": [P, P, P, P, P, P, P, I, I, I, P, P, P, P, P, P],
    "nopesac_ransac_score_maps": [P, P, P, P, P, P, I, I, P, P, P, P, P, P, P, P, P, P],
    "nopesac_ransac_soft_vote": [P] * 21 + [I, I, I] + [P] * 6 + [P],
    "nopesac_plane_cam_ref_losses": [P] * 11 + [I, I, F, P, P],
    "nopesac_camera_pose_loss": [P, P, P, I, P, I, I, F, F, P, P],
    "nopesac_refilter_assignment": [P, P, P, P, P, P, P, I, I, P, P],
    "nopesac_tape_create": [P, ctypes.POINTER(c_void_p), P],
    "nopesac_tape_create_ex": [P, I, ctypes.POINTER(c_void_p), P],
    "nopesac_tape_replay": [P, P],
    "nopesac_tape_replay_on": [P, P, P, I],
    "nopesac_posenet_branch_tail_bf16": [P, P, P, P, P, P, P, I, I, I, I, P],
    "nopesac_tape_destroy": [P],
    "nopesac_lds_canary": [I, I, L, I, P, P, I, P],
    "nopesac_force_k_select": [P, I, P, P, P, I, I, I, I, P, P, P],
    "nopesac_normalize_rows": [P, P, I, I, I, P],
    "nopesac_count_nonfinite": [P, L, P, P],
    "nopesac_count_nonfinite_batch": [P, P, I, P, P],
    "nopesac_clock_probe": [P, L, P],
    "nopesac_u8_to_f32": [P, P, L, P],
    "nopesac_gather_bytes": [P, P, P, P, I, P, P],
    "nopesac_jpeg_huffman": [P, P, P, P, P, I, P, L, P, P, P],
    "nopesac_jpeg_huffman_parallel": [P, P, P, I, P, L, P, L, P, P, P, P, P, P, P, P],
    "nopesac_jpeg_prepare_scan": [P, L, I, P, L, P, P, P, L, P],
    "nopesac_jpeg_batch_scan_host": [P, I, I, I, P, P],
    "nopesac_jpeg_batch_fill_host": [P, P, P, P, P, P, P, P, P],
    "nopesac_jpeg_batch_free_host": [P],
    "nopesac_jpeg_idct": [P, P, P, I, I, P, P, P],
    "nopesac_jpeg_color": [P, P, I, I, P, P, I, P],
    "nopesac_png_info_host": [P, L, P, P, P, P],
    "nopesac_png_decode_host": [P, L, P, L, I],
    "nopesac_png_decode_files_host": [P, I, P, L, I, I, I, I, P],
    "nopesac_inflate_zlib_host": [P, L, P, L],
    "nopesac_refine_losses_backward": [P] * 11 + [I, I, F] + [P] * 7 + [P],
    "nopesac_refine_vote_backward": [P] * 15 + [I, I] + [P] * 20 + [P],
    "nopesac_refine_score_maps_backward": [P] * 6 + [I, I] + [P] * 7 + [P],
    "nopesac_transpose_f32": [P, I, I, L, P, P],
    "nopesac_col_sum_f32": [P, I, I, L, P, P],
    "nopesac_relu_backward_f32": [P, P, L, P, P],
    "nopesac_normalize_rows_backward": [P, P, I, I, I, P, P],
    "nopesac_camera_pose_loss_backward": [P, P, P, I, P, I, I, F, F, P, P, P, P, P, P],
    "nopesac_sumsq_accumulate_f32": [P, L, P, P],
    "nopesac_clip_coefficient": [P, F, P, P],
    "nopesac_scale_by_f32": [P, L, P, P],
    "nopesac_adamw_step": [P, P, P, P, L, F, F, F, F, F, I, P],
    "nopesac_sgd_step": [P, P, P, L, F, F, F, I, P],
    "nopesac_mlp_padded_k": [I, I],
    "nopesac_mlp_packed_elems": [I, I],
    "nopesac_mlp_chain_bf16": [P, P],
}
_RESTYPE = {"nopesac_jpeg_prepare_scan": c_int64, "nopesac_last_error": c_char_p, "nopesac_rle_compress_batch_host": c_int64, "nopesac_mlp_packed_elems": c_int64,
            "nopesac_conv2d_p8_sk_workspace_bytes": c_int64, "nopesac_inflate_zlib_host": c_int64, "nopesac_jpeg_batch_scan_host": c_void_p, "nopesac_jpeg_batch_free_host": None}

MLP_MAX_IN, MLP_MAX_WIDTH, MLP_MAX_LAYERS = 1280, 1024, 12       # NOPESAC_MLP_* of the header


class MlpLayer(ctypes.Structure):           # nopesac_mlp_layer
    _fields_ = [("w", c_void_p), ("bias", c_void_p), ("out", c_void_p), ("out_ld", c_int64), ("K", c_int), ("N", c_int), ("act", c_int),
                ("reserved", c_int)]


class MlpChain(ctypes.Structure):           # nopesac_mlp_chain
    _fields_ = [("x", c_void_p), ("x_ld", c_int64), ("xb", c_void_p), ("xb_ld", c_int64), ("x_width", c_int), ("xb_width", c_int),
                ("xb_rows_per", c_int), ("rows", c_int), ("n_layers", c_int), ("reserved", c_int), ("layers", MlpLayer * MLP_MAX_LAYERS)]


_lib = None


def declared_symbols() -> list:
    """Every entry point include/nopesac_hip.h declares."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nopesac_\w+)\s*\(", text)))


def load():
    """Load (once) and type the library.  Raises RuntimeError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    lib_path = os.environ.get("NOPESAC_AB_LIBRARY") or LIB_PATH           # A/B runs only: another BUILD of this library (e.g. last round's)
    if not os.path.exists(lib_path):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built.  Run `python -m nopesac_amd.build` "
            "(needs hipcc for gfx950).  nopesac_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64; whichever copy is loaded first serves the whole process.  Import torch first so
    # that this library (linked against the same SONAME) shares torch's runtime, streams and device context - loading the
    # system copy first leaves the kernels of this library without a device ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    lib = ctypes.CDLL(lib_path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, c_int)
    _lib = lib
    return lib


class HipKernelError(RuntimeError):
    pass


def check(rc: int, name: str):
    if rc != 0:
        msg = load().nopesac_last_error()
        raise HipKernelError(f"{name} failed (rc={rc}): {msg.decode() if msg else ''}")
