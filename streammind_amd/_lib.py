"""ctypes binding of libstreammind_hip.so (the C ABI declared in include/streammind_hip.h).

The product path has NO fallback: if the shared library is missing or fails to load, importing any
compute entry point raises.  Build it with `python -m streammind_amd.build` (or `__graft_entry__.build()`).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libstreammind_hip.so")
# A/B builds for tools/ (kernel variants compiled side by side): STREAMMIND_HIP_LIB=/path/to/other.so
LIB_PATH = os.environ.get("STREAMMIND_HIP_LIB", LIB_PATH)

SM_ACT_NONE, SM_ACT_QUICK_GELU, SM_ACT_LEAKY_RELU, SM_ACT_SOFTPLUS, SM_ACT_SILU, SM_ACT_GELU, SM_ACT_SWIGLU_DUAL = 0, 1, 2, 3, 4, 5, 6
SM_X_BF16, SM_X_F32 = 0, 1
SM_W_BF16, SM_W_FP8, SM_W_FP8_MFMA = 0, 1, 2
SM_OP_BF16, SM_OP_F16 = 0, 1
SM_DT_BF16, SM_DT_F32, SM_DT_F16 = 0, 1, 2

vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class sm_linear_t(C.Structure):
    _fields_ = [
        ("w", vp), ("w2", vp), ("N", i32), ("K", i32),
        ("x", vp), ("x_dtype", i32), ("precise", i32), ("M", i32), ("ldx", i32),
        ("bias", vp), ("act", i32), ("residual", vp), ("ldr", i32),
        ("out_f32", vp), ("out_bf16", vp), ("ldo", i32), ("ldo_bf16", i32),
        ("remap_in", i32), ("remap_out", i32), ("remap_off", i32),
        ("vt", vp), ("vt_n0", i32), ("vt_S", i32), ("vt_dh", i32), ("vt_ld", i32),
        ("w_dtype", i32), ("w_scale", vp), ("w2_scale", vp),
        ("norm_gamma", vp), ("norm_eps", f32), ("tile_hint", i32), ("op_dtype", i32),
        ("post_ln_gamma", vp), ("post_ln_beta", vp), ("post_ln_eps", f32), ("post_ln_out", vp), ("post_ln_ldo", i32),
        ("post_ln_out_f32", vp), ("post_ln_act", i32), ("x_rep", i32), ("x_rep_dh", i32),
        ("fold_stats_out", vp), ("fold_stats_in", vp), ("fold_g", vp), ("fold_c", vp), ("fold_eps", f32),
    ]


class sm_config_t(C.Structure):
    _fields_ = [
        ("vit_image", i32), ("vit_patch", i32), ("vit_hidden", i32), ("vit_heads", i32), ("vit_mlp", i32),
        ("vit_layers_run", i32), ("vit_eps", f32), ("img_mean", f32 * 3), ("img_std", f32 * 3),
        ("conn_mm_hidden", i32), ("conn_d_model", i32), ("conn_d_state", i32), ("conn_d_conv", i32),
        ("conn_expand", i32), ("conn_dt_rank", i32), ("conn_eps", f32),
        ("gate_hidden", i32), ("gate_layers", i32), ("gate_heads", i32), ("gate_kv_heads", i32), ("gate_mlp", i32),
        ("gate_eps", f32),
        ("llm_hidden", i32), ("llm_layers", i32), ("llm_heads", i32), ("llm_kv_heads", i32), ("llm_mlp", i32),
        ("llm_vocab", i32), ("llm_eps", f32), ("llm_rope_theta", f32),
        ("max_frames_per_call", i32), ("gate_precise", i32), ("weights_fp8", i32), ("vit_fp16", i32), ("llm_fp16", i32), ("proj_fp16", i32), ("llm_sliding_window", i32),
    ]


class sm_jpeg_huff_t(C.Structure):
    _fields_ = [("fast", C.c_uint16 * 512), ("maxcode", C.c_int32 * 18), ("valoff", C.c_int32 * 17), ("vals", C.c_uint8 * 256)]


class sm_jpeg_scan_t(C.Structure):
    _fields_ = [("scan_offset", C.c_uint32), ("scan_len", C.c_uint32), ("restart", C.c_int32), ("n_intervals", C.c_int32), ("ncomp", C.c_int32),
                ("qt", (C.c_uint16 * 64) * 3), ("dc", sm_jpeg_huff_t * 3), ("ac", sm_jpeg_huff_t * 3)]


class sm_jpeg_info_t(C.Structure):
    _fields_ = [("width", i32), ("height", i32), ("ncomp", i32), ("hs", i32 * 3), ("vs", i32 * 3), ("mcu_w", i32), ("mcu_h", i32),
                ("mcus_x", i32), ("mcus_y", i32), ("blocks_x", i32 * 3), ("blocks_y", i32 * 3), ("coef_offset", i32 * 3), ("coef_count", i32)]


# name -> (restype, argtypes); every symbol include/streammind_hip.h declares
SIGNATURES = {
    "sm_last_error": (C.c_char_p, []),
    "sm_abi_version": (i32, []),
    "sm_packed_elems": (sz, [i32, i32]),
    "sm_pack_weight": (i32, [vp, i32, i32, i32, vp, vp]),
    "sm_packed_fp8_bytes": (sz, [i32, i32]),
    "sm_quant_pack_weight_fp8": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "sm_linear": (i32, [C.POINTER(sm_linear_t), vp]),
    "sm_norm": (i32, [vp, i32, i32, i32, vp, vp, f32, i32, vp, vp, i32, vp]),
    "sm_norm_ex": (i32, [vp, i32, i32, i32, vp, vp, f32, i32, vp, vp, i32, i32, vp]),
    "sm_preprocess_patches": (i32, [vp, i32, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp, i32, vp, i32, vp]),
    "sm_ingest_tmp_bytes": (sz, [i32, i32, i32, i32, i32]),
    "sm_ingest_frames": (i32, [vp, i32, i32, i32, i32, vp, i32, vp, vp, vp]),
    "sm_patchify_pixels": (i32, [vp, i32, i32, i32, i32, i32, vp, i32, i32, vp]),
    "sm_pool_rows": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "sm_vit_cls_rows": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "sm_vit_attention": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "sm_pool_patches": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "sm_mamba_conv_step": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    "sm_mamba_ssm_step": (i32, [vp, vp, vp, i32, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    "sm_repeat_kv": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "sm_dwconv3x3_nhwc": (i32, [vp, i32, i32, i32, i32, vp, vp, vp]),
    "sm_se_scale": (i32, [vp, vp, i32, i32, i32, vp, vp, i32, vp]),
    "sm_add_act": (i32, [vp, vp, sz, i32, vp, vp, i32, vp]),
    "sm_conv3d_patches": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "sm_avgpool3d_nhwc": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp]),
    "sm_gate_decide": (i32, [vp, i32, vp, vp]),
    "sm_embed_splice": (i32, [vp, i32, vp, vp, i32, vp, vp]),
    "sm_rope_kv_append": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp]),
    "sm_llm_attention": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "sm_llm_decode_attention": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, vp, vp]),
    "sm_llm_attention_window": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "sm_llm_decode_attention_window": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp]),
    "sm_swiglu": (i32, [vp, i32, i32, vp, vp]),
    "sm_argmax": (i32, [vp, i32, vp, vp]),
    "sm_model_create": (i32, [C.POINTER(sm_config_t), C.POINTER(vp)]),
    "sm_model_load_tensor": (i32, [vp, C.c_char_p, vp, i32, i32, C.POINTER(C.c_int64), vp]),
    "sm_model_finalize": (i32, [vp, vp]),
    "sm_model_destroy": (None, [vp]),
    "sm_model_set_fp8_mode": (i32, [vp, i32]),
    "sm_model_missing": (i32, [vp, C.c_char_p, sz]),
    "sm_vit_encode": (i32, [vp, vp, i32, vp, vp, vp, vp]),
    "sm_vit_encode_pixels": (i32, [vp, vp, i32, i32, vp, vp, vp]),
    "sm_stream_open": (i32, [vp, i32, i32, C.POINTER(vp)]),
    "sm_stream_reset": (i32, [vp, vp]),
    "sm_stream_close": (None, [vp]),
    "sm_stream_push_pooled": (i32, [vp, vp, i32, vp, vp, vp]),
    "sm_stream_push_frames": (i32, [vp, vp, i32, vp, vp, vp]),
    "sm_stream_push_frames_pipelined": (i32, [vp, vp, i32, vp, vp, vp]),
    "sm_stream_join": (i32, [vp, vp]),
    "sm_stream_pass_ticket": (i32, [vp]),
    "sm_stream_join_ticket": (i32, [vp, i32, vp]),
    "sm_stream_num_frames": (i32, [vp]),
    "sm_stream_tokens": (vp, [vp]),
    "sm_stream_kv_len": (i32, [vp]),
    "sm_stream_set_kv_len": (i32, [vp, i32]),
    "sm_stream_kv_capacity": (i32, [vp]),
    "sm_set_vit_ln_fold": (i32, [i32]),
    "sm_set_vit_frame_lanes": (i32, [i32]),
    "sm_set_prefill_attention_kernel": (i32, [i32]),
    "sm_llm_prefill": (i32, [vp, vp, i32, vp]),
    "sm_llm_forward_logits": (i32, [vp, vp, i32, vp, vp]),
    "sm_cross_entropy": (i32, [vp, i32, i32, i32, vp, i32, vp, vp, vp]),
    "sm_cosine_rows": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "sm_llm_decode": (i32, [vp, i32, vp, vp]),
    "sm_stream_set_next_token": (i32, [vp, vp, vp]),
    "sm_stream_logits": (vp, [vp]),
    "sm_stream_read_tokens": (i32, [vp, i32, i32, vp, vp]),
    "sm_stream_read_state": (i32, [vp, vp, vp, vp]),
    "sm_stream_read_logits": (i32, [vp, vp, vp, vp]),
    "sm_stream_write_tokens": (i32, [vp, i32, i32, vp, vp]),
    "sm_group_create": (i32, [C.POINTER(vp), i32, C.POINTER(vp)]),
    "sm_group_destroy": (None, [vp]),
    "sm_group_size": (i32, [vp]),
    "sm_group_push_frames": (i32, [vp, vp, i32, vp, vp, vp]),
    "sm_group_push_pooled": (i32, [vp, vp, i32, vp, vp, vp]),
    "sm_group_llm_decode": (i32, [vp, C.POINTER(C.c_int32), i32, vp, vp]),
    "sm_jpeg_info": (i32, [vp, sz, C.POINTER(sm_jpeg_info_t)]),
    "sm_jpeg_decode_coefs": (i32, [vp, sz, C.POINTER(sm_jpeg_info_t), vp, vp]),
    "sm_jpeg_planes_bytes": (sz, [C.POINTER(sm_jpeg_info_t), i32]),
    "sm_jpeg_reconstruct": (i32, [vp, vp, C.POINTER(sm_jpeg_info_t), i32, vp, vp, vp]),
    "sm_jpeg_scan_prepare": (i32, [vp, sz, C.POINTER(sm_jpeg_info_t), C.POINTER(sm_jpeg_scan_t)]),
    "sm_jpeg_entropy_decode": (i32, [vp, sz, vp, vp, C.POINTER(sm_jpeg_info_t), i32, vp, vp, vp, vp]),
    "sm_jpeg_sync_rounds": (i32, [vp, C.POINTER(C.c_int32), i32]),
    "sm_jpeg_entropy_decode_sync": (i32, [vp, sz, sz, vp, vp, C.POINTER(sm_jpeg_info_t), i32, vp, vp, vp, vp]),
    "sm_comm_handle_bytes": (i32, []),
    "sm_comm_init": (i32, [i32, i32, i32, i32, C.POINTER(vp)]),
    "sm_comm_export": (i32, [vp, vp]),
    "sm_comm_connect": (i32, [vp, vp]),
    "sm_comm_destroy": (None, [vp]),
    "sm_comm_post": (i32, [vp, vp, i32, vp]),
    "sm_comm_collect": (i32, [vp, vp, vp, vp]),
    "sm_comm_host_counts": (i32, [vp, i32, C.POINTER(C.c_int32)]),
    "sm_comm_poll_counts": (i32, [vp, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sm_comm_recollect": (i32, [vp, vp, vp, vp]),
    "sm_allgather_gated": (i32, [vp, vp, i32, vp, vp, vp]),
    "sm_comm_max_rows": (i32, [vp]),
    "sm_comm_tick": (i32, [vp]),
    "sm_prof_enable": (i32, [i32]),
    "sm_prof_reset": (i32, []),
    "sm_prof_read": (i32, [i32, C.POINTER(i32), C.POINTER(f32)]),
    "sm_prof_read_tag": (i32, [i32, C.c_longlong, C.POINTER(i32), C.POINTER(f32)]),
}

_lib = None


class StreamMindHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen the library and type every entry point.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StreamMindHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m streammind_amd.build`. "
            "There is no CPU fallback on the product path.")
    # PyTorch-ROCm ships its own HIP runtime (same SONAME libamdhip64.so.7); import it first so that this library
    # binds to the SAME runtime instance torch allocates device memory and streams from.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    older = os.environ.get("STREAMMIND_HIP_LIB_OLDER") == "1"      # A/B runs against the library of an EARLIER revision (tools/build_base.sh): entry points it lacks are skipped
    for name, (res, args) in SIGNATURES.items():
        if older and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> int:
    if rc < 0:
        msg = load().sm_last_error().decode(errors="replace")
        raise StreamMindHipError(f"{what or 'libstreammind_hip'} failed ({rc}): {msg}")
    return rc
