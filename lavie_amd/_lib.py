"""ctypes binding of liblavie_hip.so (C ABI: include/lavie_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C lavie_amd/csrc`.  There is no
fallback: if the shared object is missing or a call fails, a RuntimeError is raised."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LAVIE_HIP_LIB") or os.path.join(_HERE, "liblavie_hip.so")   # env override: A/B builds
ABI_VERSION = 7
FUSED_DEFAULT = 0x137    # lavie_debug_fused_mask: bits 0, 1, 2, 4, 5, 8 (include/lavie_hip.h)
MAX_LEVELS = 8

c_void_p, c_int, c_float, c_ll, c_char_p = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_char_p
c_float_p = C.c_void_p      # fp32 device pointers are passed as raw addresses


class UNetConfigC(C.Structure):
    _fields_ = [
        ("struct_size", c_int),
        ("in_channels", c_int), ("out_channels", c_int), ("num_levels", c_int),
        ("block_out_channels", c_int * MAX_LEVELS), ("attn_levels", c_int * MAX_LEVELS),
        ("layers_per_block", c_int), ("heads", c_int), ("cross_attention_dim", c_int), ("norm_groups", c_int),
        ("norm_eps", c_float), ("rotary_dim", c_int), ("rel_buckets", c_int), ("rel_max_distance", c_int),
        ("sparse_causal_attn1", c_int), ("temporal_plain", c_int), ("ff_before_temporal", c_int),
        ("vsr_blocks", c_int), ("only_cross_attention", c_int * MAX_LEVELS),
        ("vsr_temporal_modules", c_int), ("num_class_embeds", c_int),
    ]


# name -> (restype, argtypes); every symbol declared in include/lavie_hip.h
SIGNATURES = {
    "lavie_last_error": (c_char_p, []),
    "lavie_abi_version": (c_int, []),
    "lavie_linear_lnfold_f16": (c_int, [c_void_p, c_void_p, c_float_p, c_float_p, c_float_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "lavie_linear_f16": (c_int, [c_void_p, c_int, c_void_p, c_float_p, c_float_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                  c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lavie_conv3x3_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_float_p,
                                   c_float_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_void_p, c_void_p]),
    "lavie_pack_conv3x3_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "lavie_temporal_conv_f16": (c_int, [c_void_p, c_int, c_void_p, c_float_p, c_float_p, c_int, c_int, c_void_p, c_void_p, c_int,
                                         c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "lavie_pack_temporal_conv_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "lavie_pack_geglu_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_float_p, c_int, c_int, c_void_p]),
    "lavie_geglu_mlp_image_bytes": (c_ll, [c_int]),
    "lavie_geglu_mlp_bias_floats": (c_ll, [c_int]),
    "lavie_pack_geglu_mlp_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float_p, c_void_p]),
    "lavie_geglu_mlp_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                     c_float, c_void_p]),
    "lavie_temporal_block_image_bytes": (c_ll, [c_int, c_int, c_int, c_int]),
    "lavie_pack_temporal_block_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "lavie_temporal_block_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_float_p, c_float_p,
                                          c_float_p, c_float_p, c_float_p, c_float_p, c_int, c_float, c_float, c_void_p]),
    "lavie_cross_block_image_bytes": (c_ll, [c_int, c_int]),
    "lavie_pack_cross_block_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "lavie_bind_cross_block_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "lavie_cross_block_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_float_p, c_float_p,
                                       c_float_p, c_float_p, c_int, c_float, c_float, c_void_p]),
    "lavie_pack_conv3x3_parity_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "lavie_upsample_conv3x3_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "lavie_upsample_conv3x3_f16": (c_int, [c_void_p, c_void_p, c_float_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "lavie_proj_qkv_image_bytes": (c_ll, [c_int]),
    "lavie_pack_proj_qkv_f16": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "lavie_group_norm_affine_f16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float_p, c_float_p, c_float, c_float_p, c_float_p, c_void_p]),
    "lavie_proj_qkv_f16": (c_int, [c_void_p, c_float_p, c_int, c_void_p, c_float_p, c_float_p, c_float_p, c_float, c_void_p, c_void_p,
                                    c_int, c_int, c_void_p]),
    "lavie_group_norm_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_float_p, c_float_p, c_float,
                                      c_int, c_float_p, c_void_p, c_void_p]),
    "lavie_group_norm_ws_floats": (c_ll, [c_int, c_int]),
    "lavie_layer_norm_f16": (c_int, [c_void_p, c_float_p, c_float_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "lavie_attention_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                     c_int, c_int, c_int, c_float, c_void_p]),
    "lavie_sparse_causal_attention_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int,
                                                   c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "lavie_temporal_attention_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float_p,
                                              c_float_p, c_float_p, c_int, c_float, c_void_p]),
    "lavie_relpos_buckets": (c_int, [c_int, c_int, c_int, C.POINTER(c_int)]),
    "lavie_cfg_ddpm_step": (c_int, [c_void_p, c_float_p, c_float_p, c_void_p, c_ll, c_float, c_float, c_float, c_float,
                                     c_float, c_float, c_void_p]),
    "lavie_latents_to_model_input": (c_int, [c_float_p, c_void_p, c_ll, c_void_p]),
    "lavie_cfg_sampler_step": (c_int, [c_void_p, c_float_p, c_float_p, c_void_p, c_ll, c_float, c_float, c_float, c_float,
                                        c_float, c_float, c_float, c_void_p]),
    "lavie_latents_to_scaled_model_input": (c_int, [c_float_p, c_void_p, c_ll, c_float, c_void_p]),
    "lavie_sampler_step": (c_int, [c_void_p, c_float_p, c_float_p, c_void_p, c_ll, c_float, c_float, c_float, c_float, c_float,
                                    c_float, c_void_p]),
    "lavie_latents_to_scaled_model_input1": (c_int, [c_float_p, c_void_p, c_ll, c_float, c_void_p]),
    "lavie_debug_force_tile": (c_int, [c_int]),
    "lavie_debug_force_splits": (c_int, [c_int]),
    "lavie_debug_fused_mask": (c_int, [c_int]),
    "lavie_debug_gn_producer_count": (c_ll, []),
    "lavie_debug_temporal_block_dump": (c_int, [c_void_p]),
    "lavie_debug_rowfuse_variant": (c_int, [c_int]),
    "lavie_debug_rowfuse_stamps": (c_int, [c_void_p]),
    "lavie_debug_conv_tap_major": (c_int, [c_int]),
    "lavie_debug_attention_qt": (c_int, [c_int]),
    "lavie_debug_temporal_budget": (c_int, [c_int]),
    "lavie_debug_patch_stamps": (c_int, [c_void_p]),
    "lavie_debug_ppx_stamps": (c_int, [c_void_p]),
    "lavie_profile_begin": (c_int, [C.c_uint, c_int]),
    "lavie_profile_end": (c_int, [c_void_p, C.POINTER(c_ll), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                   C.POINTER(C.c_double)]),
    "lavie_unet_config_size": (c_int, []),
    "lavie_unet_create": (c_int, [C.POINTER(UNetConfigC), C.POINTER(c_void_p)]),
    "lavie_unet_destroy": (c_int, [c_void_p]),
    "lavie_unet_num_params": (c_int, [c_void_p]),
    "lavie_unet_param_info": (c_int, [c_void_p, c_int, C.POINTER(c_char_p), C.POINTER(c_ll)]),
    "lavie_unet_set_param": (c_int, [c_void_p, c_char_p, c_void_p, c_ll]),
    "lavie_unet_finalize": (c_int, [c_void_p, c_void_p]),
    "lavie_unet_prepare": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "lavie_unet_cache_context": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "lavie_unet_set_ln_fold": (c_int, [c_void_p, c_int]),
    "lavie_unet_set_cfg_shared_input": (c_int, [c_void_p, c_int]),
    "lavie_unet_weight_bytes": (c_ll, [c_void_p]),
    "lavie_unet_workspace_bytes": (c_ll, [c_void_p]),
    "lavie_unet_forward": (c_int, [c_void_p, c_void_p, c_float_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p]),
    "lavie_unet_forward_graph": (c_int, [c_void_p, c_void_p, c_float_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                          c_void_p]),
    "lavie_unet_forward_labels": (c_int, [c_void_p, c_void_p, c_float_p, c_void_p, C.POINTER(c_int), c_void_p, c_int, c_int,
                                           c_int, c_int, c_int, c_void_p]),
    "lavie_unet_resnet_forward": (c_int, [c_void_p, c_char_p, c_void_p, c_int, c_void_p, c_int, c_float_p, c_void_p, c_int,
                                           c_int, c_int, c_int, c_void_p]),
    "lavie_unet_transformer_forward": (c_int, [c_void_p, c_char_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                                c_void_p]),
}

_lib = None


def load():
    """Returns the bound CDLL; raises RuntimeError if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C lavie_amd/csrc).  lavie_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so and the library links the same SONAME.
    # Whichever copy is loaded first serves both; if this library came first (its /opt/rocm copy) and torch later
    # loaded its own, the two runtimes would not see each other's device (hipMalloc: "no ROCm-capable device").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)      # AttributeError here = header/library mismatch
        except AttributeError:
            if os.environ.get("LAVIE_HIP_LIB"):
                continue                  # A/B run against an older build: tolerate missing debug symbols
            raise
        fn.restype = res
        fn.argtypes = args
    if lib.lavie_abi_version() != ABI_VERSION and not os.environ.get("LAVIE_HIP_LIB"):
        raise RuntimeError(f"liblavie_hip.so ABI {lib.lavie_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str = "lavie call"):
    if rc != 0:
        msg = load().lavie_last_error()
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else 'unknown error'}")
