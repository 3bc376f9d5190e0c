"""ctypes binding of libocm_vit.so (the C ABI declared in include/ocm_vit.h).

The library is the product: there is no Python/torch fallback. If the shared object
is missing or fails to load, importing the compute path raises immediately.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OCM_VIT_LIB lets kernel experiments A/B two builds of the same ABI; the default is the in-tree build.
LIB_PATH = os.environ.get("OCM_VIT_LIB") or os.path.join(_HERE, "libocm_vit.so")

OCM_ABI_VERSION = 9
OCM_OK, OCM_EINVAL, OCM_ESTATE, OCM_EHIP, OCM_ENOMEM, OCM_ENAME = 0, 1, 2, 3, 4, 5

OCM_PREC_BF16 = 0
OCM_PREC_FP32 = 1
OCM_PREC_BF16X3 = 2
OCM_SWIN_OPT_FUSE_MLP = 0
PRECISIONS = {"bf16": OCM_PREC_BF16, "fp32": OCM_PREC_FP32, "bf16x3": OCM_PREC_BF16X3}
DEFAULT_PRECISION = "bf16x3"  # holds the north star's 1e-3 on every golden weight set with an unsaturated softmax
OCM_LN_F32, OCM_LN_BF16, OCM_LN_SPLIT = 0, 1, 2
OCM_OPT_FUSE_LN = 0  # ocm_vit_set_option: 0 auto, 1 never, 2 always
OCM_OPT_FOLD_LN = 1  # 0 auto (on for split-bf16 engines), 1 never

OCM_OUT_FEAT = 1 << 0
OCM_OUT_ATTN = 1 << 1
OCM_OUT_QKV = 1 << 2
OCM_OUT_TOKENS = 1 << 3
OCM_OUT_ROWS = 1 << 4
OCM_LAST_ATTN_ONLY = 1 << 5
OCM_OUT_FMAP = 1 << 6
OCM_USE_GRAPH = 1 << 7

KERNEL_CLASSES = ("patch_embed", "layernorm", "qkv_gemm", "attention", "attn_probs", "proj_gemm", "fc1_gemm",
                  "fc2_gemm")

OCM_EPI_BIAS_F32 = 0
OCM_EPI_BIAS_RESID_F32 = 1
OCM_EPI_BIAS_GELU_BF16 = 2
OCM_EPI_BIAS_BF16 = 3


class OcmVitConfig(C.Structure):
    _fields_ = [
        ("patch_size", C.c_int32),
        ("in_chans", C.c_int32),
        ("embed_dim", C.c_int32),
        ("depth", C.c_int32),
        ("num_heads", C.c_int32),
        ("mlp_hidden", C.c_int32),
        ("ln_eps", C.c_float),
        ("qk_scale", C.c_float),
        ("precision", C.c_int32),
        ("reserved", C.c_int32),
    ]


class OcmVitIO(C.Structure):
    _fields_ = [
        ("image", C.c_void_p),
        ("img_stride_b", C.c_int64),
        ("img_stride_c", C.c_int64),
        ("img_stride_y", C.c_int64),
        ("tile_origins", C.c_void_p),
        ("batch", C.c_int32),
        ("tile_h", C.c_int32),
        ("tile_w", C.c_int32),
        ("pos_embed", C.c_void_p),
        ("flags", C.c_int32),
        ("n_last", C.c_int32),
        ("out_feat", C.c_void_p),
        ("out_attn", C.c_void_p),
        ("out_qkv", C.c_void_p),
        ("out_tokens", C.c_void_p),
        ("query_rows", C.c_void_p),
        ("n_rows", C.c_int32),
        ("reserved", C.c_int32),
        ("out_rows", C.c_void_p),
        ("patch_mask", C.c_void_p),
        ("out_fmap", C.c_void_p),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_size_t),
        ("stream", C.c_void_p),
    ]


class OcmSwinConfig(C.Structure):  # include/ocm_swin.h
    _fields_ = [
        ("image_size", C.c_int32),
        ("patch_size", C.c_int32),
        ("num_channels", C.c_int32),
        ("embed_dim", C.c_int32),
        ("num_stages", C.c_int32),
        ("depths", C.c_int32 * 4),
        ("num_heads", C.c_int32 * 4),
        ("window_size", C.c_int32),
        ("num_labels", C.c_int32),
        ("mlp_ratio", C.c_float),
        ("ln_eps", C.c_float),
        ("precision", C.c_int32),
        ("reserved", C.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/ocm_vit.h and include/ocm_swin.h declare
_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
SIGNATURES = {
    "ocm_abi_version": (C.c_int, []),
    "ocm_last_error": (C.c_char_p, []),
    "ocm_vit_create": (C.c_int, [C.POINTER(OcmVitConfig), C.POINTER(_vp)]),
    "ocm_vit_destroy": (None, [_vp]),
    "ocm_vit_set_param": (C.c_int, [_vp, C.c_char_p, _vp, _sz, _vp]),
    "ocm_vit_params_ready": (C.c_int, [_vp]),
    "ocm_vit_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "ocm_vit_forward": (C.c_int, [_vp, C.POINTER(OcmVitIO)]),
    "ocm_vit_prepare_tokens": (C.c_int, [_vp, C.POINTER(OcmVitIO), _vp]),
    "ocm_vit_block_forward": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "ocm_vit_final_norm": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "ocm_op_layernorm": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _f32, _vp]),
    "ocm_op_cast_bf16": (C.c_int, [_vp, _vp, _sz, _vp]),
    "ocm_op_cast_split": (C.c_int, [_vp, _vp, _sz, _vp]),
    "ocm_op_merge_split": (C.c_int, [_vp, _vp, _sz, _vp]),
    "ocm_n_pad_prec": (_i32, [_i32, _i32]),
    "ocm_op_linear": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "ocm_linear_resid_ln_supported": (C.c_int, [_i32]),
    "ocm_op_linear_resid_ln": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "ocm_n_pad": (_i32, [_i32]),
    "ocm_op_qkv_proj": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "ocm_op_attention": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "ocm_op_qkv_proj_hd": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "ocm_op_attention_hd": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "ocm_op_attention_probs_hd": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "ocm_op_attention_generic": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_float, C.c_void_p]),
    "ocm_op_attention_probs": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "ocm_op_attention_rows": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _f32, _vp]),
    "ocm_op_attention_map": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ocm_op_tile_postprocess": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "ocm_op_bilinear_upsample": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "ocm_op_nearest_upsample": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "ocm_op_stitch": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "ocm_op_normalize_u8": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "ocm_otsu_threshold": (_i32, [C.POINTER(C.c_uint64), _i64]),
    "ocm_op_threshold_u8": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "ocm_vit_graph_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ocm_swin_create": (C.c_int, [C.POINTER(OcmSwinConfig), C.POINTER(C.c_void_p)]),
    "ocm_swin_destroy": (None, [C.c_void_p]),
    "ocm_swin_set_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ocm_swin_params_ready": (C.c_int, [C.c_void_p]),
    "ocm_swin_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32]),
    "ocm_swin_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_size_t, C.c_void_p]),
    "ocm_swin_set_option": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "ocm_op_swin_lnqkv": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "ocm_op_swin_mlp": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "ocm_op_swin_attn_block": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "ocm_op_swin_window_attention": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                               C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                               C.c_int32, C.c_void_p]),
    "ocm_op_pixel_shuffle": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ocm_op_stitch_image_u8": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "ocm_op_weighted_u8": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ocm_op_histogram_u8": (C.c_int, [_vp, _i64, _vp, _vp]),
    "ocm_op_median_filter": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "ocm_op_downscale_centre": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "ocm_op_im2col3x3": (C.c_int, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ocm_op_head_mean": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "ocm_op_image_to_gray_u8": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _vp, _vp]),
    "ocm_op_blend_u8": (C.c_int, [_vp, _vp, _i64, C.c_double, C.c_double, _vp, _vp, _vp]),
    "ocm_vit_set_option": (C.c_int, [C.c_void_p, _i32, _i32]),
    "ocm_prof_begin": (C.c_int, [C.c_uint32, _i32]),
    "ocm_prof_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(_i64)]),
    "ocm_sw_count": (_i32, [_i32, _i32]),
    "ocm_sw_origins": (_i32, [_i32, _i32, _i32, C.POINTER(_i32), _i32]),
    "ocm_sw_shard": (_i32, [_i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
}

_lib = None


class OcmError(RuntimeError):
    """HIP / engine-state failure reported by libocm_vit (OCM_EHIP, OCM_ESTATE, OCM_ENOMEM)."""


def load():
    """Load libocm_vit.so once and bind every declared symbol. Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OcmError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
            f"{os.path.join(_HERE, 'csrc')}`). There is no CPU/torch fallback for this path.")
    # torch must own the HIP runtime instance: import it first so that this library's
    # NEEDED libamdhip64.so.7 resolves to the copy torch already mapped (same streams/pointers).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL if hasattr(C, "RTLD_GLOBAL") else 0)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.ocm_abi_version() != OCM_ABI_VERSION:
        raise OcmError(f"libocm_vit ABI {lib.ocm_abi_version()} != binding ABI {OCM_ABI_VERSION}")
    # development build only (include/ocm_vit_dev.h, loaded through OCM_VIT_LIB): the kernel-variant knobs; OCM_KNOBS="0=4,3=-1"
    # presets them for A/B runs. The product library exports neither.
    if hasattr(lib, "ocm_debug_knob"):
        lib.ocm_debug_knob.restype, lib.ocm_debug_knob.argtypes = C.c_int, [_i32, _i32]
        for item in filter(None, os.environ.get("OCM_KNOBS", "").split(",")):
            which, value = item.split("=")
            if lib.ocm_debug_knob(int(which), int(value)) != OCM_OK:
                raise OcmError(f"OCM_KNOBS: bad entry {item!r}")
    elif os.environ.get("OCM_KNOBS"):
        raise OcmError("OCM_KNOBS is set but the loaded library is the product build (no ocm_debug_knob): "
                       "`make -C csrc dev` and point OCM_VIT_LIB at exp_libs/libocm_vit_dev.so")
    _lib = lib
    return lib


def check(rc):
    """Translate an OCM_E* return code into the Python exception the reference's callers expect."""
    if rc == OCM_OK:
        return
    msg = load().ocm_last_error().decode("utf-8", "replace")
    if rc == OCM_EINVAL:
        raise ValueError(msg)
    if rc == OCM_ENAME:
        raise KeyError(msg)
    raise OcmError(f"libocm_vit error {rc}: {msg}")
