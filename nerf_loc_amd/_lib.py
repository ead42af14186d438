"""ctypes binding of libnerfloc_render.so (C-ABI in include/nerfloc_render.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent this module
raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch  # noqa: F401  -- MUST precede CDLL: torch bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1 (same
# SONAMEs as /opt/rocm's); loading ours first would put two HIP runtimes in the process and every torch pointer
# would be foreign to our kernels.  Importing torch first makes the dynamic loader resolve our DT_NEEDED to its copy.

_HERE = os.path.dirname(os.path.abspath(__file__))
# NERFLOC_LIB: developer override used for same-box A/B timing of two builds of the library
LIB_PATH = os.environ.get("NERFLOC_LIB") or os.path.join(_HERE, "csrc", "libnerfloc_render.so")

NL_OK = 0
NL_ERR_BAD_ARG, NL_ERR_UNSUPPORTED, NL_ERR_WORKSPACE, NL_ERR_HIP, NL_ERR_NO_DEVICE = -1, -2, -3, -4, -5
PREC_F32, PREC_BF16X3, PREC_BF16, PREC_F16MX = 0, 1, 2, 3
# "f16mx" (round 4): BF16X3 everywhere except the fused neural-point kernel of render_rays, which multiplies as fp16 hi.hi + two MX cross terms (FP6 elements since round 5, FP8 in round 4)
PRECISIONS = {"fp32": PREC_F32, "f32": PREC_F32, "bf16x3": PREC_BF16X3, "bf16": PREC_BF16, "f16mx": PREC_F16MX}
MAX_VIEWS = 16
RENDER_NO_SIDE_STREAM = 1   # nl_render_opts.flags
RENDER_PRECISION_GUARD = 2  # nl_render_opts.flags (ABI 7): the library checks the conditioning indicator after the batch and re-renders it in a more exact mode if needed
GUARD_LOGIT_LIMIT = {"f16mx": 50.0, "bf16x3": 500.0}   # include/nerfloc_render.h: NL_GUARD_LOGIT_LIMIT_* (tests/test_abi_symbols.py holds the two in step)
PRECISION_NAMES = {PREC_F32: "fp32", PREC_BF16X3: "bf16x3", PREC_BF16: "bf16", PREC_F16MX: "f16mx"}
DIAG_COUNT = 5
ABI_VERSION = 7   # include/nerfloc_render.h: NL_ABI_VERSION


class NlConfig(C.Structure):
    _fields_ = [("W", C.c_int32), ("C", C.c_int32), ("S", C.c_int32), ("precision", C.c_int32)]


class NlFrameDesc(C.Structure):
    _fields_ = [
        ("V", C.c_int32), ("H", C.c_int32), ("Wimg", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
        ("vis_h", C.c_int32), ("vis_w", C.c_int32),
        ("near_", C.c_float), ("far_", C.c_float),
        ("images", C.c_void_p), ("featmaps", C.c_void_p), ("vis_featmaps", C.c_void_p),
        ("proj_ibr", C.c_void_p), ("proj_neuray", C.c_void_p), ("cam_centers", C.c_void_p),
        ("M", C.c_int64),
        ("sp_xyz", C.c_void_p), ("sp_feature", C.c_void_p), ("sp_confidence", C.c_void_p), ("sp_direction", C.c_void_p),
    ]


class NlRenderOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "rgb", "depth", "weights", "mask", "depth_uncertainty", "feat",
        "sigma", "feature_agg", "mv_feature_agg", "geo", "knn_idx", "knn_d2")]


class NlRenderOpts(C.Structure):
    _fields_ = [("early_term_eps", C.c_float), ("flags", C.c_uint32), ("ray_centers", C.c_void_p), ("reserved", C.c_int32 * 4)]


class NlRenderJob(C.Structure):   # include/nerfloc_render.h: nl_render_job
    _fields_ = [("frame", C.c_void_p), ("query_center", C.c_void_p), ("rays_o", C.c_void_p), ("rays_d", C.c_void_p), ("z_vals", C.c_void_p), ("R", C.c_int64),
                ("out", C.POINTER(NlRenderOut)), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t), ("opts", C.POINTER(NlRenderOpts))]


class NlTrainGrads(C.Structure):
    _fields_ = [("weights", C.POINTER(C.c_void_p)), ("support_feature", C.c_void_p), ("feat_maps", C.c_void_p), ("vis_featmaps", C.c_void_p),
                ("blend_feat_maps", C.c_void_p), ("scratch", C.c_void_p), ("scratch_bytes", C.c_size_t), ("reserved", C.c_int32 * 4)]


class NlRenderCotangents(C.Structure):
    _fields_ = [("g_rgb", C.c_void_p), ("g_depth", C.c_void_p), ("g_depth_uncertainty", C.c_void_p), ("g_feat", C.c_void_p), ("g_weights", C.c_void_p),
                ("knn_idx", C.c_void_p), ("knn_d2", C.c_void_p), ("reserved", C.c_void_p * 1)]


class NlBetaHead(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p), ("beta_min", C.c_float), ("beta", C.c_void_p), ("g_beta", C.c_void_p), ("g_weight", C.c_void_p),
                ("g_bias", C.c_void_p)]


# every symbol include/nerfloc_render.h declares: (name, restype, argtypes)
_P, _I, _L, _Z, _F = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float
_CFG, _DESC, _OUT = C.POINTER(NlConfig), C.POINTER(NlFrameDesc), C.POINTER(NlRenderOut)
SYMBOLS = [
    ("nl_abi_version", _I, []),
    ("nl_profile_begin", _I, []),
    ("nl_profile_end", _I, [C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    ("nl_strerror", C.c_char_p, [_I]),
    ("nl_num_weights", _I, []),
    ("nl_weight_name", C.c_char_p, [_I]),
    ("nl_packed_weights_bytes", _Z, [_CFG]),
    ("nl_pack_weights", _I, [_CFG, C.POINTER(_P), _I, _P, _Z, _P]),
    ("nl_frame_bytes", _Z, [_CFG, _DESC]),
    ("nl_frame_create", _I, [_CFG, _DESC, _P, _Z, _P, C.POINTER(_P)]),
    ("nl_frame_destroy", _I, [_P]),
    ("nl_frame_diagnostics", _I, [_P, C.POINTER(C.c_float), C.c_int32, _P]),
    ("nl_knn", _I, [_P, _P, _L, _I, _P, _P, _P]),
    ("nl_sample_points", _I, [_P, _P, _L, _I, _F, _F, _P, _P, _P, _P]),
    ("nl_mv_aggregate_workspace_bytes", _Z, [_CFG, _I, _L]),
    ("nl_mv_aggregate", _I, [_CFG, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    ("nl_point_mlp_workspace_bytes", _Z, [_CFG, _L]),
    ("nl_point_mlp", _I, [_CFG, _P, _P, _P, _P, _L, _P, _L, _I, _P, _P, _P, _P, _Z, _P]),
    ("nl_ray_unet_workspace_bytes", _Z, [_CFG, _L]),
    ("nl_ray_unet", _I, [_CFG, _P, _P, _L, _P, _P, _Z, _P]),
    ("nl_heads_composite_workspace_bytes", _Z, [_CFG, _I, _L]),
    ("nl_heads_composite", _I, [_CFG, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _OUT, _P, _Z, _P]),
    ("nl_coarse_weights_workspace_bytes", _Z, [_I, _L, _I]),
    ("nl_coarse_weights", _I, [_CFG, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P, _Z, _P]),
    ("nl_sample_pdf", _I, [_P, _P, _I, _P, _I, _P, _I, _L, _P, _P]),
    ("nl_render_rays_workspace_bytes", _Z, [_CFG, _I, _L]),
    ("nl_render_rays_min_workspace_bytes", _Z, [_CFG, _I]),
    ("nl_render_rays", _I, [_CFG, _P, _P, _P, _P, _P, _P, _L, _I, _OUT, _P, _Z, _P]),
    ("nl_render_rays_ex", _I, [_CFG, _P, _P, _P, _P, _P, _P, _L, _I, _OUT, _P, _Z, _P, C.POINTER(NlRenderOpts)]),
    ("nl_render_rays_multi", _I, [_CFG, _P, C.POINTER(NlRenderJob), _I, _I, _P]),
    ("nl_setup_workspace_bytes", _Z, [_I, _I, _I, _I]),
    ("nl_cross_view_features", _I, [_P, _P, _P, _P, _I, _I, _I, _F, _F, _P, _P, _Z, _P]),
    ("nl_get_rays", _I, [_P, _P, _P, _I, _I, _L, _P, _P, _P]),
    ("nl_composite_backward", _I, [_P, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("nl_point_mlp_backward_workspace_bytes", _Z, [_CFG, _L]),
    ("nl_point_mlp_backward", _I, [_CFG, _P, _P, _P, _P, _L, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    ("nl_debug_bump_gap", _I, [_Z]),
    ("nl_debug_check_gaps", _I, [_I, _P, C.POINTER(C.c_int), C.POINTER(C.c_int), _P]),
    ("nl_train_scratch_bytes", _Z, [_CFG]),
    ("nl_render_rays_keep_workspace_bytes", _Z, [_CFG, _I, _L, _I]),
    ("nl_render_rays_forward_keep", _I, [_CFG, _P, _P, _P, _P, _P, _P, _P, _L, _I, _OUT, C.POINTER(NlBetaHead), _I, _P, _Z, _P]),
    ("nl_render_rays_backward_kept", _I, [_CFG, _P, _P, _P, _P, _P, _L, _I, C.POINTER(NlRenderCotangents), C.POINTER(NlBetaHead), _P, _P, _P,
                                          C.POINTER(NlTrainGrads), _P, _Z, _P]),
    ("nl_render_rays_backward_workspace_bytes", _Z, [_CFG, _I, _L, _I]),
    ("nl_render_rays_backward", _I, [_CFG, _P, _P, _P, _P, _P, _P, _P, _L, _I, C.POINTER(NlRenderCotangents), _P, _P, _P, C.POINTER(NlTrainGrads), _P, _Z, _P]),
    ("nl_ray_unet_backward_train_workspace_bytes", _Z, [_CFG, _L]),
    ("nl_ray_unet_backward_train", _I, [_CFG, _P, _P, _L, _P, _P, C.POINTER(NlTrainGrads), _P, _Z, _P]),
    ("nl_mv_aggregate_backward_train_workspace_bytes", _Z, [_CFG, _I, _L]),
    ("nl_mv_aggregate_backward_train", _I, [_CFG, _P, _P, _P, _L, _P, _P, C.POINTER(NlTrainGrads), _P, _Z, _P]),
    ("nl_blend_backward_train_workspace_bytes", _Z, [_CFG, _I, _L]),
    ("nl_blend_backward_train", _I, [_CFG, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, C.POINTER(NlTrainGrads), _P, _Z, _P]),
    ("nl_point_mlp_backward_train_workspace_bytes", _Z, [_CFG, _L]),
    ("nl_point_mlp_backward_train", _I, [_CFG, _P, _P, _P, _P, _L, _P, _L, _I, _P, _P, _P, _P, _P, _P, C.POINTER(NlTrainGrads), _P, _Z, _P]),
    ("nl_mv_aggregate_backward_workspace_bytes", _Z, [_CFG, _I, _L]),
    ("nl_mv_aggregate_backward", _I, [_CFG, _P, _P, _P, _L, _P, _P, _P, _Z, _P]),
    ("nl_blend_workspace_bytes", _Z, [_CFG, _I, _L]),
    ("nl_blend", _I, [_CFG, _P, _P, _P, _P, _P, _L, _P, _P, _Z, _P]),
    ("nl_blend_backward", _I, [_CFG, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _Z, _P]),
    ("nl_ray_unet_backward_workspace_bytes", _Z, [_CFG, _L]),
    ("nl_ray_unet_backward", _I, [_CFG, _P, _P, _L, _P, _P, _P, _Z, _P]),
    ("nl_knn_backward", _I, [_P, _P, _P, _P, _L, _I, _L, _P, _P, _P]),
    ("nl_backproject_support", _I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _P, _P, _P, _P, C.POINTER(_L), _P, _Z, _P]),
]

_lib = None


def build(verbose: bool = False) -> str:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libnerfloc_render.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    if verbose:
        print(res.stdout[-2000:])
    return LIB_PATH


def load() -> C.CDLL:
    """Load the library and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the renderer)")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.nl_abi_version() != ABI_VERSION:
        raise RuntimeError("libnerfloc_render.so ABI version mismatch")
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != NL_OK:
        msg = load().nl_strerror(status).decode()
        raise RuntimeError(f"libnerfloc_render {what}: {msg} ({status})")


def weight_names():
    lib = load()
    return [lib.nl_weight_name(i).decode() for i in range(lib.nl_num_weights())]
