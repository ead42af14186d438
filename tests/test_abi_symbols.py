"""CPU-side checks of the drop-in boundary: the C-ABI library is built in-tree, loads, and exports every
symbol include/nerfloc_render.h declares; the product path refuses to run without a GPU (no CPU fallback)."""
import os
import re

import pytest
import torch

from nerf_loc_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "nerfloc_render.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nl_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = _lib.load()
    declared = _declared()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert sorted(n for n, _, _ in _lib.SYMBOLS) == declared, "ctypes table and header disagree"


def test_dynamic_symbol_table_holds_nothing_but_the_declared_abi():
    """`nm -D`: the shared library exports exactly the nl_* entry points of the header — no kernel host stubs (round 5 leaked `table_add_t_kernel`),
    no weak template instantiations, no per-object hipcc markers (csrc/exports.map)."""
    import subprocess
    res = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True)
    names = sorted(line.split()[-1] for line in res.stdout.splitlines() if line.strip())
    assert names == _declared(), sorted(set(names) ^ set(_declared()))


def test_guard_limits_of_the_binding_equal_the_header():
    """NL_RENDER_PRECISION_GUARD (ABI 7): flag value and the two |logit| limits the library escalates at are the ones the Python side documents and tests against."""
    hdr = open(os.path.join(ROOT, "include", "nerfloc_render.h")).read()
    assert re.search(r"#define\s+NL_RENDER_PRECISION_GUARD\s+(\d+)u", hdr).group(1) == str(_lib.RENDER_PRECISION_GUARD)
    assert float(re.search(r"#define\s+NL_GUARD_LOGIT_LIMIT_F16MX\s+([0-9.]+)f", hdr).group(1)) == _lib.GUARD_LOGIT_LIMIT["f16mx"]
    assert float(re.search(r"#define\s+NL_GUARD_LOGIT_LIMIT_BF16X3\s+([0-9.]+)f", hdr).group(1)) == _lib.GUARD_LOGIT_LIMIT["bf16x3"]
    assert int(re.search(r"#define\s+NL_DIAG_COUNT\s+(\d+)", hdr).group(1)) == _lib.DIAG_COUNT
    flags_all = int(re.search(r"#define\s+NL_RENDER_FLAGS_ALL\s+(\d+)u", hdr).group(1))
    assert flags_all == (_lib.RENDER_NO_SIDE_STREAM | _lib.RENDER_PRECISION_GUARD)


def test_weight_table_matches_state_dict_contract():
    from nerf_loc_amd.synth import CONFIGS, weight_shapes
    names = _lib.weight_names()
    shapes = weight_shapes(CONFIGS["c1"])
    assert len(names) == 84 and len(set(names)) == 84
    for n in names:
        assert n in shapes, n


def test_error_codes_and_bad_args_do_not_crash():
    lib = _lib.load()
    assert lib.nl_strerror(0) == b"ok"
    assert lib.nl_strerror(-3) == b"workspace too small"
    bad = _lib.NlConfig(100, 192, 30, 0)   # W not multiple of 32, S not multiple of 8
    import ctypes as ct
    assert lib.nl_packed_weights_bytes(ct.byref(bad)) == 0
    ok = _lib.NlConfig(256, 192, 128, 1)
    assert lib.nl_packed_weights_bytes(ct.byref(ok)) > 1_000_000
    assert lib.nl_render_rays_min_workspace_bytes(ct.byref(ok), 10) > 0
    assert lib.nl_render_rays(ct.byref(ok), None, None, None, None, None, None, 4, 0, None, None, 0, None) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a GPU-less host")
def test_product_path_fails_loudly_without_gpu():
    from nerf_loc_amd.renderer import HipRenderer
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HipRenderer(64, 192, 32)


def test_render_opts_struct_matches_the_header():
    """nl_render_opts is passed by pointer across the C-ABI: 32 bytes, the per-ray centre pointer at offset 8 (include/nerfloc_render.h)."""
    import ctypes
    from nerf_loc_amd import _lib as L
    assert ctypes.sizeof(L.NlRenderOpts) == 32
    assert L.NlRenderOpts.early_term_eps.offset == 0 and L.NlRenderOpts.ray_centers.offset == 8
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "include", "nerfloc_render.h")).read()
    assert L.NlRenderOpts.flags.offset == 4
    assert f"#define NL_ABI_VERSION {L.ABI_VERSION}" in hdr and "const float* ray_centers;" in hdr and "uint32_t flags;" in hdr


def test_render_opts_validation_rejects_what_a_later_abi_could_define():
    """nl_render_opts: unknown flag bits, non-zero reserved words and an early_term_eps outside [0, 1) (NaN included) are refused before
    anything else is looked at — callers with uninitialised fields fail today instead of changing behaviour under a later ABI."""
    import ctypes as ct
    lib = _lib.load()
    cfg = _lib.NlConfig(256, 192, 128, 1)
    out = _lib.NlRenderOut()

    def call(o):   # (no GPU here: the other arguments are null, which is refused too — the GPU suite repeats this with a live frame,
        # tests/test_gpu_configs.py::test_render_opts_are_validated_and_streams_do_not_interfere)
        return lib.nl_render_rays_ex(ct.byref(cfg), None, None, None, None, None, None, 4, 0, ct.byref(out), None, 0, None, ct.byref(o))
    for bad in (dict(flags=4), dict(flags=0x80000000), dict(early_term_eps=float("nan")), dict(early_term_eps=1.0), dict(early_term_eps=-0.1)):
        o = _lib.NlRenderOpts()
        for k, v in bad.items():
            setattr(o, k, v)
        assert call(o) == _lib.NL_ERR_BAD_ARG, bad
    o = _lib.NlRenderOpts()
    o.reserved[2] = 7
    assert call(o) == _lib.NL_ERR_BAD_ARG
    assert lib.nl_frame_destroy(None) == _lib.NL_OK


def test_build_post_check_agrees_with_the_header():
    """__graft_entry__.build() ends with this check (round 2 shipped a stale hard-coded version there)."""
    import __graft_entry__ as g
    g.post_build_check()


def test_train_grads_are_validated_before_anything_is_dereferenced():
    """nl_train_grads (training entry points): non-zero reserved words are NL_ERR_BAD_ARG, a missing / short / misaligned split-K scratch is
    NL_ERR_WORKSPACE — checked before the frame or any device pointer is touched (no GPU needed: the pointers below are host buffers that are never read)."""
    import ctypes as ct
    lib = _lib.load()
    cfg = _lib.NlConfig(64, 192, 32, 1)
    buf = (ct.c_char * 4096)()
    p = ct.cast(buf, ct.c_void_p)
    need = lib.nl_train_scratch_bytes(ct.byref(cfg))
    assert need > 0 and lib.nl_train_scratch_bytes(None) == 0

    def unet(g):
        return lib.nl_ray_unet_backward_train(ct.byref(cfg), p, p, 4, p, p, ct.byref(g), p, 4096, None)
    g = _lib.NlTrainGrads()
    g.scratch, g.scratch_bytes = p, need
    g.reserved[1] = 5
    assert unet(g) == _lib.NL_ERR_BAD_ARG
    g = _lib.NlTrainGrads()
    assert unet(g) == _lib.NL_ERR_WORKSPACE                       # no scratch
    g.scratch, g.scratch_bytes = p, need - 1
    assert unet(g) == _lib.NL_ERR_WORKSPACE                       # short
    g.scratch, g.scratch_bytes = ct.c_void_p(p.value + 4), need
    assert unet(g) == _lib.NL_ERR_WORKSPACE                       # misaligned
    # the cotangent block of the whole-path backward: its reserved word and a half-given neighbour pair
    c = _lib.NlRenderCotangents()
    c.reserved[0] = p
    call = lambda cc: lib.nl_render_rays_backward(ct.byref(cfg), p, p, p, None, p, p, p, 4, 0, ct.byref(cc), p, p, None, None, p, 4096, None)
    assert call(c) == _lib.NL_ERR_BAD_ARG
    c = _lib.NlRenderCotangents()
    c.knn_idx = p
    assert call(c) == _lib.NL_ERR_BAD_ARG
    assert ct.sizeof(_lib.NlTrainGrads) == 72 and ct.sizeof(_lib.NlRenderCotangents) == 64
