"""Golden-vector case recipes: inputs are regenerated from seeds (nerf_loc_amd.synth), only the
reference's OUTPUTS are stored in tests/golden/<case>.npz (made by tools/gen_golden.py, which imports
the reference in the build container).  Edge cases follow SURVEY.md §8(c)."""
from __future__ import annotations

import numpy as np

from nerf_loc_amd.synth import SceneConfig, add_setup_inputs, make_depth_fusion_weights, make_frame, make_rays, make_u, make_weights

TINY = SceneConfig("tiny", R=16, S=16, W=32, V=3, H=32, Wimg=40, seed=11)

CASES = {
    # name: (config, store_intermediates)
    "c1": (SceneConfig("c1", R=256, S=32, W=64, V=5, H=64, Wimg=80, seed=1), False),
    "tiny_full": (TINY, True),
    "white": (TINY.replace(name="white", white_bkgd=True, seed=12), True),
    "fewpts": (TINY.replace(name="fewpts", seed=13), True),      # M=5 < K=8 support points
    "offview": (TINY.replace(name="offview", V=4, seed=14), True),  # views behind / outside, zero visibility
    "ties": (TINY.replace(name="ties", seed=15), True),          # duplicated support points -> exact KNN ties
    "hier": (TINY.replace(name="hier", S=16, N_importance=16, seed=16), True),
    "w128s64": (SceneConfig("w128s64", R=32, S=64, W=128, V=10, H=64, Wimg=80, seed=17), False),
    "w256s128": (SceneConfig("w256s128", R=24, S=128, W=256, V=10, H=64, Wimg=80, seed=18), False),
    # BASELINE config 4's shape class: 192 samples per ray, outdoor depth range (posenc arguments up to 2^9 x 25 / 24.75), 4:7 views
    "s192out": (SceneConfig("s192out", R=12, S=192, W=256, V=10, H=64, Wimg=112, near=0.25, far=25.0, seed=19), False),
    # BASELINE config 5's shape class: hierarchical 64 coarse + 64 + 128 resampled = 192 samples, 16 views
    "hier192": (SceneConfig("hier192", R=12, S=64, N_importance=128, W=256, V=16, H=64, Wimg=64, seed=20), False),
    # render.lindisp = True (model.py:451-458: depths linear in disparity), plain and hierarchical (round 4)
    "lindisp": (TINY.replace(name="lindisp", lindisp=True, seed=23), True),
    "hier_lindisp": (TINY.replace(name="hier_lindisp", S=16, N_importance=16, lindisp=True, seed=24), True),
}


def build_case(name: str):
    """-> dict(cfg, frame, rays, weights, u) of numpy arrays for golden case `name`."""
    cfg, _ = CASES[name]
    frame = make_frame(cfg)
    if name == "fewpts":
        frame["support_fine"] = {k: np.ascontiguousarray(v[:5]) for k, v in frame["support_fine"].items()}
    if name == "ties":
        sp = frame["support_fine"]
        m = sp["xyz"].shape[0]
        src = np.arange(0, m, 3)
        dst = (src + 1) % m
        sp["xyz"][dst] = sp["xyz"][src]          # exact duplicates with different features
    if name == "offview":
        poses = frame["topk_poses"]
        flip = np.diag([-1.0, 1.0, -1.0]).astype(np.float32)
        poses[1, :3, :3] = poses[1, :3, :3] @ flip      # view 1 looks backwards: samples behind camera
        poses[2, :3, 3] += np.array([30.0, 0, 0], np.float32)   # view 2 far to the side: out of image
        poses[3, :3, 3] += np.array([0, 0, 2.0], np.float32)    # view 3 inside the volume: mixed signs
    rays = make_rays(cfg, frame)
    if name == "offview":
        # half of the rays start far outside every frustum -> <=8 valid samples -> mask False
        rays["rays_o"][::2] += np.array([0, 50.0, 0], np.float32)
    return {"cfg": cfg, "frame": frame, "rays": rays, "weights": make_weights(cfg), "u": make_u(cfg)}


# Per-frame setup cases (row a21): the whole `data` dict the reference's pose estimator hands over, DepthFusionNet weights included.
SETUP_CASES = {
    "setup": SceneConfig("setup", R=24, S=16, W=32, V=3, H=32, Wimg=48, seed=21),
    # ragged support depth: ~35 % of the pixels invalid (0), a few negative, one view with no valid depth at all —
    # nonzero()'s order and the empty-view path of backproject_support_frame (model.py:231)
    "setup_holes": SceneConfig("setup_holes", R=24, S=16, W=32, V=4, H=48, Wimg=64, seed=22),
}


def build_setup_case(name: str):
    """-> dict(cfg, frame, rays, weights) for per-frame setup case `name`."""
    cfg = SETUP_CASES[name]
    frame = add_setup_inputs(cfg, make_frame(cfg))
    if name == "setup_holes":
        rng = np.random.default_rng(99)
        d = frame["topk_depths"].copy()
        d[rng.random(d.shape) < 0.35] = 0.0
        d[rng.random(d.shape) < 0.02] = -1.0
        d[2] = 0.0
        frame["topk_depths"] = d
    weights = dict(make_weights(cfg))
    weights.update(make_depth_fusion_weights(cfg.seed))
    return {"cfg": cfg, "frame": frame, "rays": make_rays(cfg, frame), "weights": weights}
