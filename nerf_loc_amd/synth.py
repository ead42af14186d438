"""Seeded numpy recipes for synthetic frames, rays and weights.

Everything the parity tests, the golden-vector generator (tools/gen_golden.py) and bench.py
feed to the renderer comes from here, so the GPU box regenerates bit-identical inputs from a
seed instead of shipping input tensors.  No torch RNG is used anywhere.

The *shapes and dict keys* follow the reference's `data` / `rays` contract
(nerf_loc/models/nerf_pose_estimator.py:255-290, conditional_nerf/model.py:472-480,687-700);
the per-frame caches that the reference computes in `build_support_neural_points`
(model.py:144-201) are synthesised directly (back-projected synthetic depth maps) and injected.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

F32 = np.float32


@dataclass(frozen=True)
class SceneConfig:
    """Sizes of one synthetic workload. Field meanings mirror SURVEY.md §8 symbols."""
    name: str = "c1"
    R: int = 256            # rays per batch
    S: int = 32             # render.N_samples
    N_importance: int = 0   # render.N_importance (hierarchical adds these to S)
    W: int = 64             # model_3d_hidden_dim
    C: int = 192            # backbone2d_fpn_dim
    V: int = 5              # support views
    H: int = 64             # image height
    Wimg: int = 80          # image width
    near: float = 0.3
    far: float = 5.0
    white_bkgd: bool = False
    seed: int = 1
    lindisp: bool = False   # render.lindisp: base (and coarse) depths linear in disparity (model.py:451-458)

    @property
    def h(self) -> int:
        return self.H // 4

    @property
    def w(self) -> int:
        return self.Wimg // 4

    @property
    def S_total(self) -> int:
        return self.S + self.N_importance

    def replace(self, **kw) -> "SceneConfig":
        return dataclasses.replace(self, **kw)


# BASELINE.json configs (SURVEY.md §8 header).  c2 is the headline bench workload.
CONFIGS: Dict[str, SceneConfig] = {
    "c1": SceneConfig("c1", R=256, S=32, W=64, V=5, H=64, Wimg=80, seed=1),
    "c2": SceneConfig("c2", R=4096, S=128, W=256, V=10, H=256, Wimg=336, seed=2),
    "c3": SceneConfig("c3", R=8192, S=128, W=256, V=10, H=256, Wimg=336, seed=3),
    "c4": SceneConfig("c4", R=16384, S=192, W=256, V=10, H=256, Wimg=448, near=0.25, far=25.0, seed=4),
    "c5": SceneConfig("c5", R=4096, S=64, N_importance=128, W=256, V=16, H=256, Wimg=256, seed=5),
}


def _rodrigues(rvec: np.ndarray) -> np.ndarray:
    th = float(np.linalg.norm(rvec))
    if th < 1e-12:
        return np.eye(3)
    k = rvec / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def _pose(rng: np.random.Generator, trans_sigma: float, rot_sigma_deg: float) -> np.ndarray:
    T = np.eye(4)
    T[:3, :3] = _rodrigues(rng.normal(0.0, np.deg2rad(rot_sigma_deg), 3))
    T[:3, 3] = rng.normal(0.0, trans_sigma, 3)
    return T.astype(F32)


def _ray_dirs(u: np.ndarray, v: np.ndarray, K: np.ndarray, c2w: np.ndarray) -> np.ndarray:
    """Unit world-space ray directions for pixel centres (same math as utils.get_rays, fp32)."""
    dirs = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u)], -1).astype(F32)
    d = (dirs[..., None, :] * c2w[:3, :3]).sum(-1).astype(F32)
    return (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(F32)


def make_frame(cfg: SceneConfig, depth_mode: str = "surface") -> Dict[str, np.ndarray]:
    """Per-frame state: support views + injected caches (support neural points, vis_featmaps).

    Returns numpy fp32 arrays keyed like the reference's `data` dict plus
    `support_fine` (dict xyz/feature/confidence/direction) and `vis_featmaps`.
    """
    rng = np.random.default_rng(cfg.seed)
    V, H, Wi, h, w, C = cfg.V, cfg.H, cfg.Wimg, cfg.h, cfg.w, cfg.C
    f = 0.9 * Wi
    K = np.array([[f, 0, (Wi - 1) / 2.0], [0, f, (H - 1) / 2.0], [0, 0, 1]], dtype=F32)
    Ks = np.repeat(K[None], V, 0).copy()
    poses = np.stack([_pose(rng, 0.05 * (cfg.far - cfg.near) / 4.7, 2.0) for _ in range(V)])
    pose_q = _pose(rng, 0.03 * (cfg.far - cfg.near) / 4.7, 1.5)

    images = rng.random((V, 3, H, Wi), dtype=F32)
    feat_fine = rng.standard_normal((V, h, w, C), dtype=F32)
    vis_featmaps = rng.standard_normal((V, 32, h, w), dtype=F32)

    # depth maps: a smooth surface per view inside (near, far) plus a little noise, so the
    # back-projected neural points form sheets (realistic for KNN) rather than a uniform cloud.
    vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(Wi, dtype=np.float64), indexing="ij")
    span = cfg.far - cfg.near
    depths = np.empty((V, H, Wi), dtype=F32)
    for i in range(V):
        ph = rng.uniform(0, 2 * np.pi, 3)
        base = cfg.near + span * (0.42 + 0.12 * np.sin(2 * np.pi * 1.3 * uu / Wi + ph[0]) * np.cos(2 * np.pi * 0.9 * vv / H + ph[1])
                                  + 0.05 * np.sin(2 * np.pi * 3.1 * (uu + vv) / (Wi + H) + ph[2]))
        if depth_mode == "noise":
            base = cfg.near + span * rng.uniform(0.2, 0.6, (H, Wi))
        depths[i] = (base + 0.004 * span * rng.standard_normal((H, Wi))).astype(F32)

    # support neural points (fine level): stride-4 back-projection of every view
    xyz_l, feat_l, dir_l = [], [], []
    ys = (np.arange(h) * 4)  # nearest-neighbour F.interpolate picks floor(i*scale)
    xs = (np.arange(w) * 4)
    Ks4 = K.copy()
    Ks4[:2] /= 4.0
    gv, gu = np.meshgrid(np.arange(h, dtype=F32), np.arange(w, dtype=F32), indexing="ij")
    for i in range(V):
        z = depths[i][np.ix_(ys, xs)].reshape(-1).astype(F32)
        rgb = images[i][:, ys][:, :, xs].transpose(1, 2, 0).reshape(-1, 3)
        u = gu.reshape(-1)
        v = gv.reshape(-1)
        cam = np.stack([(u - Ks4[0, 2]) / Ks4[0, 0] * z, (v - Ks4[1, 2]) / Ks4[1, 1] * z, z], -1).astype(F32)
        c2w = poses[i]
        world = (cam @ c2w[:3, :3].T + c2w[:3, 3]).astype(F32)
        d = _ray_dirs(u, v, Ks4, c2w)
        xyz_l.append(world)
        feat_l.append(np.concatenate([rgb, feat_fine[i].reshape(-1, C)], 1).astype(F32))
        dir_l.append(np.concatenate([d, z[:, None]], 1).astype(F32))
    xyz = np.concatenate(xyz_l)
    M = xyz.shape[0]
    support = {
        "xyz": np.ascontiguousarray(xyz, dtype=F32),
        "feature": np.ascontiguousarray(np.concatenate(feat_l), dtype=F32),
        "confidence": rng.uniform(0.2, 1.0, (M, 1)).astype(F32),
        "direction": np.ascontiguousarray(np.concatenate(dir_l), dtype=F32),
    }
    return {
        "topk_images": images, "topk_depths": depths, "topk_Ks": Ks, "topk_poses": poses,
        "feat_fine_src": feat_fine, "depth_range": np.array([[cfg.near, cfg.far]], dtype=F32),
        "K": K, "pose": pose_q, "H": H, "W": Wi, "white_bkgd": cfg.white_bkgd,
        "support_fine": support, "vis_featmaps": vis_featmaps,
    }


def make_rays(cfg: SceneConfig, frame: Dict[str, np.ndarray], R: int | None = None, seed_offset: int = 1000) -> Dict[str, np.ndarray]:
    """R uniformly random pixels of the query view -> rays dict (model.py:687-700 contract)."""
    R = cfg.R if R is None else R
    rng = np.random.default_rng(cfg.seed + seed_offset)
    u = rng.integers(0, cfg.Wimg, R).astype(F32)
    v = rng.integers(0, cfg.H, R).astype(F32)
    K, pose = frame["K"], frame["pose"]
    d = _ray_dirs(u, v, K, pose)
    o = np.broadcast_to(pose[:3, 3], d.shape).astype(F32).copy()
    return {
        "rays_o": o, "rays_d": d, "pixel_coordinates": np.stack([u, v], 1).astype(F32),
        "depth_range": frame["depth_range"][0].copy(), "K": K, "pose": pose, "H": cfg.H, "W": cfg.Wimg,
    }


def borderline_rays(cfg: SceneConfig, frame: Dict[str, np.ndarray], rays_o: np.ndarray, rays_d: np.ndarray, z: np.ndarray, tol_px: float = 1e-3) -> np.ndarray:
    """(R,) bool: rays with a sample that projects within `tol_px` pixel of an image border of some support view (or within 1e-4 of its camera plane).  The
    in-image masks of the reference (ibrnet.py:169-199, neuray_ops.py) are hard thresholds on fp32 projections, so such a sample is inside for one summation
    order of the projection and outside for another (seen: y = 60.49999 against the bound 60.5, visibility 0.99 vs 0) and the ray's outputs legitimately differ
    between two correct fp32 evaluations.  Parity comparisons report these rays separately (tools/forward_fuzz.py, bench.py `parity`, the full-batch test)."""
    o, d = np.asarray(rays_o, np.float64), np.asarray(rays_d, np.float64)
    x = o[:, None, :] + d[:, None, :] * np.asarray(z, np.float64)[..., None]
    flag = np.zeros(x.shape[0], bool)
    for v in range(frame["topk_poses"].shape[0]):
        w2c = np.linalg.inv(frame["topk_poses"][v].astype(np.float64))
        pc = x @ w2c[:3, :3].T + w2c[:3, 3]
        uv = pc @ frame["topk_Ks"][v].astype(np.float64)[:3, :3].T
        px, py, pz = uv[..., 0] / uv[..., 2], uv[..., 1] / uv[..., 2], pc[..., 2]
        near = np.abs(pz) < 1e-4
        for val, size in ((px, cfg.Wimg), (py, cfg.H)):
            for b in (-0.5, 0.0, size - 1.0, size - 0.5):
                near |= np.abs(val - b) < tol_px
        flag |= near.any(1)
    return flag


def make_u(cfg: SceneConfig, R: int | None = None) -> np.ndarray:
    """Uniform draws for sample_pdf (the reference hard-wires torch.rand, utils.py:96)."""
    R = cfg.R if R is None else R
    rng = np.random.default_rng(cfg.seed + 2000)
    return rng.random((R, max(cfg.N_importance, 1)), dtype=F32)


def weight_shapes(cfg: SceneConfig) -> Dict[str, Tuple[int, ...]]:
    """state_dict names/shapes of ConditionalNeRF on the ray path.

    The 72 `multiview_aggregator.depth_fusion.*` tensors (per-frame CNN) are not listed here;
    the drop-in module carries them separately (nerf_loc_amd/depth_fusion.py).
    """
    W, C, S = cfg.W, cfg.C, cfg.S_total
    F = C + 3
    sh: Dict[str, Tuple[int, ...]] = {}

    def lin(name, o, i, bias=True):
        sh[f"{name}.weight"] = (o, i)
        if bias:
            sh[f"{name}.bias"] = (o,)

    lin("ray_diff_fc.0", 16, 4)
    lin("ray_diff_fc.2", 27, 16)
    for dec, nout in (("mean", 2), ("var", 2), ("aw", 1), ("vis", 1)):
        p = f"multiview_aggregator.dist_decoder.{dec}_decoder"
        lin(f"{p}.0", 32, 32)
        lin(f"{p}.2", 32, 32)
        lin(f"{p}.4", nout, 32)
    lin("multiview_aggregator.out_fc.0", 64, 2 * F + 3)
    lin("multiview_aggregator.out_fc.2", W, 64)
    lin("confidence_mlp.0", 64, W)
    lin("confidence_mlp.2", 1, 64)
    lin("keypoint_head.0", 1, C)
    lin("base_mlp.0", W, F + 63 + 27)
    lin("base_mlp.2", W, W)
    lin("base_mlp.4", W, W)
    lin("base_mlp_attn.w_qs", 128, W, bias=False)
    lin("base_mlp_attn.w_ks", 128, W, bias=False)
    lin("base_mlp_attn.w_vs", 128, W, bias=False)
    lin("base_mlp_attn.fc", W, 128, bias=False)
    sh["base_mlp_attn.layer_norm.weight"] = (W,)
    sh["base_mlp_attn.layer_norm.bias"] = (W,)
    lin("base_mlp_agg_weight.0", W, W)
    lin("base_mlp_agg_weight.2", 1, W)

    def conv(name, co, ci, ln_c, ln_s, transposed=False):
        sh[f"ray_unet.{name}.0.weight"] = (ci, co, 3) if transposed else (co, ci, 3)
        sh[f"ray_unet.{name}.0.bias"] = (co,)
        sh[f"ray_unet.{name}.1.weight"] = (ln_c, ln_s)
        sh[f"ray_unet.{name}.1.bias"] = (ln_c, ln_s)

    conv("conv1", 64, W, 64, S)
    conv("conv2", 128, 64, 128, S // 2)
    conv("conv3", 128, 128, 128, S // 4)
    conv("trans_conv3", 128, 128, 128, S // 4, transposed=True)
    conv("trans_conv2", 64, 256, 64, S // 2, transposed=True)
    conv("trans_conv1", 32, 128, 32, S, transposed=True)
    conv("conv_out", W, W + 32, W, S)
    lin("sigma_mlp.0", 1, W)
    lin("feat_mlp.0", W, W)
    lin("feat_mlp.2", C, W)
    lin("rgb_blending_mlp.0", 32, W + F + 1 + 4)
    lin("rgb_blending_mlp.2", 16, 32)
    lin("rgb_blending_mlp.4", 1, 16)
    lin("beta_mlp.0", 1, W)
    lin("proj_layer_3d_coarse", 192, W + F)
    lin("proj_layer_3d_fine", 192, W + F)
    return sh


def make_weights(cfg: SceneConfig, seed: int | None = None) -> Dict[str, np.ndarray]:
    """N(0, gain/fan_in) weights, small biases, LayerNorm affine near (1, 0).

    Drawn tensor-by-tensor in sorted-name order from default_rng so the recipe is order-stable.
    """
    rng = np.random.default_rng((cfg.seed if seed is None else seed) + 7919)
    out: Dict[str, np.ndarray] = {}
    shapes = weight_shapes(cfg)
    for name in sorted(shapes):
        shp = shapes[name]
        is_ln = (".1.weight" in name or ".1.bias" in name or "layer_norm" in name)
        if is_ln and name.endswith("weight"):
            a = 1.0 + 0.1 * rng.standard_normal(shp)
        elif is_ln:
            a = 0.1 * rng.standard_normal(shp)
        elif name.endswith("bias"):
            a = 0.1 * rng.standard_normal(shp)
        else:
            if len(shp) == 3:  # conv: (co, ci, 3) or transposed (ci, co, 3)
                fan_in = shp[1] * 3 if "trans_conv" not in name else shp[0] * 1.5
            else:
                fan_in = shp[1]
            a = rng.standard_normal(shp) * np.sqrt(2.0 / fan_in)
        out[name] = a.astype(F32)
    return out


def make_coord_desc_weights(cfg: "SceneConfig", seed: int = 0, matcher_dim: int = 192) -> Dict[str, np.ndarray]:
    """Seeded weights of `coord_desc_mlp_{coarse,fine}` (model.py:115-131: 63 -> W -> W -> matcher_hidden_dim, the per-scene fine-tuning heads that
    `use_scene_coord_memorization` adds — every shipped per-scene config turns it on)."""
    rng = np.random.default_rng(seed + 7919)
    out = {}
    for lvl in ("coarse", "fine"):
        for i, (n_out, n_in) in zip((0, 2, 4), ((cfg.W, 63), (cfg.W, cfg.W), (matcher_dim, cfg.W))):
            out[f"coord_desc_mlp_{lvl}.{i}.weight"] = (rng.standard_normal((n_out, n_in)) / np.sqrt(n_in)).astype(F32)
            out[f"coord_desc_mlp_{lvl}.{i}.bias"] = (0.1 * rng.standard_normal(n_out)).astype(F32)
    return out


def make_depth_fusion_weights(seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded weights for the per-frame CNN (`multiview_aggregator.depth_fusion.*`, 72 tensors).  Names/shapes are taken
    from nerf_loc_amd.depth_fusion.DepthFusionNet, whose state_dict equals the reference's (asserted by gen_golden)."""
    from .depth_fusion import DepthFusionNet
    rng = np.random.default_rng(seed + 104729)
    out = {}
    sd = DepthFusionNet().state_dict()
    for name in sorted(sd):
        shp = tuple(sd[name].shape)
        if len(shp) == 4:
            a = rng.standard_normal(shp) * np.sqrt(2.0 / (shp[1] * shp[2] * shp[3]))
        elif name.endswith("weight"):
            a = 1.0 + 0.1 * rng.standard_normal(shp)
        else:
            a = 0.1 * rng.standard_normal(shp)
        out["multiview_aggregator.depth_fusion." + name] = a.astype(F32)
    return out


def add_setup_inputs(cfg: SceneConfig, frame: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Extra `data` keys the per-frame setup reads (model.py:144-165): coarse feature map + strides."""
    rng = np.random.default_rng(cfg.seed + 31337)
    frame = dict(frame)
    frame["feat_coarse_src"] = rng.standard_normal((cfg.V, cfg.H // 8, cfg.Wimg // 8, cfg.C), dtype=F32)
    frame["stride_fine"], frame["stride_coarse"] = 4, 8
    return frame
