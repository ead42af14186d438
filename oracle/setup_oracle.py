"""CPU oracle for the per-frame setup.  TEST INFRASTRUCTURE ONLY — same rules as render_oracle.py:
only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` may import it; the product (nerf_loc_amd/) never does.

fp32 PyTorch-CPU restatement, in the reference's unfused tensor formulation, of
  * `cnn_input`                — what DepthFusionNet feeds its CNN (conditional_nerf/depth_fusion.py:150-227, 269-278);
  * `backproject_support_frame` — conditional_nerf/model.py:203-265.
Pinned on `tests/golden/setup.npz` and `tests/golden/setup_holes.npz`, produced by the imported reference
(`tools/gen_golden.py setup`); `tests/test_oracle_golden.py` checks them.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .render_oracle import get_rays, interpolate_feats, project_points_ref

Tensor = torch.Tensor


def normalised_inverse_depth(depths: Tensor, near: float, far: float) -> Tensor:
    """depth_fusion.py:209-227 (extract_depth_for_init): (V,H,W) metric -> (V,1,H,W) in [0,1], linear in inverse depth."""
    ni, fi = -1.0 / near, -1.0 / far
    d = torch.clamp(depths.unsqueeze(1), min=1e-5)
    return torch.clamp((-1 / d - ni) / (fi - ni), min=0, max=1.0)


def masked_mean_var(x: Tensor, mask: Tensor, dim: int):
    """neuray_ops.py:38-43."""
    s = torch.clamp_min(mask.sum(dim, keepdim=True), 1e-4)
    mean = (x * mask).sum(dim, keepdim=True) / s
    return mean, ((x - mean) ** 2 * mask).sum(dim, keepdim=True) / s


def cross_view_consistency(imgs: Tensor, depth_norm: Tensor, Ks: Tensor, Rt: Tensor, near: float, far: float) -> Tensor:
    """depth_fusion.py:150-207 (depth2pts3d + get_diff_feats) -> (V,8,H,W)."""
    V, _, h, w = imgs.shape
    ni, fi = -1.0 / near, -1.0 / far
    depth = -1 / (depth_norm * (fi - ni) + ni)
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    coords = torch.stack([xs, ys, torch.ones_like(xs)], -1).float()[None]                    # (1,h,w,3) = (x, y, 1)
    pts = (depth.permute(0, 2, 3, 1).unsqueeze(-1) * coords.unsqueeze(-2)).reshape(V, h * w, 3).permute(0, 2, 1)
    pts = torch.inverse(Ks) @ pts
    R = Rt[:, :3, :3].permute(0, 2, 1)
    t = -R @ Rt[:, :3, 3:]
    pts = (R @ pts + t).permute(0, 2, 1).reshape(-1, 3)
    pix, prj_depth, valid = project_points_ref(pts, Rt, Ks, h, w)
    d_int = interpolate_feats(depth, pix, h, w, "border", True)
    c_int = interpolate_feats(imgs, pix, h, w, "border", True)
    rgb_diff = (c_int - imgs.permute(0, 2, 3, 1).reshape(1, V * h * w, 3)).abs()
    d_int = torch.clamp(d_int, min=1e-5)
    prj_depth = torch.clamp(prj_depth, min=1e-5)
    d_diff = torch.clamp((-1 / d_int + 1 / prj_depth).abs() / (fi - ni), max=1.5)
    m = valid.float().unsqueeze(-1)
    dm, dv = masked_mean_var(d_diff, m, 0)
    cm, cv = masked_mean_var(rgb_diff, m, 0)

    def fold(x, c):
        return x.reshape(V, h, w, c).permute(0, 3, 1, 2)
    return torch.cat([fold(cm, 3), fold(cv, 3), fold(dm, 1), fold(dv, 1)], 1)


def cnn_input(imgs: Tensor, depths: Tensor, Ks: Tensor, c2w: Tensor, near: float, far: float) -> Tensor:
    """depth_fusion.py:269-278 — cat([imgs, depth, diff_feats]) -> (V,12,H,W)."""
    d = normalised_inverse_depth(depths, near, far)
    return torch.cat([imgs, d, cross_view_consistency(imgs, d, Ks, c2w.inverse()[:, :3], near, far)], 1)


def backproject_support_frame(imgs: Tensor, feats: Tensor, depths: Tensor, Ks: Tensor, c2ws: Tensor, stride: int):
    """model.py:203-265 -> feature (M,3+C), xyz world (M,3), xyz in view 0's camera (M,3), direction+depth (M,4)."""
    refs, worlds, descs, dirs = [], [], [], []
    w2c_ref = c2ws[0].inverse()
    for img, feat, depth, K, c2w in zip(imgs, feats, depths, Ks, c2ws):
        H, W = int(img.shape[-2] / stride), int(img.shape[-1] / stride)
        K = K.clone()
        K[:2] /= stride
        depth = F.interpolate(depth[None, None], size=(H, W)).squeeze()
        img = F.interpolate(img[None], size=(H, W)).squeeze().permute(1, 2, 0)
        v, u = torch.nonzero(depth > 0, as_tuple=True)
        z = depth[v, u]
        cam = torch.matmul(K.inverse(), torch.stack([u, v, torch.ones_like(u)], 0).float()) * z
        world = torch.matmul(c2w[:3, :3], cam) + c2w[:3, 3:]
        ref = torch.matmul(torch.matmul(w2c_ref, c2w), torch.cat([cam, torch.ones_like(cam[:1])]))[:3]
        _, rd = get_rays(H, W, K, c2w)
        refs.append(ref.T)
        worlds.append(world.T)
        descs.append(torch.cat([img[v, u], feat[v, u]], 1))
        dirs.append(torch.cat([rd[v, u], z.view(-1, 1)], 1))
    return torch.cat(descs), torch.cat(worlds), torch.cat(refs), torch.cat(dirs)
