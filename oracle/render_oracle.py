"""CPU oracle for the NeRF-Loc conditional-NeRF render path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product path (nerf_loc_amd/) never does and fails loudly without its HIP library.

What it is: an fp32 PyTorch-CPU restatement, in the reference's own unfused op formulation, of
    ray sampling -> multi-view (NeuRay-style) aggregation -> K-nearest neural-point MLP +
    attention -> ray U-Net -> heads -> front-to-back alpha compositing
i.e. SURVEY.md §8(a) rows a1-a20.  Every function cites the reference lines it follows
(paths relative to /root/reference/nerf_loc/models/).

Parity pinning: the reference ships no tests/golden vectors for this path, so this
oracle is pinned against outputs of the reference itself, imported in the build container by
`tools/gen_golden.py` (4 sys.modules shims) and committed as `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every fixture.  KNN follows the reference's CPU op
(ops/knn/src/knn_cpu.cpp:13-64); `oracle/knn_oracle.c` is the C restatement used for large
sizes and `oracle/_ref/` is the reference's own knn_cpu.cpp compiled in place (build container).

Parameters are addressed by the reference's state_dict names.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
_HERE = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------- rays / sampling
def get_rays(H: int, W: int, K: Tensor, c2w: Tensor):
    """conditional_nerf/utils.py:56-70 — pixel grid -> (rays_o, rays_d) with unit directions."""
    u = torch.linspace(0, W - 1, W).view(1, W).expand(H, W)
    v = torch.linspace(0, H - 1, H).view(H, 1).expand(H, W)
    cam = torch.stack([(u - K[0][2]) / K[0][0], (v - K[1][2]) / K[1][1], torch.ones_like(u)], -1)
    d = (cam[..., None, :] * c2w[:3, :3]).sum(-1)
    d = d / torch.norm(d, dim=-1, keepdim=True)
    o = c2w[:3, -1].expand(d.shape)
    return o, d


def points_2d_to_rays(pts2d: Tensor, H: int, W: int, K: Tensor, pose: Tensor) -> Dict[str, Tensor]:
    """conditional_nerf/model.py:687-700 — integer-truncated pixel look-up into the ray grid."""
    x, y = pts2d[:, 0].long(), pts2d[:, 1].long()
    o, d = get_rays(H, W, K, pose)
    return {"pose": pose, "K": K, "H": H, "W": W, "pixel_coordinates": pts2d, "rays_o": o[y, x], "rays_d": d[y, x]}


def sample_depths(n: int, near: Tensor, far: Tensor, lindisp: bool = False) -> Tensor:
    """conditional_nerf/model.py:451-458."""
    t = torch.linspace(0, 1, n)
    if lindisp:
        return 1 / (1 / near * (1 - t) + 1 / far * t)
    return near * (1 - t) + far * t


def posenc(x: Tensor, n_freqs: int = 10) -> Tensor:
    """conditional_nerf/utils.py:5-35 — [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(...)]."""
    bands = 2.0 ** torch.linspace(0.0, n_freqs - 1, steps=n_freqs)
    parts = [x]
    for fr in bands:
        parts.append(torch.sin(x * fr))
        parts.append(torch.cos(x * fr))
    return torch.cat(parts, -1)


def sample_pdf(bins: Tensor, weights: Tensor, n_imp: int, u: Tensor, eps: float = 1e-5) -> Tensor:
    """conditional_nerf/utils.py:73-112 with the uniform draws `u` passed in (reference: torch.rand)."""
    n_bins = weights.shape[1]
    w = weights + eps
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    lo = torch.clamp_min(inds - 1, 0)
    hi = torch.clamp_max(inds, n_bins)
    cdf_lo, cdf_hi = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
    b_lo, b_hi = torch.gather(bins, 1, lo), torch.gather(bins, 1, hi)
    den = cdf_hi - cdf_lo
    den = torch.where(den < eps, torch.ones_like(den), den)
    return b_lo + (u - cdf_lo) / den * (b_hi - b_lo)


# ----------------------------------------------------------------------------- projections
def hom_intrinsics(Ks: Tensor) -> Tensor:
    """multiview_aggregator.py:171-172 — (V,3,3) -> (V,4,4)."""
    K4 = torch.eye(4).expand(Ks.shape[0], 4, 4).clone()
    K4[:, :3, :3] = Ks
    return K4


def projector_projections(xyz: Tensor, K4: Tensor, poses: Tensor):
    """ibrnet/ibrnet.py:169-192 — (K4 · inv(c2w)) · [x;1]; pixel clamp ±1e6; in-front mask."""
    V = K4.shape[0]
    xyz_h = torch.cat([xyz, torch.ones_like(xyz[..., :1])], -1)
    P = K4.bmm(torch.inverse(poses))
    proj = P.bmm(xyz_h.t()[None].repeat(V, 1, 1)).permute(0, 2, 1)
    pix = proj[..., :2] / torch.clamp(proj[..., 2:3], min=1e-8)
    pix = torch.clamp(pix, min=-1e6, max=1e6)
    return pix, proj[..., 2], proj[..., 2] > 0


def inbound(pix: Tensor, h: int, w: int) -> Tensor:
    """ibrnet/ibrnet.py:126-137 — closed interval [0, w-1] x [0, h-1]."""
    return (pix[..., 0] <= w - 1.0) & (pix[..., 0] >= 0) & (pix[..., 1] <= h - 1.0) & (pix[..., 1] >= 0)


def projector_compute(xyz: Tensor, K4: Tensor, poses: Tensor, images: Tensor, featmaps: Tensor):
    """ibrnet/ibrnet.py:194-231 — bilinear (zeros pad, align_corners=True) taps of rgb + features.

    Both maps are addressed with coordinates normalised by the FULL-resolution (W-1, H-1).
    Returns rgb (N,V,3), feat (N,V,C), mask (N,V,1).
    """
    h, w = images.shape[-2:]
    pix, _, front = projector_projections(xyz, K4, poses)
    scale = torch.tensor([w - 1.0, h - 1.0])[None, None, :]
    grid = (2 * pix / scale - 1.0).unsqueeze(2)
    rgb = F.grid_sample(images, grid, align_corners=True).squeeze(-1).permute(2, 0, 1)
    feat = F.grid_sample(featmaps, grid, align_corners=True).squeeze(-1).permute(2, 0, 1)
    mask = (inbound(pix, h, w) * front).float().permute(1, 0)[..., None]
    return rgb, feat, mask


def compute_angle(xyz: Tensor, query_pose: Tensor, train_poses: Tensor) -> Tensor:
    """ibrnet/ibrnet.py:144-167 — (V,N,4): unit difference of unit view rays + their dot product."""
    to_q = query_pose[:3, 3].view(1, 1, 3) - xyz.unsqueeze(0)
    to_q = to_q / (torch.norm(to_q, dim=-1, keepdim=True) + 1e-6)
    to_t = train_poses[:, :3, 3].unsqueeze(1) - xyz.unsqueeze(0)
    to_t = to_t / (torch.norm(to_t, dim=-1, keepdim=True) + 1e-6)
    diff = to_q.expand_as(to_t) - to_t
    nrm = torch.norm(diff, dim=-1, keepdim=True)
    dot = torch.sum(to_q * to_t, dim=-1, keepdim=True)
    return torch.cat([diff / torch.clamp(nrm, min=1e-6), dot], -1)


def project_points_ref(pts: Tensor, Rt: Tensor, Ks: Tensor, h: int, w: int):
    """conditional_nerf/depth_fusion.py:78-100,113-126 — NeuRay-convention projection.

    Rt (V,3,4) world->camera. Returns pix (V,N,2), depth (V,N,1), valid (V,N) with
    |z|<1e-4 -> z:=1e-3 (invalid) and in-image test  -0.5 <= x < w-0.5.
    """
    n = pts.shape[0]
    hp = torch.cat([pts, torch.ones(n, 1)], 1)
    KRt = Ks @ Rt
    last = torch.zeros(Rt.shape[0], 1, 4)
    last[:, :, 3] = 1.0
    Hm = torch.cat([KRt, last], 1)
    cam = (Hm[:, None] @ hp[None, :, :, None])[:, :, :3, 0]
    depth = cam[:, :, 2:].clone()
    bad = torch.abs(depth) < 1e-4
    depth[bad] = 1e-3
    pix = cam[:, :, :2] / depth
    outside = (pix[..., 0] < -0.5) | (pix[..., 0] >= w - 0.5) | (pix[..., 1] < -0.5) | (pix[..., 1] >= h - 0.5)
    return pix, depth, (~bad[..., 0]) & (~outside)


def interpolate_feats(feats: Tensor, pts: Tensor, h: int, w: int, padding_mode: str, align_corners: bool) -> Tensor:
    """conditional_nerf/neuray_ops.py:14-36 — normalise by (w-1, h-1) then grid_sample."""
    xn = pts[:, :, 0] / (w - 1) * 2 - 1
    yn = pts[:, :, 1] / (h - 1) * 2 - 1
    grid = torch.stack([xn, yn], -1).unsqueeze(1)
    out = F.grid_sample(feats, grid, mode="bilinear", padding_mode=padding_mode, align_corners=align_corners)
    return out.squeeze(2).permute(0, 2, 1)


def project_ray_feats(vis_featmaps: Tensor, pts: Tensor, Rt: Tensor, Ks: Tensor, H: int, W: int):
    """depth_fusion.py:128-147 + :60-76 — only the entries the path reads: depth, mask, ray_feats.

    The /4-resolution map is sampled with border padding and align_corners=False (fh != h branch).
    """
    pix, depth, valid = project_points_ref(pts, Rt, Ks, H, W)
    fh, fw = vis_featmaps.shape[-2:]
    ac = (fh == H and fw == W)
    rf = interpolate_feats(vis_featmaps, pix, H, W, "border", ac) * valid.float().unsqueeze(-1)
    return depth, valid.float().unsqueeze(-1), rf


# ----------------------------------------------------------------------------- NeuRay decoder
def _mlp3(p: Dict[str, Tensor], prefix: str, x: Tensor) -> Tensor:
    x = F.elu(F.linear(x, p[f"{prefix}.0.weight"], p[f"{prefix}.0.bias"]))
    x = F.elu(F.linear(x, p[f"{prefix}.2.weight"], p[f"{prefix}.2.bias"]))
    return F.linear(x, p[f"{prefix}.4.weight"], p[f"{prefix}.4.bias"])


def dist_decoder(p: Dict[str, Tensor], feats: Tensor):
    """conditional_nerf/visibility_decoder.py:64-107 — mean, var(+0.05), vis, aw."""
    pre = "multiview_aggregator.dist_decoder"
    mean = F.softplus(_mlp3(p, f"{pre}.mean_decoder", feats))
    var = F.softplus(_mlp3(p, f"{pre}.var_decoder", feats)) + 0.05
    aw = torch.sigmoid(_mlp3(p, f"{pre}.aw_decoder", feats))
    vis = torch.sigmoid(_mlp3(p, f"{pre}.vis_decoder", feats))
    return mean, var, vis, aw


def decode_ref_depths(mean: Tensor, depth_range: Tensor) -> Tensor:
    """visibility_decoder.py:140-148 — first mixture mean -> metric depth, clamped to [near, far]."""
    near = depth_range[:, 0][:, None, None]
    far = depth_range[:, 1][:, None, None]
    ni, fi = -1 / near, -1 / far
    d = -1 / (mean * (fi - ni) + ni)
    return d.clamp(near.min(), far.max())[:, :, 0]


def _inv_norm_depth(depth: Tensor, near: Tensor, far: Tensor) -> Tensor:
    ni, fi = -1 / near, -1 / far
    return (-1 / torch.clamp(depth, min=1e-5) - ni) / (fi - ni)


def compute_visibility(depth: Tensor, mean: Tensor, var: Tensor, vis: Tensor, aw: Tensor, depth_range: Tensor) -> Tensor:
    """visibility_decoder.py:109-138 — mixture-of-logistics survival at the projected depth."""
    d = _inv_norm_depth(depth, depth_range[:, 0][:, None, None], depth_range[:, 1][:, None, None])
    mix = torch.cat([aw, 1 - aw], -1)
    cdf = (0.5 + 0.5 * torch.tanh((d - mean) * var)) * vis
    return torch.sum((1 - cdf) * mix, -1)


def predict_visibility(p, vis_featmaps, pts, Rt, Ks, H, W, depth_range_v):
    """multiview_aggregator.py:63-93 — (vis (V,N,1), depth_diff (V,N))."""
    depth, mask, rf = project_ray_feats(vis_featmaps, pts, Rt, Ks, H, W)
    mean, var, vis, aw = dist_decoder(p, rf)
    ref_d = decode_ref_depths(mean, depth_range_v)
    ddiff = (depth.squeeze(-1) - ref_d).abs() / (depth_range_v[:, 1:] - depth_range_v[:, :1])
    v = compute_visibility(depth, mean, var, vis, aw, depth_range_v)
    return v.reshape(*mask.shape) * mask, ddiff


def mv_aggregate(p: Dict[str, Tensor], frame: Dict, xyz: Tensor):
    """multiview_aggregator.py:156-222 — visibility-weighted mean/var over views -> out_fc.

    Returns (feat (N,W), rgb_feat (N,V,3+C), vis (N,V,1), mask1 (N,V,1)).  `mask1` (Projector
    in-bounds mask) is what render_rays recomputes for the valid-ray mask (model.py:563-571).
    """
    Ks, poses, images = frame["topk_Ks"], frame["topk_poses"], frame["topk_images"]
    featmaps = frame["feat_fine_src"].permute(0, 3, 1, 2)
    depth_range = frame["depth_range"][0]
    V = images.shape[0]
    H, W = images.shape[-2:]
    rgb, feat, mask1 = projector_compute(xyz, hom_intrinsics(Ks), poses, images, featmaps)
    rgb_feat = torch.cat([rgb, feat], -1)
    dr_v = depth_range.view(1, 2).repeat(V, 1).float()
    vis, ddiff = predict_visibility(p, frame["vis_featmaps"], xyz, poses.inverse()[:, :3], Ks, H, W, dr_v)
    vis = vis.view(V, -1, 1).permute(1, 0, 2)
    ddiff = ddiff.view(V, -1, 1).permute(1, 0, 2)
    wgt = vis / (torch.sum(vis, dim=1, keepdim=True) + 1e-8)

    def mean_var(x):  # ibrnet.py:8-12
        m = torch.sum(x * wgt, dim=1, keepdim=True)
        return m, torch.sum(wgt * (x - m) ** 2, dim=1, keepdim=True)

    m1, v1 = mean_var(rgb_feat)
    m2, v2 = mean_var(ddiff)
    g = torch.cat([torch.cat([m1, v1, m2, v2], -1).squeeze(1), wgt.mean(dim=1)], -1)
    pre = "multiview_aggregator.out_fc"
    y = F.elu(F.linear(g, p[f"{pre}.0.weight"], p[f"{pre}.0.bias"]))
    y = F.elu(F.linear(y, p[f"{pre}.2.weight"], p[f"{pre}.2.bias"]))
    return y, rgb_feat, vis, mask1


# ----------------------------------------------------------------------------- hierarchical (a20)
def _coords2rays(coords, Rt, Ks):
    """depth_fusion.py:9-30 — un-normalised K^-1[u,v,1] directions in world frame."""
    rot = Rt[:, :, :3].unsqueeze(1).permute(0, 1, 3, 2)
    trans = -rot @ Rt[:, :, 3:].unsqueeze(1)
    qn, rn, _ = coords.shape
    centers = trans.repeat(1, rn, 1, 1).squeeze(-1)
    ch = torch.cat([coords, torch.ones(qn, rn, 1)], 2)
    cam = torch.inverse(Ks).unsqueeze(1) @ ch.unsqueeze(3)
    cam = rot @ cam + trans
    return centers, cam.squeeze(3) - centers


def _depth2inv_dists(depth, depth_range):
    """depth_fusion.py:47-58 — interval lengths in normalised inverse depth; last = 1e6."""
    ni, fi = -1 / depth_range[:, 0], -1 / depth_range[:, 1]
    di = (-1 / depth - ni[:, None, None]) / (fi - ni)[:, None, None]
    d = di[..., 1:] - di[..., :-1]
    return torch.cat([d, torch.full([*di.shape[:-1], 1], 1e6)], -1)


def _near_far_ref(depth, interval, depth_range):
    """visibility_decoder.py:6-51, is_ref=True, variable interval."""
    ni = -1 / depth_range[:, 0][:, None, None, None]
    fi = -1 / depth_range[:, 1][:, None, None, None]
    d = (-1 / torch.clamp(depth, min=1e-5) - ni) / (fi - ni)
    half = interval / 2
    ext = torch.cat([half[..., 0:1], half], -1)
    return d - ext[..., :-1], d + ext[..., 1:]


def compute_prob(depth, interval, mean, var, vis, aw, depth_range):
    """visibility_decoder.py:150-181 (is_ref=True) -> (alpha logit, visibility)."""
    near, far = _near_far_ref(depth, interval, depth_range)
    mix = torch.cat([aw, 1 - aw], -1)
    near, far = near[..., None], far[..., None]
    c0 = (0.5 + 0.5 * torch.tanh((near - mean) * var)) * vis
    c1 = (0.5 + 0.5 * torch.tanh((far - mean) * var)) * vis
    visibility = torch.sum((1 - c0) * mix, -1)
    hit = torch.sum((c1 - c0) * mix, -1)
    eps = 1e-5
    return torch.log(hit / (visibility - hit + eps) + eps), visibility


def predict_weights_from_neuray(p, frame, rays, que_depth: Tensor) -> Tensor:
    """multiview_aggregator.py:95-154 — coarse hit-probability weights (R, dn) from the support views."""
    Ks, poses, images = frame["topk_Ks"], frame["topk_poses"], frame["topk_images"]
    dr = frame["depth_range"][0]
    V = images.shape[0]
    H, W = images.shape[-2:]
    dr_v = dr.view(1, 2).repeat(V, 1).float()
    Rt = poses.inverse()[:, :3]
    q_Rt = rays["pose"][None].inverse()[:, :3]
    rn, dn = que_depth.shape
    qd = que_depth[None]
    dists = _depth2inv_dists(qd, dr[None].float())
    cen, dirs = _coords2rays(rays["pixel_coordinates"][None], q_Rt, rays["K"][None])
    pts = cen.unsqueeze(2) + dirs.unsqueeze(2) * qd.unsqueeze(3)
    depth, mask, rf = project_ray_feats(frame["vis_featmaps"], pts.view(-1, 3), Rt, Ks, H, W)
    mean, var, vis, aw = dist_decoder(p, rf)
    alpha, visib = compute_prob(depth.view(V, 1, rn, dn), dists.view(1, 1, rn, dn), mean.view(V, 1, rn, dn, -1),
                                var.view(V, 1, rn, dn, -1), vis.view(V, 1, rn, dn, -1), aw.view(V, 1, rn, dn, -1), dr_v)
    ground = -15
    m = mask.view(V, 1, rn, dn, 1)
    a = alpha.reshape(V, 1, rn, dn, 1) * m + (1 - m) * ground
    vs = visib.reshape(V, 1, rn, dn, 1) * m
    a = (a * vs).sum(0) / torch.clip(vs.sum(0), min=1e-8)
    none = (torch.sum(m.int().squeeze(-1), 0) == 0).float().unsqueeze(-1)
    a = a * (1 - none) + none * ground
    a = torch.sigmoid(a).squeeze(0).squeeze(-1)
    T = torch.cumprod(torch.cat([torch.ones_like(a[:, :1]), 1 - a], -1)[:, :-1], -1)
    return a * T


# ----------------------------------------------------------------------------- KNN (a8)
_knn_lib = None


def _load_knn_lib():
    global _knn_lib
    if _knn_lib is None:
        path = os.path.join(_HERE, "libknn_oracle.so")
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            lib.knn_oracle_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                           ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
            lib.knn_oracle_f32.restype = ctypes.c_int
            _knn_lib = lib
        else:
            _knn_lib = False
    return _knn_lib


def knn_points_np(q: np.ndarray, p: np.ndarray, K: int):
    """numpy restatement of ops/knn/src/knn_cpu.cpp:13-64 (+ knn_utils.py:60-74 ascending sort).

    The heap there keeps the K lexicographically smallest (dist2, idx) tuples (strict-'<'
    admission in increasing idx order, eviction of the max tuple); dist2 is the fp32 chain
    ((dx*dx) + dy*dy) + dz*dz without fma.  Slots beyond P2 stay 0 (knn_cpu.cpp:23-24).
    """
    q = np.ascontiguousarray(q, np.float32)
    p = np.ascontiguousarray(p, np.float32)
    n, m = q.shape[0], p.shape[0]
    idx = np.zeros((n, K), np.int64)
    d2o = np.zeros((n, K), np.float32)
    kk = min(K, m)
    step = max(1, (1 << 24) // max(m, 1))
    for s in range(0, n, step):
        qq = q[s:s + step]
        acc = np.zeros((qq.shape[0], m), np.float32)
        for d in range(q.shape[1]):
            diff = qq[:, d:d + 1] - p[None, :, d]
            acc = acc + diff * diff
        order = np.argsort(acc, axis=1, kind="stable")[:, :kk]
        idx[s:s + step, :kk] = order
        d2o[s:s + step, :kk] = np.take_along_axis(acc, order, 1)
    return d2o, idx


def knn_points(q: Tensor, p: Tensor, K: int, threads: int = 1):
    """(dist2 (N,K) fp32 ascending, idx (N,K) int64) — C restatement when built, else numpy."""
    lib = _load_knn_lib()
    if lib:
        qn = np.ascontiguousarray(q.detach().numpy(), np.float32)
        pn = np.ascontiguousarray(p.detach().numpy(), np.float32)
        n, m = qn.shape[0], pn.shape[0]
        idx = np.zeros((n, K), np.int64)
        d2 = np.zeros((n, K), np.float32)
        rc = lib.knn_oracle_f32(qn.ctypes.data, n, pn.ctypes.data, m, K, d2.ctypes.data, idx.ctypes.data, threads)
        assert rc == 0
        return torch.from_numpy(d2), torch.from_numpy(idx)
    d2, idx = knn_points_np(q.detach().numpy(), p.detach().numpy(), K)
    return torch.from_numpy(d2), torch.from_numpy(idx)


def knn_gather(x: Tensor, idx: Tensor, K_valid: Optional[int] = None) -> Tensor:
    """ops/knn/knn_utils.py:176-222 — x[idx]; columns k >= len(x) are zero-filled."""
    out = x[idx]
    m = x.shape[0]
    if m < idx.shape[1]:
        out[:, m:] = 0.0
    return out


# ----------------------------------------------------------------------------- neural-point branch
def mha(p: Dict[str, Tensor], q: Tensor, k: Tensor, v: Tensor, n_head: int = 4, d_k: int = 32) -> Tensor:
    """ibrnet/ibrnet.py:89-119 — 4-head attention, no biases, residual + LayerNorm(eps=1e-6)."""
    pre = "base_mlp_attn"
    b, lq, lk = q.size(0), q.size(1), k.size(1)
    res = q
    qq = F.linear(q, p[f"{pre}.w_qs.weight"]).view(b, lq, n_head, d_k).transpose(1, 2)
    kk = F.linear(k, p[f"{pre}.w_ks.weight"]).view(b, lk, n_head, d_k).transpose(1, 2)
    vv = F.linear(v, p[f"{pre}.w_vs.weight"]).view(b, lk, n_head, d_k).transpose(1, 2)
    att = F.softmax(torch.matmul(qq / (d_k ** 0.5), kk.transpose(2, 3)), dim=-1)
    o = torch.matmul(att, vv).transpose(1, 2).contiguous().view(b, lq, -1)
    o = F.linear(o, p[f"{pre}.fc.weight"]) + res
    return F.layer_norm(o, (o.shape[-1],), p[f"{pre}.layer_norm.weight"], p[f"{pre}.layer_norm.bias"], eps=1e-6)


def _lrelu(x):
    return F.leaky_relu(x, 0.01)


def query(p: Dict[str, Tensor], frame: Dict, xyz: Tensor, direction: Optional[Tensor], K: int = 8,
          knn_threads: int = 1, timers: Optional[dict] = None) -> Dict[str, Tensor]:
    """conditional_nerf/model.py:344-436 — conditional feature of each 3-D sample."""
    import time
    t0 = time.perf_counter()
    mv, rgb_feat, vis, mask1 = mv_aggregate(p, frame, xyz)
    t1 = time.perf_counter()
    sp = frame["support_fine"]
    d2, idx = knn_points(xyz, sp["xyz"], K, threads=knn_threads)
    t2 = time.perf_counter()
    dists = d2.sqrt()
    nb_xyz = knn_gather(sp["xyz"], idx)
    nb_feat = knn_gather(sp["feature"], idx)
    nb_conf = knn_gather(sp["confidence"], idx)
    nb_dir = knn_gather(sp["direction"], idx)
    if direction is None:
        direction = nb_dir[:, 0, :]
    off = xyz[:, None, :].repeat(1, K, 1) - nb_xyz
    rd = direction[:, :3].unsqueeze(1) - nb_dir[..., :3]
    rd = rd / (torch.norm(rd, dim=-1, keepdim=True) + 1e-8)
    dot = torch.sum(direction[:, :3].unsqueeze(1) * nb_dir[..., :3], dim=-1, keepdim=True)
    rd = torch.cat([rd, dot], -1)
    near, far = frame["depth_range"][0]
    a = _lrelu(F.linear(rd, p["ray_diff_fc.0.weight"], p["ray_diff_fc.0.bias"]))
    a = _lrelu(F.linear(a, p["ray_diff_fc.2.weight"], p["ray_diff_fc.2.bias"]))
    x = torch.cat([nb_feat, posenc(off / (far - near)), a], -1)
    for i in (0, 2, 4):
        x = _lrelu(F.linear(x, p[f"base_mlp.{i}.weight"], p[f"base_mlp.{i}.bias"]))
    feat = mha(p, mv.unsqueeze(1).repeat(1, K, 1), x, x)
    lg = F.linear(_lrelu(F.linear(feat, p["base_mlp_agg_weight.0.weight"], p["base_mlp_agg_weight.0.bias"])),
                  p["base_mlp_agg_weight.2.weight"], p["base_mlp_agg_weight.2.bias"]).squeeze(-1)
    corr = F.softmax(lg, dim=1)
    w = 1.0 / torch.clamp(dists, min=1e-8)
    w = w * corr
    w = w * nb_conf.squeeze(-1)
    w = w / torch.clamp(w.sum(dim=1, keepdim=True), min=1e-8)
    agg = (feat * w.unsqueeze(-1)).sum(dim=1)
    if timers is not None:
        timers["mv_aggregate"] = timers.get("mv_aggregate", 0.0) + (t1 - t0)
        timers["knn"] = timers.get("knn", 0.0) + (t2 - t1)
        timers["point_mlp"] = timers.get("point_mlp", 0.0) + (time.perf_counter() - t2)
    return {"feature_agg": agg, "feature": feat, "weights": w, "multiview_feature": rgb_feat,
            "multiview_visibility": vis, "multiview_feature_agg": mv, "mask1": mask1,
            "knn_d2": d2, "knn_idx": idx}


# ----------------------------------------------------------------------------- ray U-Net (a13)
def ray_unet(p: Dict[str, Tensor], x: Tensor) -> Tensor:
    """conditional_nerf/ray_unet.py:5-69 — x (R, W, S) -> (R, W, S)."""
    pre = "ray_unet"

    def block(name, t, transposed=False):
        w, b = p[f"{pre}.{name}.0.weight"], p[f"{pre}.{name}.0.bias"]
        if transposed:
            t = F.conv_transpose1d(t, w, b, stride=2, padding=1, output_padding=1)
        else:
            t = F.conv1d(t, w, b, stride=1, padding=1)
        g, be = p[f"{pre}.{name}.1.weight"], p[f"{pre}.{name}.1.bias"]
        return F.elu(F.layer_norm(t, tuple(g.shape), g, be, eps=1e-5))

    c1 = F.max_pool1d(block("conv1", x), 2)
    c2 = F.max_pool1d(block("conv2", c1), 2)
    c3 = F.max_pool1d(block("conv3", c2), 2)
    x0 = block("trans_conv3", c3, True)
    x1 = block("trans_conv2", torch.cat([c2, x0], 1), True)
    x2 = block("trans_conv1", torch.cat([c1, x1], 1), True)
    return block("conv_out", torch.cat([x, x2], 1))


# ----------------------------------------------------------------------------- render_rays (a2-a18)
def composite(sigma: Tensor, z_vals: Tensor):
    """conditional_nerf/model.py:544-553 — deltas (last 1e2), alpha, exclusive-cumprod T, weights."""
    deltas = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], 1e2 * torch.ones_like(z_vals[:, :1])], -1)
    alphas = 1 - torch.exp(-deltas * sigma)
    T = torch.cumprod(torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas], -1)[:, :-1], -1)
    return alphas * T


def render_rays(p: Dict[str, Tensor], frame: Dict, rays: Dict, n_samples: int, n_importance: int = 0,
                u: Optional[Tensor] = None, lindisp: bool = False, white_bkgd: Optional[bool] = None,
                knn_threads: int = 1, timers: Optional[dict] = None, intermediates: bool = False) -> Dict[str, Tensor]:
    """conditional_nerf/model.py:472-600 (eval mode: no `beta`)."""
    near, far = rays["depth_range"]
    o, d = rays["rays_o"], rays["rays_d"]
    R = o.shape[0]
    S = n_samples
    z = sample_depths(S, near, far, lindisp).expand(R, S).contiguous()
    depth_coarse = None
    if n_importance > 0:
        zc = sample_depths(64, near, far, lindisp).expand(R, 64).contiguous()
        wc = predict_weights_from_neuray(p, frame, rays, zc)
        depth_coarse = (wc * zc).sum(1)
        mid = 0.5 * (zc[:, :-1] + zc[:, 1:])
        zf = sample_pdf(mid, wc[:, 1:-1], n_importance, u)
        z = torch.sort(torch.cat([z, zf], -1), -1)[0]
        S = n_samples + n_importance
    xyz = (o[:, None, :] + d[:, None, :] * z[..., None]).view(-1, 3)
    dir_d = torch.cat([d[:, None, :].repeat(1, S, 1).view(-1, 3), z.reshape(-1, 1)], -1)
    qd = query(p, frame, xyz, dir_d, K=8, knn_threads=knn_threads, timers=timers)
    agg, mvf, mvv = qd["feature_agg"], qd["multiview_feature"], qd["multiview_visibility"]
    import time
    t0 = time.perf_counter()
    W = agg.shape[1]
    geo = ray_unet(p, agg.view(R, S, W).permute(0, 2, 1)).permute(0, 2, 1).reshape(R * S, W)
    t1 = time.perf_counter()
    sigma = F.softplus(F.linear(geo, p["sigma_mlp.0.weight"], p["sigma_mlp.0.bias"]))
    V = mvf.shape[1]
    ang = compute_angle(xyz, frame["pose"], frame["topk_poses"]).permute(1, 0, 2)
    xb = torch.cat([agg.unsqueeze(1).expand(-1, V, -1), mvf, mvv, ang], -1)
    for i in (0, 2):
        xb = _lrelu(F.linear(xb, p[f"rgb_blending_mlp.{i}.weight"], p[f"rgb_blending_mlp.{i}.bias"]))
    bw = F.linear(xb, p["rgb_blending_mlp.4.weight"], p["rgb_blending_mlp.4.bias"])
    bw = F.softmax(bw.masked_fill(mvv == 0, -1e9), dim=1)
    rgb_s = torch.sum(mvf[:, :, :3] * bw, dim=1).view(R, S, 3)
    wts = composite(sigma.view(R, S), z)
    wsum = wts.sum(1)
    rgb = (wts[..., None] * rgb_s).sum(1)
    wb = frame.get("white_bkgd", False) if white_bkgd is None else white_bkgd
    if wb:
        rgb = rgb + (1 - wsum[:, None])
    depth = (wts * z).sum(1)
    dunc = (wts * (z - depth[:, None]) ** 2).sum(1)
    valid = (qd["mask1"].view(R, S, V).sum(2) > 1).float().sum(1) > 8
    ft = F.linear(_lrelu(F.linear(agg, p["feat_mlp.0.weight"], p["feat_mlp.0.bias"])), p["feat_mlp.2.weight"], p["feat_mlp.2.bias"])
    feat = (wts[..., None] * ft.view(R, S, -1)).sum(1)
    if timers is not None:
        timers["ray_unet"] = timers.get("ray_unet", 0.0) + (t1 - t0)
        timers["heads_composite"] = timers.get("heads_composite", 0.0) + (time.perf_counter() - t1)
    out = {"rgb": rgb, "depth": depth, "weights": wts, "mask": valid, "depth_uncertainty": dunc, "feat": feat,
           "z_vals": z}   # (the depths the samples were placed at: not an output of model.py:577-598; callers use it to find hard-threshold border samples)
    if depth_coarse is not None:
        out["depth_coarse"] = depth_coarse
    if intermediates:
        out.update({"z_vals": z, "knn_d2": qd["knn_d2"], "knn_idx": qd["knn_idx"], "multiview_feature_agg": qd["multiview_feature_agg"],
                    "feature_agg": agg, "sigma": sigma.view(R, S), "multiview_visibility": mvv.squeeze(-1), "rgb_samples": rgb_s,
                    "geo": geo})
    return out


# ----------------------------------------------------------------------------- helpers for callers
def to_torch(d):
    """Recursively wrap numpy arrays of a synth recipe dict into torch tensors (shares memory)."""
    if isinstance(d, dict):
        return {k: to_torch(v) for k, v in d.items()}
    if isinstance(d, np.ndarray):
        return torch.from_numpy(d)
    return d
