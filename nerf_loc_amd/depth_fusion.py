"""Per-frame visibility feature maps — stays on PyTorch-ROCm ("backbone" side of north_star).

Own PyTorch restatement of the reference's per-frame CNN so that reference checkpoints load by name:
`DepthFusionNet` (conditional_nerf/depth_fusion.py:239-282) = cross-view depth/colour consistency features
(:163-207) -> `ResEncoder` (conditional_nerf/neuray_ops.py:164-239) + a 2-layer depth skip -> 32-channel map at 1/4
resolution.  It runs once per query frame (not per ray); its output is what `nl_frame_create` receives as
`vis_featmaps`.  Module/parameter names mirror the reference's state_dict (72 tensors under
`multiview_aggregator.depth_fusion.*`).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def _grid_sample_pts(feats, pts, h=None, w=None, padding_mode="zeros", align_corners=False):
    """neuray_ops.py:14-36 — pts (b,n,2) in pixels of an (h,w) image -> (b,n,f)."""
    _, _, ch, cw = feats.shape
    if h is None and w is None:
        h, w = ch, cw
    gx = pts[:, :, 0] / (w - 1) * 2 - 1
    gy = pts[:, :, 1] / (h - 1) * 2 - 1
    grid = torch.stack([gx, gy], -1).unsqueeze(1)
    out = F.grid_sample(feats, grid, mode="bilinear", padding_mode=padding_mode, align_corners=align_corners)
    return out.squeeze(2).permute(0, 2, 1)


def _project(pts, Rt, Ks, h, w):
    """depth_fusion.py:78-126 — NeuRay projection of pts (n,3) into every view: pix (V,n,2), depth (V,n,1), valid (V,n)."""
    KRt = Ks @ Rt                                                                    # (V,3,4)
    # rows 0-2 of [K Rt; 0 0 0 1] . [x y z 1]^T for every (view, point) as four broadcast multiply-adds.  (The literal
    # form, a (V,1,4,4) @ (1,n,4,1) matmul, becomes V*n = 5e5 batched 4x4 GEMMs: 59 ms of the 136-ms frame setup on ROCm.)
    x, y, z = pts[None, :, 0:1], pts[None, :, 1:2], pts[None, :, 2:3]
    cam = ((KRt[:, None, :, 0] * x + KRt[:, None, :, 1] * y) + KRt[:, None, :, 2] * z) + KRt[:, None, :, 3]
    depth = cam[:, :, 2:].clone()
    bad = depth.abs() < 1e-4
    depth[bad] = 1e-3
    pix = cam[:, :, :2] / depth
    outside = (pix[..., 0] < -0.5) | (pix[..., 0] >= w - 0.5) | (pix[..., 1] < -0.5) | (pix[..., 1] >= h - 0.5)
    return pix, depth, (~bad[..., 0]) & (~outside)


def _masked_mean_var(x, mask, dim):
    """neuray_ops.py:38-43."""
    mask = mask.float()
    s = torch.clamp_min(mask.sum(dim, keepdim=True), 1e-4)
    mean = (x * mask).sum(dim, keepdim=True) / s
    var = ((x - mean) ** 2 * mask).sum(dim, keepdim=True) / s
    return mean, var


def cross_view_consistency(imgs, depth_norm, Ks, Rt, depth_range):
    """depth_fusion.py:150-207 (depth2pts3d + get_diff_feats): re-project every view's depth into every other view and
    summarise colour / inverse-depth disagreement -> (V,8,h,w)."""
    V, _, h, w = imgs.shape
    near = depth_range[:, 0][:, None, None, None]
    far = depth_range[:, 1][:, None, None, None]
    ni, fi = -1 / near, -1 / far
    depth = -1 / (depth_norm * (fi - ni) + ni)
    ys, xs = torch.meshgrid(torch.arange(h, device=imgs.device), torch.arange(w, device=imgs.device), indexing="ij")
    coords = torch.stack([xs, ys, torch.ones_like(xs)], -1).float()[None]           # 1,h,w,3 = (x, y, 1)
    pts = (depth.permute(0, 2, 3, 1).unsqueeze(-1) * coords.unsqueeze(-2)).reshape(V, h * w, 3).permute(0, 2, 1)
    pts = torch.inverse(Ks) @ pts
    R = Rt[:, :3, :3].permute(0, 2, 1)
    t = -R @ Rt[:, :3, 3:]
    pts = (R @ pts + t).permute(0, 2, 1).reshape(-1, 3)                              # world points of all views
    pix, prj_depth, valid = _project(pts, Rt, Ks, h, w)
    d_int = _grid_sample_pts(depth, pix, padding_mode="border", align_corners=True)
    c_int = _grid_sample_pts(imgs, pix, padding_mode="border", align_corners=True)
    rgb_diff = (c_int - imgs.permute(0, 2, 3, 1).reshape(1, V * h * w, 3)).abs()
    d_int = torch.clamp(d_int, min=1e-5)
    prj_depth = torch.clamp(prj_depth, min=1e-5)
    d_diff = (-1 / d_int + 1 / prj_depth).abs()
    ni2, fi2 = -1 / depth_range[:, 0][:, None, None], -1 / depth_range[:, 1][:, None, None]
    d_diff = torch.clamp(d_diff / (fi2 - ni2), max=1.5)
    m = valid.float().unsqueeze(-1)
    dm, dv = _masked_mean_var(d_diff, m, 0)
    cm, cv = _masked_mean_var(rgb_diff, m, 0)

    def fold(x, c):
        return x.reshape(V, h, w, c).permute(0, 3, 1, 2)
    return torch.cat([fold(cm, 3), fold(cv, 3), fold(dm, 1), fold(dv, 1)], 1)


def _conv3x3(i, o, stride=1):
    return nn.Conv2d(i, o, 3, stride, 1, bias=False, padding_mode="reflect")


class _BasicBlock(nn.Module):
    """neuray_ops.py:93-131 (InstanceNorm variant)."""

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = nn.InstanceNorm2d(planes, track_running_stats=False, affine=True)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = nn.InstanceNorm2d(planes, track_running_stats=False, affine=True)
        self.downsample = downsample

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idn)


class _ConvIN(nn.Module):
    """neuray_ops.py:133-147 — reflect-padded conv + InstanceNorm + ELU."""

    def __init__(self, i, o, k, stride):
        super().__init__()
        self.conv = nn.Conv2d(i, o, k, stride, (k - 1) // 2, padding_mode="reflect")
        self.bn = nn.InstanceNorm2d(o, track_running_stats=False, affine=True)

    def forward(self, x):
        return F.elu(self.bn(self.conv(x)), inplace=True)


class _UpConv(nn.Module):
    """neuray_ops.py:149-157."""

    def __init__(self, i, o, k, scale):
        super().__init__()
        self.scale = scale
        self.conv = _ConvIN(i, o, k, 1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=self.scale, align_corners=True, mode="bilinear"))


class ResEncoder(nn.Module):
    """neuray_ops.py:159-239 — 12-channel input -> 32 channels at 1/4 resolution."""

    def __init__(self):
        super().__init__()
        self.inplanes = 32
        self.conv1 = nn.Conv2d(12, 32, 8, 2, 2, bias=False, padding_mode="reflect")
        self.bn1 = nn.InstanceNorm2d(32, track_running_stats=False, affine=True)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = self._layer(32, 2, 2)
        self.layer2 = self._layer(64, 2, 2)
        self.layer3 = self._layer(128, 2, 2)
        self.upconv3 = _UpConv(128, 64, 3, 2)
        self.iconv3 = _ConvIN(128, 64, 3, 1)
        self.upconv2 = _UpConv(64, 32, 3, 2)
        self.iconv2 = _ConvIN(64, 32, 3, 1)
        self.out_conv = nn.Conv2d(32, 32, 1, 1)

    def _layer(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False, padding_mode="reflect"),
                                 nn.InstanceNorm2d(planes, track_running_stats=False, affine=True))
        layers = [_BasicBlock(self.inplanes, planes, stride, down)]
        self.inplanes = planes
        layers += [_BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    @staticmethod
    def _skip(x1, x2):
        dy, dx = x2.size(2) - x1.size(2), x2.size(3) - x1.size(3)
        x1 = F.pad(x1, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
        return torch.cat([x2, x1], 1)

    def forward(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        x1 = self.layer1(x)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        x = self.iconv3(self._skip(x2, self.upconv3(x3)))
        x = self.iconv2(self._skip(x1, self.upconv2(x)))
        return self.out_conv(x)


class DepthFusionNet(nn.Module):
    """depth_fusion.py:239-282 — (imgs, depths, Ks, c2w poses, depth_range) -> (V,32,H/4,W/4)."""

    def __init__(self, cfg=None, in_channels=None):
        super().__init__()
        self.fuse_net = ResEncoder()
        self.depth_skip = nn.Sequential(nn.Conv2d(1, 8, 2, 2), nn.ReLU(True), nn.Conv2d(8, 16, 2, 2))
        self.conv_out = nn.Conv2d(16 + 32, 32, 1, 1)
        self.out_channels = 32

    def forward(self, imgs, feats, depths, Ks, poses, depth_range):
        V = imgs.shape[0]
        dr = depth_range.view(1, 2).repeat(V, 1).float()
        near = dr[:, 0][:, None, None, None]
        far = dr[:, 1][:, None, None, None]
        ni, fi = -1 / near, -1 / far
        d = torch.clamp(depths.unsqueeze(1), min=1e-5)
        d = torch.clamp((-1 / d - ni) / (fi - ni), min=0, max=1.0)      # extract_depth_for_init (:209-227)
        diff = cross_view_consistency(imgs, d, Ks, poses.inverse()[:, :3], dr)
        x = self.fuse_net(torch.cat([imgs, d, diff], 1))
        return self.conv_out(torch.cat([self.depth_skip(d), x], 1))
