"""Per-frame visibility feature maps — stays on PyTorch-ROCm ("backbone" side of north_star).

Own PyTorch restatement of the reference's per-frame CNN so that reference checkpoints load by name:
`DepthFusionNet` (conditional_nerf/depth_fusion.py:239-282) = cross-view depth/colour consistency features
(:163-207; on the HIP library, `nl_cross_view_features`) -> `ResEncoder` (conditional_nerf/neuray_ops.py:164-239) + a 2-layer depth skip -> 32-channel map at 1/4
resolution.  It runs once per query frame (not per ray); its output is what `nl_frame_create` receives as
`vis_featmaps`.  Module/parameter names mirror the reference's state_dict (72 tensors under
`multiview_aggregator.depth_fusion.*`).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


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
        """The hand-made input channels (normalised inverse depth + cross-view consistency statistics, reference :150-227) come
        from the HIP library in one pass; the CNN is PyTorch-ROCm (MIOpen)."""
        from .frame_setup import cross_view_features
        near, far = [float(x) for x in depth_range.reshape(-1)[:2]]
        return self.encode(cross_view_features(imgs, depths, Ks, poses, near, far))

    def encode(self, cnn_in):
        """(V,12,H,W) = [rgb | normalised inverse depth | 8 consistency channels] -> (V,32,H/4,W/4) (reference :279-282)."""
        x = self.fuse_net(cnn_in)
        return self.conv_out(torch.cat([self.depth_skip(cnn_in[:, 3:4]), x], 1))
