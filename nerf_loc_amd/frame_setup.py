"""Per-frame setup on the HIP library: the host side of `nl_cross_view_features` and
`nl_backproject_support`.  Torch tensors are device memory + the current stream here; the arithmetic is in
`csrc/setup.hip`.  No CPU path: the functions raise on anything but HIP tensors."""
from __future__ import annotations

import ctypes as ct
from typing import Tuple

import torch

from . import _lib as L


def _f32(t: torch.Tensor, what: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{what} must be a tensor on the HIP device: the per-frame setup has no CPU fallback")
    return t.detach().to(torch.float32).contiguous()


def _workspace(lib, V: int, H: int, W: int, stride: int, device) -> torch.Tensor:
    n = lib.nl_setup_workspace_bytes(V, H, W, stride)
    if n == 0:
        raise ValueError(f"bad setup shape V={V} H={H} W={W} stride={stride}")
    return torch.empty(n, dtype=torch.uint8, device=device)


def cross_view_features(imgs: torch.Tensor, depths: torch.Tensor, Ks: torch.Tensor, c2w: torch.Tensor, near: float, far: float) -> torch.Tensor:
    """conditional_nerf/depth_fusion.py:150-227 + the cat at :269-278 -> the (V,12,H,W) tensor DepthFusionNet's CNN consumes."""
    lib = L.load()
    imgs, depths, Ks, c2w = _f32(imgs, "imgs"), _f32(depths, "depths"), _f32(Ks, "Ks"), _f32(c2w, "poses")
    V, _, H, W = imgs.shape
    out = torch.empty(V, 12, H, W, dtype=torch.float32, device=imgs.device)
    ws = _workspace(lib, V, H, W, 1, imgs.device)
    st = torch.cuda.current_stream(imgs.device).cuda_stream
    L.check(lib.nl_cross_view_features(imgs.data_ptr(), depths.data_ptr(), Ks.data_ptr(), c2w.data_ptr(), V, H, W, float(near), float(far),
                                       out.data_ptr(), ws.data_ptr(), ws.numel(), st), "nl_cross_view_features")
    ws.record_stream(torch.cuda.current_stream(imgs.device))
    return out


def backproject_support(imgs: torch.Tensor, feats: torch.Tensor, depths: torch.Tensor, Ks: torch.Tensor, c2w: torch.Tensor,
                        stride: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """conditional_nerf/model.py:203-265 -> (feature (M,3+C), xyz world (M,3), xyz in view 0's camera (M,3), direction+depth (M,4))."""
    lib = L.load()
    imgs, feats, depths, Ks, c2w = _f32(imgs, "imgs"), _f32(feats, "feats"), _f32(depths, "depths"), _f32(Ks, "Ks"), _f32(c2w, "poses")
    V, _, H, W = imgs.shape
    _, fh, fw, C = feats.shape
    cap = V * (H // stride) * (W // stride)
    dev = imgs.device
    feature = torch.empty(cap, 3 + C, dtype=torch.float32, device=dev)
    xyz = torch.empty(cap, 3, dtype=torch.float32, device=dev)
    ref = torch.empty(cap, 3, dtype=torch.float32, device=dev)
    direction = torch.empty(cap, 4, dtype=torch.float32, device=dev)
    ws = _workspace(lib, V, H, W, stride, dev)
    m = ct.c_int64(0)
    st = torch.cuda.current_stream(dev).cuda_stream
    L.check(lib.nl_backproject_support(imgs.data_ptr(), feats.data_ptr(), depths.data_ptr(), Ks.data_ptr(), c2w.data_ptr(), V, H, W, fh, fw, C,
                                       int(stride), cap, feature.data_ptr(), xyz.data_ptr(), ref.data_ptr(), direction.data_ptr(), ct.byref(m),
                                       ws.data_ptr(), ws.numel(), st), "nl_backproject_support")
    ws.record_stream(torch.cuda.current_stream(dev))
    M = int(m.value)
    return feature[:M], xyz[:M], ref[:M], direction[:M]


def get_rays(H: int, W: int, K: torch.Tensor, c2w: torch.Tensor, uv: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """conditional_nerf/utils.py:56-70 (whole grid -> (H,W,3) tensors) or model.py:687-700 (uv (R,2) pixel positions -> (R,3))."""
    lib = L.load()
    K, c2w = _f32(K, "K"), _f32(c2w, "pose")
    R = H * W if uv is None else uv.shape[0]
    o = torch.empty(R, 3, dtype=torch.float32, device=K.device)
    d = torch.empty(R, 3, dtype=torch.float32, device=K.device)
    uvp = None if uv is None else _f32(uv, "pts2d")
    st = torch.cuda.current_stream(K.device).cuda_stream
    L.check(lib.nl_get_rays(K.data_ptr(), c2w.data_ptr(), None if uvp is None else uvp.data_ptr(), int(H), int(W), R, o.data_ptr(), d.data_ptr(), st),
            "nl_get_rays")
    return (o.view(H, W, 3), d.view(H, W, 3)) if uv is None else (o, d)
