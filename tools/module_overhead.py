#!/usr/bin/env python3
"""Where the drop-in module's render_rays / render_image time goes beside the raw library call (VERDICT r5 'What's weak' 9: 475 k rays/s through the
module against 530 k raw).  Times, at BASELINE config 2 in the headline precision: the raw HipRenderer call in a pipelined loop and call-by-call,
the module's render_rays call-by-call with and without the precision guard, render_image, and the raw renderer on the SAME image rays in one call
(so that what is left between the last two is host work, and what is left between image rays and the bench's random rays is the data)."""
import os, sys, time
from types import SimpleNamespace as NS
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights, make_depth_fusion_weights, add_setup_inputs
from nerf_loc_amd.conditional_nerf import ConditionalNeRF, get_rays

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
prec = sys.argv[2] if len(sys.argv) > 2 else "f16mx"
args = NS(multires=10, multires_views=4, i_embed=0, backbone2d_fpn_dim=cfg.C, model_3d_hidden_dim=cfg.W,
          render=NS(N_samples=cfg.S, N_importance=cfg.N_importance, N_rand=1024, chunk=4096, lindisp=False, white_bkgd=False,
                    use_render_uncertainty=True, render_feature=True),
          use_scene_coord_memorization=False, matcher_hidden_dim=192, use_depth_supervision=False, matching=NS(fine_num_3d_keypoints=1024))
frame = add_setup_inputs(cfg, make_frame(cfg))
rays = make_rays(cfg, frame)
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
net = ConditionalNeRF(args, precision=prec).to(dev).eval()
w = dict(make_weights(cfg)); w.update(make_depth_fusion_weights(cfg.seed))
net.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
data = {k: torch.from_numpy(frame[k]).to(dev) for k in ("topk_images", "topk_depths", "topk_Ks", "topk_poses", "feat_fine_src", "feat_coarse_src", "depth_range", "K", "pose")}
data.update({"embedding_a": None, "H": frame["H"], "W": frame["W"], "stride_fine": 4, "stride_coarse": 8})
rd = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in rays.items()}

def sync(): torch.cuda.synchronize()
def each(fn, n=20):
    """call-by-call: a sync after every call (what a caller that reads the result sees)"""
    fn(); sync(); ts = []
    for _ in range(n):
        sync(); t0 = time.perf_counter(); fn(); sync(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2] * 1e3
def piped(fn, n=20):
    fn(); sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    sync(); return (time.perf_counter() - t0) / n * 1e3
def host_only(fn, n=20):
    """host time of the call itself (returns before the device is done unless the call syncs)"""
    fn(); sync(); ts = []
    for _ in range(n):
        sync(); t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0); sync()
    ts.sort(); return ts[len(ts) // 2] * 1e3

net.render_rays(data, rd); sync()
r = net._renderer("fine")
o, d = rd["rays_o"], rd["rays_d"]
near, far = rd["depth_range"]
z = net.sample_depths(cfg.S, near, far).expand(o.shape[0], cfg.S).contiguous()
c = data["pose"][:3, 3]
raw = lambda: r.render_rays(o, d, c, z_vals=z, want_feat=True)
rawg = lambda: r.render_rays(o, d, c, z_vals=z, want_feat=True, precision_guard=True)
mod = lambda: net.render_rays(data, rd)
print(f"{cfg.name} {prec}, {o.shape[0]} rays per call, ms per call (median of 20):")
print(f"  raw HipRenderer.render_rays      pipelined {piped(raw):.3f}   call-by-call {each(raw):.3f}   host part {host_only(raw):.3f}")
print(f"  raw + NL_RENDER_PRECISION_GUARD  pipelined {piped(rawg):.3f}   call-by-call {each(rawg):.3f}   host part {host_only(rawg):.3f}")
net.precision_guard = True
print(f"  module render_rays, guard on     pipelined {piped(mod):.3f}   call-by-call {each(mod):.3f}   host part {host_only(mod):.3f}")
net.precision_guard = False
print(f"  module render_rays, guard off    pipelined {piped(mod):.3f}   call-by-call {each(mod):.3f}   host part {host_only(mod):.3f}")
net.precision_guard = True
# the image
H, Wd = data["H"], data["W"]
io, idr = get_rays(H, Wd, data["K"], data["pose"])
io, idr = io.reshape(-1, 3).contiguous(), idr.reshape(-1, 3).contiguous()
R = io.shape[0]
zi = net.sample_depths(cfg.S, *data["depth_range"][0]).expand(R, cfg.S).contiguous()
img_raw = lambda: r.render_rays(io, idr, c, z_vals=zi, want_feat=True)
img_rawg = lambda: r.render_rays(io, idr, c, z_vals=zi, want_feat=True, precision_guard=True)
img_mod = lambda: net.render_image(data)
perm = torch.randperm(R, device=dev)
io_p, id_p = io[perm].contiguous(), idr[perm].contiguous()
img_perm = lambda: r.render_rays(io_p, id_p, c, z_vals=zi, want_feat=True)
n4 = (R // 4096) * 4096
def img_chunks():
    for s in range(0, R, 4096):
        r.render_rays(io[s:s + 4096], idr[s:s + 4096], c, z_vals=zi[s:s + 4096], want_feat=True)
t_bench = piped(raw)
print(f"image {H}x{Wd} = {R} rays, ms per image (median of 5) and k rays/s; the bench's random rays would take {t_bench * R / o.shape[0]:.1f} ms:")
for name, fn in (("raw, one call", img_raw), ("raw, one call, guard flag", img_rawg), ("raw, 4096-ray calls back to back", img_chunks), ("raw, one call, rays permuted", img_perm),
                 ("module render_image, guard on", img_mod)):
    t = each(fn, 5)
    print(f"  {name:36s} {t:8.2f} ms  {R / t:7.1f} k rays/s")
net.precision_guard = False
t = each(img_mod, 5); print(f"  {'module render_image, guard off':36s} {t:8.2f} ms  {R / t:7.1f} k rays/s")
