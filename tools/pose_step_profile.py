"""Where does a PoseOptimizer-sized gradient step spend its time?  Kernel-level totals (torch profiler) of one step of the gradient path with
the HIP point-branch backward, grouped by the autograd-level op that launched them.  python tools/pose_step_profile.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loc_amd import diff_render as dr
from nerf_loc_amd.renderer import HipRenderer
from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights

R = 512
cfg = CONFIGS["c2"]
dev = torch.device("cuda:0")
frame, weights, rays = make_frame(cfg), make_weights(cfg), make_rays(cfg, make_frame(cfg))
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
r = HipRenderer(cfg.W, cfg.C, cfg.S_total, "bf16x3")
r.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
p = {k: t(v) for k, v in weights.items()}
fr = {k: t(frame[k]) for k in ("topk_Ks", "topk_poses", "topk_images", "feat_fine_src", "vis_featmaps")}
fr.update({"near": float(cfg.near), "far": float(cfg.far), "support": {k: t(v) for k, v in frame["support_fine"].items()}})
sel = np.random.default_rng(0).choice(cfg.R, R, replace=False)
uv, K = t(rays["pixel_coordinates"][sel]), t(rays["K"])
lin = torch.linspace(0, 1, cfg.S, device=dev)
z = (cfg.near * (1 - lin) + cfg.far * lin).expand(R, cfg.S).contiguous()
pose = t(frame["pose"]).clone().requires_grad_(True)
tf = torch.randn(R, cfg.C, device=dev)
knn = lambda q: r.knn(q, 8)[1]


def step():
    o, d = dr.rays_from_pose(uv, K, pose)
    out = dr.render_rays_diff(p, fr, o, d, z, pose, knn, frozen_renderer=r)
    loss = torch.mean(((out["feat"] - tf) * out["mask"].unsqueeze(1)) ** 2)
    return torch.autograd.grad(loss, pose)[0]


for _ in range(3):
    step()
torch.cuda.synchronize()
# stage timing by synchronised sections of the forward (backward attributed through the profiler table below)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)[:int(os.environ.get("ROWS", "40"))]:
    print(f"{e.self_device_time_total / 3e3:8.3f} ms/step  {e.count // 3:4d} calls  {e.key[:120]}")
