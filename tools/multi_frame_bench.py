"""Several query frames per launch (nl_render_opts.ray_centers): F frames x R rays against one support frame, one call vs F calls.
python tools/multi_frame_bench.py [frames] [rays_per_frame]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loc_amd.renderer import HipRenderer
from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights
F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = CONFIGS["c2"]
frame, weights = make_frame(cfg), make_weights(cfg)
rays = make_rays(cfg, frame)
r = HipRenderer(cfg.W, cfg.C, cfg.S_total, "bf16x3")
r.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
dev = torch.device("cuda:0")
o = torch.from_numpy(rays["rays_o"][: F * R]).to(dev); d = torch.from_numpy(rays["rays_d"][: F * R]).to(dev)
centres = torch.from_numpy(frame["pose"][:3, 3]).to(dev) + 0.01 * torch.arange(F, device=dev).float()[:, None] * torch.ones(3, device=dev)
per_ray = centres.repeat_interleave(R, 0).contiguous()
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
one = timed(lambda: r.render_rays(o, d, per_ray))
sep = timed(lambda: [r.render_rays(o[i * R:(i + 1) * R], d[i * R:(i + 1) * R], centres[i]) for i in range(F)])
print(f"{F} query frames x {R} rays x {cfg.S} samples: one launch {one:.2f} ms ({one / F:.2f} per frame), {F} launches {sep:.2f} ms ({sep / F:.2f} per frame)")
