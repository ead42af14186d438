"""Time the exact KNN alone on a config's sample points (development: knock-out builds through NERFLOC_LIB).  python tools/knn_time.py [config] [reps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loc_amd.renderer import HipRenderer
from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
frame, rays = make_frame(cfg), None
rays = make_rays(cfg, frame)
r = HipRenderer(cfg.W, cfg.C, cfg.S_total, "bf16x3")
r.load_weights({k: torch.from_numpy(v) for k, v in make_weights(cfg).items()})
r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
o, d = t(rays["rays_o"]), t(rays["rays_d"])
lin = torch.linspace(0, 1, cfg.S, device=dev)
z = (cfg.near * (1 - lin) + cfg.far * lin).expand(o.shape[0], cfg.S)
xyz = (o[:, None, :] + d[:, None, :] * z[..., None]).reshape(-1, 3).contiguous()
for _ in range(3): r.knn(xyz, 8)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(reps): d2, idx = r.knn(xyz, 8)
e1.record(); torch.cuda.synchronize()
print(f"{cfg.name}: {xyz.shape[0]} queries, M = {frame['support_fine']['xyz'].shape[0]}: {e0.elapsed_time(e1) / reps:.3f} ms per search; mean 8th distance {float(d2[:, 7].sqrt().mean()):.4f}")
