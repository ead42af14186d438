import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from nerf_loc_amd.renderer import HipRenderer
from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights
cfg = CONFIGS["c2"]
frame, weights = make_frame(cfg), make_weights(cfg)
rays = make_rays(cfg, frame, seed_offset=1000)
r = HipRenderer(cfg.W, cfg.C, cfg.S_total, "f16mx")
r.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
lin = torch.linspace(0, 1, cfg.S)
qc = frame["pose"][:3, 3]
for R in (512, 1024, 4096):
    o, d = torch.from_numpy(rays["rays_o"][:R]).cuda(), torch.from_numpy(rays["rays_d"][:R]).cuda()
    z = (torch.tensor(cfg.near) * (1 - lin) + torch.tensor(cfg.far) * lin).expand(R, cfg.S).contiguous().cuda()
    for _ in range(5): r.render_rays(o, d, qc, z_vals=z)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 40
    for _ in range(n): r.render_rays(o, d, qc, z_vals=z)
    torch.cuda.synchronize()
    print(f"{os.environ.get('NERFLOC_LIB', 'default')[-18:]:18s} R={R}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per step")
