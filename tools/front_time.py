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
o, d = torch.from_numpy(rays["rays_o"]).cuda(), torch.from_numpy(rays["rays_d"]).cuda()
lin = torch.linspace(0, 1, cfg.S)
z = (torch.tensor(cfg.near) * (1 - lin) + torch.tensor(cfg.far) * lin).expand(cfg.R, cfg.S).contiguous().cuda()
qc = frame["pose"][:3, 3]
for ss in (False, True):
    for _ in range(3): r.render_rays(o, d, qc, z_vals=z, side_stream=ss)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): r.render_rays(o, d, qc, z_vals=z, side_stream=ss)
    torch.cuda.synchronize()
    print(f"{os.environ.get('NERFLOC_LIB', 'default')[-22:]} side_stream={ss}: {(time.perf_counter() - t0) * 50:.3f} ms per step")
