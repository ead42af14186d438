#!/usr/bin/env python3
"""Check that a warm nl_render_rays call can be captured into a HIP graph by the caller (torch.cuda.CUDAGraph) and replayed:
small batches (c1: 256 rays x 32 samples, ~40 launches) are launch-bound otherwise."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights
from nerf_loc_amd.renderer import HipRenderer

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c1"]
frame = make_frame(cfg); rays = make_rays(cfg, frame); w = make_weights(cfg)
r = HipRenderer(cfg.W, cfg.C, cfg.S_total, "bf16x3")
r.load_weights({k: torch.from_numpy(v) for k, v in w.items()})
r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
o = torch.from_numpy(rays["rays_o"]).cuda(); d = torch.from_numpy(rays["rays_d"]).cuda()
t = torch.linspace(0, 1, cfg.S_total)
z = (cfg.near * (1 - t) + cfg.far * t).expand(cfg.R, cfg.S_total).contiguous().cuda()
qc = frame["pose"][:3, 3]
ref = r.render_rays(o, d, qc, z_vals=z)          # warm: per-frame tables, workspace
torch.cuda.synchronize()
def bench(fn, n=50):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
eager = bench(lambda: r.render_rays(o, d, qc, z_vals=z))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    out = r.render_rays(o, d, qc, z_vals=z)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        out = r.render_rays(o, d, qc, z_vals=z)
torch.cuda.synchronize()
graph = bench(g.replay)
ok = all(torch.equal(out[k], ref[k]) for k in ref)
print(f"{cfg.name}: eager {eager:.3f} ms, graph replay {graph:.3f} ms, identical outputs: {ok}")
