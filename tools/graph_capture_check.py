#!/usr/bin/env python3
"""Check that a warm nl_render_rays call can be captured into a HIP graph by the caller (torch.cuda.CUDAGraph) and replayed, and what the replay buys:
small batches (c1: 256 rays x 32 samples; a 512-ray shard of config 2 on 8 GPUs) pay ~10 us of ramp / gap per launch of the 18-kernel dependency chain.
  python tools/graph_capture_check.py [config] [rays ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights
from nerf_loc_amd.renderer import HipRenderer

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c1"]
sizes = [int(x) for x in sys.argv[2:]] or [cfg.R]
frame = make_frame(cfg); rays = make_rays(cfg, frame, R=max(sizes)); w = make_weights(cfg)
r = HipRenderer(cfg.W, cfg.C, cfg.S_total, os.environ.get("PREC", "f16mx"))
r.load_weights({k: torch.from_numpy(v) for k, v in w.items()})
r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
t = torch.linspace(0, 1, cfg.S_total)
qc = frame["pose"][:3, 3]
def bench(fn, n=50):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for R in sizes:
    o = torch.from_numpy(rays["rays_o"][:R]).cuda(); d = torch.from_numpy(rays["rays_d"][:R]).cuda()
    z = (cfg.near * (1 - t) + cfg.far * t).expand(R, cfg.S_total).contiguous().cuda()
    ref = r.render_rays(o, d, qc, z_vals=z)          # warm: per-frame tables, workspace
    torch.cuda.synchronize()
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
    print(f"{cfg.name}: {R} rays: eager {eager:.3f} ms, graph replay {graph:.3f} ms, identical outputs: {ok}", flush=True)
