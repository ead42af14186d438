"""VERDICT r5 item 1c, measured instead of estimated: what does the exact KNN get when it runs BESIDE the fused neural-point kernel?

The two-chunk pipeline the verdict proposes (KNN of chunk k + 1 on the side stream under point_fused2 of chunk k) only pays if a KNN wave that shares a SIMD with the
persistent matrix kernel makes progress there.  This tool measures exactly that without building the pipeline: a second stream issues half-batch KNN searches back to
back while the first renders config-2 steps; run under `rocprofv3 --kernel-trace` the trace gives every KNN launch's duration and start, so the launches that ran
under point_fused2_kernel can be compared with the ones that ran beside the other kernels and with the search alone — and point_fused2's own slow-down is in the
same trace.  python tools/knn_under_pf2.py [steps]   (summary: tools/knn_under_pf2_summary.py results.db)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loc_amd.renderer import HipRenderer
from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfg = CONFIGS["c2"]
dev = torch.device("cuda:0")
frame, weights = make_frame(cfg), make_weights(cfg)
rays = make_rays(cfg, frame, seed_offset=1000)
r = HipRenderer(cfg.W, cfg.C, cfg.S_total, "f16mx")
r.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
o, d = torch.from_numpy(rays["rays_o"]).to(dev), torch.from_numpy(rays["rays_d"]).to(dev)
lin = torch.linspace(0, 1, cfg.S, device=dev)
z = (cfg.near * (1 - lin) + cfg.far * lin).expand(cfg.R, cfg.S).contiguous()
qc = frame["pose"][:3, 3]
xyz_half = (o[:2048, None, :] + d[:2048, None, :] * z[:2048, :, None]).reshape(-1, 3).contiguous()   # the second chunk's 262 144 queries

def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

render = lambda: r.render_rays(o, d, qc, z_vals=z, side_stream=False)
for _ in range(3): render(); r.knn(xyz_half, 8)
print(f"alone: render step {timed(render, steps):.3f} ms, half-batch KNN {timed(lambda: r.knn(xyz_half, 8), steps):.3f} ms", flush=True)
side = torch.cuda.Stream(dev)
stop_after = steps
torch.cuda.synchronize()
t0 = time.perf_counter()
n_knn = 0
for _ in range(stop_after):
    render()
    with torch.cuda.stream(side):
        for _ in range(12):   # ~12 x 0.4-0.7 ms of searches per step keep the side stream busy throughout the step
            r.knn(xyz_half, 8); n_knn += 1
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) * 1e3
print(f"together: {stop_after} render steps + {n_knn} half-batch searches in {dt:.2f} ms = {dt / stop_after:.3f} ms per step-with-12-searches", flush=True)
