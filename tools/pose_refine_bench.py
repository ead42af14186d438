"""Time one PoseOptimizer step (pose_optimizer.py:131-160: 512 rays, forward + backward to the pose) through the drop-in module's
gradient path on the c2 scene, next to the HIP forward alone.  python tools/pose_refine_bench.py [rays] [steps] [precision]"""
import gc, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loc_amd import diff_render as dr
from nerf_loc_amd.renderer import HipRenderer
from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights

R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = CONFIGS["c2"]
dev = torch.device("cuda:0")
frame, weights, rays = make_frame(cfg), make_weights(cfg), make_rays(cfg, make_frame(cfg))
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
precision = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
r = HipRenderer(cfg.W, cfg.C, cfg.S_total, precision)
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
tf = torch.randn(R, cfg.C, generator=torch.Generator().manual_seed(0)).to(dev)
knn = lambda q: r.knn(q, 8)[1]

def step(frozen):
    o, d = dr.rays_from_pose(uv, K, pose)
    out = dr.render_rays_diff(p, fr, o, d, z, pose, knn, frozen_renderer=r if frozen else None)
    loss = torch.mean(((out["feat"] - tf) * out["mask"].unsqueeze(1)) ** 2)
    g, = torch.autograd.grad(loss, pose)
    return g

res = {}
for frozen in (False, True):   # False: everything eager (round 2); True: the whole path as one library node (diff_render.RenderFn)
    torch.cuda.reset_peak_memory_stats()
    for _ in range(2): step(frozen)
    gc.collect()   # (no generation-2 pass of the interpreter inside the timed steps: tools/train_step_bench.py)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): g = step(frozen)
    t_issue = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    res[frozen] = ((time.perf_counter() - t0) / steps, torch.cuda.max_memory_allocated() / 2**30, g.clone(), t_issue)
dt = res[False][0]
print(f"{R} rays x {cfg.S} samples: gradient step through the library (RenderFn: nl_render_rays_forward_keep / nl_render_rays_backward_kept) {res[True][0]*1e3:.1f} ms (CPU issue time {res[True][3]*1e3:.1f} ms: the pair is replayed as two HIP graphs), peak memory {res[True][1]:.1f} GiB; "
      f"relative difference of dL/dpose to the eager graph {float((res[True][2] - res[False][2]).abs().max() / res[False][2].abs().max()):.2e}")
with torch.no_grad():
    o, d = dr.rays_from_pose(uv, K, pose)
    for _ in range(3): r.render_rays(o, d, pose[:3, 3])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): r.render_rays(o, d, pose[:3, 3])
    torch.cuda.synchronize(); df = (time.perf_counter() - t0) / steps
print(f"{R} rays x {cfg.S} samples: gradient step (eager fp32 autograd + HIP KNN) {dt*1e3:.1f} ms, peak memory {res[False][1]:.1f} GiB; "
      f"HIP forward alone {df*1e3:.2f} ms; |dL/dpose| max {float(g.abs().max()):.3e}")
