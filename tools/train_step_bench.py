"""Time the render part of one training step (compute_render_loss, model.py:641-685: N_rand = 1024 rays x 64 samples, forward + backward to
every parameter of the ray path, the feature maps, the DepthFusionNet maps and the support table) on the gradient path of the drop-in, with the
library's training nodes (diff_render.*TrainFn) against the all-eager fp32 graph.  python tools/train_step_bench.py [rays] [samples] [W] [steps]"""
import gc, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loc_amd import diff_render as dr
from nerf_loc_amd.renderer import HipRenderer
from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W = int(sys.argv[3]) if len(sys.argv) > 3 else 128
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
cfg = CONFIGS["c2"].replace(S=S, W=W)
dev = torch.device("cuda:0")
frame, weights = make_frame(cfg), make_weights(cfg)
rays = make_rays(cfg, frame)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
r = HipRenderer(cfg.W, cfg.C, cfg.S_total, "bf16x3")
r.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
p = {k: t(v).requires_grad_(True) for k, v in weights.items()}
fr = {k: t(frame[k]) for k in ("topk_Ks", "topk_poses", "topk_images", "feat_fine_src", "vis_featmaps")}
fr["feat_fine_src"].requires_grad_(True); fr["vis_featmaps"].requires_grad_(True)
sp = {k: t(v) for k, v in frame["support_fine"].items()}
sp["feature"].requires_grad_(True); sp["confidence"].requires_grad_(True)
fr.update({"near": float(cfg.near), "far": float(cfg.far), "support": sp})
sel = np.random.default_rng(0).choice(cfg.R, R, replace=False)
o, d = t(rays["rays_o"][sel]), t(rays["rays_d"][sel])
lin = torch.linspace(0, 1, cfg.S, device=dev)
z = (cfg.near * (1 - lin) + cfg.far * lin).expand(R, cfg.S).contiguous()
pose = t(frame["pose"])
gen = torch.Generator().manual_seed(0)
t_rgb, t_feat = torch.rand(R, 3, generator=gen).to(dev), torch.randn(R, cfg.C, generator=gen).to(dev)
knn = lambda q: r.knn(q, 8)[1]
leaves = [v for v in p.values()] + [fr["feat_fine_src"], fr["vis_featmaps"], sp["feature"], sp["confidence"]]


def step(hip):
    out = dr.render_rays_diff(p, fr, o, d, z, pose, knn, beta=True, train_renderer=r if hip else None)   # render.use_render_uncertainty: the reference configs' default
    m = out["mask"].unsqueeze(1).float()
    b = out["beta"].unsqueeze(1)
    loss = torch.mean(((out["rgb"] - t_rgb) * m) ** 2 / (2 * b ** 2)) + torch.mean(torch.log(b)) + 0.1 * torch.mean(((out["feat"] - t_feat) * m) ** 2)
    gs = torch.autograd.grad(loss, leaves, allow_unused=True)
    return loss.detach(), gs


res = {}
for hip in (False, True):
    torch.cuda.reset_peak_memory_stats()
    for _ in range(2): step(hip)
    # one full collection now: the eager pass leaves ~10^5 objects behind, and the generation-2 pass they eventually trigger (40-50 ms, the GPU idle
    # meanwhile) used to land inside the five timed steps of the library's pass whenever the allocation count happened to cross the threshold there —
    # +8 ms per step and a CPU "issue time" of 10-14 ms that earlier docs of this repo blamed on the host's load
    gc.collect()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): l, gs = step(hip)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    res[hip] = ((time.perf_counter() - t0) / steps, torch.cuda.max_memory_allocated() / 2**30, l, gs, t_issue / steps)
names = list(p.keys()) + ["feat_fine_src", "vis_featmaps", "support.feature", "support.confidence"]
gmax = max(float(g.abs().max()) for g in res[False][3] if g is not None)
worst = ("", 0.0)
for n, a, b in zip(names, res[True][3], res[False][3]):
    if a is None or b is None:
        continue
    e = float((a - b).abs().max() / max(float(b.abs().max()), 1e-5 * gmax))
    if e > worst[1]: worst = (n, e)
print(f"{R} rays x {S} samples, W = {W}: training step (render forward + backward to all parameters) eager fp32 graph {res[False][0]*1e3:.1f} ms / "
      f"{res[False][1]:.1f} GiB; with the library's training nodes {res[True][0]*1e3:.1f} ms / {res[True][1]:.1f} GiB; "
      f"(CPU issue time {res[True][4]*1e3:.1f} ms); loss {float(res[False][2]):.6f} vs {float(res[True][2]):.6f}; largest per-tensor gradient difference {worst[1]:.2e} ({worst[0]})")
if os.environ.get("PROFILE"):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(3): step(True)
        torch.cuda.synchronize()
    rows = sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)[:28]
    for e in rows:
        print(f"{e.self_device_time_total / 3e3:8.3f} ms/step  {e.count // 3:4d} calls  {e.key[:110]}")
