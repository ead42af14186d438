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
PREC = os.environ.get("PREC", "f16mx")
r = HipRenderer(cfg.W, cfg.C, cfg.S_total, PREC)
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

# ---- frames with DIFFERENT support sets: one renderer (frame tables, workspace, side stream) per support frame; the launch chains of
# the frames are independent, so they may run on separate caller streams and fill the chip together
def other_frame(i):
    c = cfg.replace(seed=cfg.seed + 100 + i)
    fr = make_frame(c)
    rr = HipRenderer(cfg.W, cfg.C, cfg.S_total, PREC, workspace_bytes=None)
    rr.packed = r.packed; rr._weights_loaded = True          # the same packed weights
    rr.set_frame(fr["topk_images"], fr["feat_fine_src"], fr["vis_featmaps"], fr["topk_Ks"], fr["topk_poses"], c.near, c.far, fr["support_fine"])
    ry = make_rays(c, fr)
    return rr, torch.from_numpy(ry["rays_o"][:R]).to(dev), torch.from_numpy(ry["rays_d"][:R]).to(dev), fr["pose"][:3, 3]
frames = [other_frame(i) for i in range(F)]
from nerf_loc_amd.renderer import render_rays_concurrent
streams = [torch.cuda.Stream() for _ in range(F)]
def serial():
    return [rr.render_rays(oo, dd, qc) for rr, oo, dd, qc in frames]
def concurrent():
    return render_rays_concurrent(frames, streams)
ts, tc = timed(serial), timed(concurrent)
a, b = serial(), concurrent()
torch.cuda.synchronize()
assert all(torch.equal(x[k], y[k]) for x, y in zip(a, b) for k in x), "concurrent rendering must be bit-identical to serial"
print(f"{F} DIFFERENT support frames x {R} rays: one stream {ts:.2f} ms ({ts / F:.2f} per frame), {F} streams {tc:.2f} ms ({tc / F:.2f} per frame)")
# ---- the same through ONE library call (round 4: nl_render_rays_multi — the fork / join over library-owned streams inside the C-ABI)
from nerf_loc_amd.renderer import render_rays_multi
def multi():
    return render_rays_multi(frames)
tm = timed(multi)
c = multi()
torch.cuda.synchronize()
assert all(torch.equal(x[k], y[k]) for x, y in zip(a, c) for k in x), "nl_render_rays_multi must be bit-identical to separate calls"
print(f"{F} DIFFERENT support frames x {R} rays: nl_render_rays_multi (one library call) {tm:.2f} ms ({tm / F:.2f} per frame)")
