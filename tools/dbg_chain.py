"""Debug: the chain kernels (default) against the separate launches (NERFLOC_NO_CHAIN=1); two subprocesses, compares intermediates."""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from nerf_loc_amd.renderer import HipRenderer
    from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights
    cfg = CONFIGS["c3"]
    fr, w = make_frame(cfg), make_weights(cfg)
    rays = make_rays(cfg, fr)
    n = int(sys.argv[3])
    sel = np.arange(0, cfg.R, cfg.R // n)[:n]
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, sys.argv[4])
    r.load_weights({k: torch.from_numpy(v) for k, v in w.items()})
    r.set_frame(fr["topk_images"], fr["feat_fine_src"], fr["vis_featmaps"], fr["topk_Ks"], fr["topk_poses"], cfg.near, cfg.far, fr["support_fine"])
    out = r.render_rays(rays["rays_o"][sel], rays["rays_d"][sel], fr["pose"][:3, 3], intermediates=True)
    np.savez(sys.argv[2], **{k: v.float().cpu().numpy() for k, v in out.items()})
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "64"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
for tag, env in (("keep", {"NERFLOC_NO_CHAIN": "1"}), ("new", {})):
    e = dict(os.environ); e.update(env)
    subprocess.check_call([sys.executable, __file__, "child", f"/tmp/dbg_{tag}.npz", n, prec], env=e)
a, b = np.load("/tmp/dbg_keep.npz"), np.load("/tmp/dbg_new.npz")
for k in a.files:
    x, y = a[k], b[k]
    err = np.abs(x - y)
    print(k, x.shape, "max abs", err.max(), "ref max", np.abs(x).max(), "nan", np.isnan(y).sum())
    if k == "feature_agg":
        bad = np.argwhere(err > 1e-3 * np.abs(x).max())
        rows = np.unique(bad[:, 0]); cols = np.unique(bad[:, 1])
        print("  bad rows", len(rows), rows[:40].tolist(), "\n  bad cols", len(cols), cols[:40].tolist())
