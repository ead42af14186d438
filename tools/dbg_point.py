"""Debug: fused neural-point kernel v2 against v1 (same inputs, same process) through nl_point_mlp."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loc_amd.renderer import HipRenderer
from nerf_loc_amd.synth import SceneConfig, make_frame, make_rays, make_weights

W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
cfg = SceneConfig("dbg", R=int(sys.argv[3]) if len(sys.argv) > 3 else 64, S=32, W=W, V=4, H=48, Wimg=64, seed=5)
frame, weights = make_frame(cfg), make_weights(cfg)
rays = make_rays(cfg, frame)
rnd = HipRenderer(cfg.W, cfg.C, cfg.S, prec, device="cuda:0")
rnd.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
rnd.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
z, xyz = rnd.sample_points(rays["rays_o"], rays["rays_d"])
mv, _, _, _ = rnd.mv_aggregate(xyz, frame["pose"][:3, 3], want_raw=False)
dirs = torch.from_numpy(rays["rays_d"]).repeat_interleave(cfg.S, 0)
res = {}
for v in ("v1", "v2"):
    if v == "v1": os.environ["NERFLOC_POINT_V1"] = "1"
    else: os.environ.pop("NERFLOC_POINT_V1", None)
    fa, d2, idx = rnd.point_mlp(xyz, dirs, mv, K=8)
    torch.cuda.synchronize()
    res[v] = fa.cpu().numpy()
a, b = res["v1"], res["v2"]
print("nan v1", np.isnan(a).sum(), "nan v2", np.isnan(b).sum(), "shape", a.shape)
err = np.abs(a - b)
print("max abs err", np.nanmax(err), "ref max", np.abs(a).max(), "rel", np.nanmax(err) / np.abs(a).max())
bad = np.argwhere(~(err < 1e-3 * np.abs(a).max()))
print("bad entries", len(bad), bad[:10].tolist())
if len(bad):
    rows = np.unique(bad[:, 0]); print("bad rows", len(rows), rows[:40].tolist())
    cols = np.unique(bad[:, 1]); print("bad cols", len(cols), cols[:64].tolist())
