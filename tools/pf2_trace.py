"""Debug: per-region cycle trace of point_fused2_kernel (needs a library built with -DPF2_TRACE; NERFLOC_LIB selects it)."""
import ctypes as ct, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loc_amd import _lib as L
from nerf_loc_amd.renderer import HipRenderer
from nerf_loc_amd.synth import SceneConfig, make_frame, make_rays, make_weights
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
cfg = SceneConfig("c2", R=4096, S=128, W=256, V=10, H=256, Wimg=336, seed=2)
frame, weights = make_frame(cfg), make_weights(cfg)
rays = make_rays(cfg, frame)
rnd = HipRenderer(cfg.W, cfg.C, cfg.S, prec, device="cuda:0")
rnd.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
rnd.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
for _ in range(3):
    out = rnd.render_rays(rays["rays_o"], rays["rays_d"], frame["pose"][:3, 3])
torch.cuda.synchronize()
buf = (ct.c_ulonglong * 256)()
lib = L.load()
f = lib.nl_debug_pf2_trace
assert f(buf) == 0
t = np.array(buf[:], dtype=np.int64).reshape(4, 64)
NC = 32
for it in range(4):
    d = np.diff(t[it, :NC + 1])
    print(f"tile {it}: total {t[it, NC] - t[it, 0]} cycles; per region:", d.tolist())
    if it < 3:
        print(f"   gap to next tile (prologue etc.): {t[it + 1, 0] - t[it, NC]}")
