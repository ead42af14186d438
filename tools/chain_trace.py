"""Debug: per-chunk cycle trace of sample_chain_kernel (needs a library whose tgemm.hip was built with -DCHAIN_TRACE; NERFLOC_LIB selects
it).  Per chunk slot: wait = cycles in s_waitcnt + s_barrier, run = cycles from the barrier to the end of the slot."""
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
buf = (ct.c_ulonglong * (2 * 4 * 96))()
assert L.load().nl_debug_chain_trace(buf) == 0
t = np.array(buf[:], dtype=np.int64).reshape(2, 4, 32, 3)
for k, (name, nch) in enumerate((("chain", 24), ("query", 12))):
    for it in (1, 2):
        x = t[k, it, :nch]
        if x[0, 0] == 0: continue
        print(f"{name} tile {it}: total {x[nch - 1, 2] - x[0, 0]} cycles\n  wait {(x[:, 1] - x[:, 0]).tolist()}\n  run  {(x[:, 2] - x[:, 1]).tolist()}")
