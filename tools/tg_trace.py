"""Debug: cycle trace of the conv_out launch of tgemm_kernel (needs a library whose tgemm.hip was built with -DTG_TRACE; NERFLOC_LIB selects it):
cycles per 32-k chunk of block 0 / wave 0 and of the fused LayerNorm epilogue's phases."""
import ctypes as ct, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loc_amd import _lib as L
from nerf_loc_amd.renderer import HipRenderer
from nerf_loc_amd.synth import SceneConfig, make_frame, make_rays, make_weights
cfg = SceneConfig("c2", R=4096, S=128, W=256, V=10, H=256, Wimg=336, seed=2)
frame, weights = make_frame(cfg), make_weights(cfg)
rays = make_rays(cfg, frame)
rnd = HipRenderer(cfg.W, cfg.C, cfg.S, "bf16x3", device="cuda:0")
rnd.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
rnd.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
for _ in range(3):
    rnd.render_rays(rays["rays_o"], rays["rays_d"], frame["pose"][:3, 3])
torch.cuda.synchronize()
buf = (ct.c_ulonglong * 64)()
assert L.load().nl_debug_tg_trace(buf) == 0
t = np.array(buf[:], dtype=np.int64)
ch = t[:27]
print("chunks:", np.diff(ch).tolist(), "| last chunk + tail to epilogue start:", int(t[48] - ch[-1]))
print("epilogue: statistics", int(t[49] - t[48]), "normalise + ELU + density dot", int(t[50] - t[49]), "| total kernel wave time", int(t[50] - t[0]))
