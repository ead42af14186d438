"""Debug: cycle trace of one workgroup of unet_inner_kernel (needs a library whose unet_inner.hip was built with -DUI_TRACE=<pair index>; NERFLOC_LIB selects it).
python tools/unet_inner_trace.py [rays]"""
import ctypes as ct, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loc_amd import _lib as L
from nerf_loc_amd.renderer import HipRenderer
from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = CONFIGS["c2"]
frame, weights = make_frame(cfg), make_weights(cfg)
rays = make_rays(cfg, frame, R=R)
rnd = HipRenderer(cfg.W, cfg.C, cfg.S, "f16mx", device="cuda:0")
rnd.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
rnd.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
for _ in range(3):
    rnd.render_rays(rays["rays_o"], rays["rays_d"], frame["pose"][:3, 3])
torch.cuda.synchronize()
lib = ct.CDLL(L.LIB_PATH)
buf = (ct.c_ulonglong * 32)()
assert lib.nl_debug_unet_inner_trace(buf) == 0
t = np.array(buf[:], dtype=np.int64)
names = ["conv2", "conv3", "trans_conv3", "trans_conv2", "trans_conv1"]
print(f"R = {R}: workgroup total {t[21] - t[0]} cycles; staging c1 {t[1] - t[0]}")
for l, n in enumerate(names):
    b = 2 + 4 * l
    prev = t[1] if l == 0 else t[b - 1]
    print(f"  {n:12s} product {t[b] - prev:6d} | to barrier 1 (bias, sum, wait for the slowest wave) {t[b + 1] - t[b]:6d} | to barrier 2 (centred squares) {t[b + 2] - t[b + 1]:6d} | epilogue + barrier 3 {t[b + 3] - t[b + 2]:6d}")
