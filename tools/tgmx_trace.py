"""Debug: cycle trace of tgemm_mx_kernel (conv_out in f16mx) — needs a library whose tgemm.hip was built with -DTG_TRACE (NERFLOC_LIB selects it): the phases of
slabs 4 .. 9 of block 0 for the two waves of SIMD 0 (waves 0 and 4): operand conversion | issue of the next slab's loads | matrix phase | wait for the loads | barrier."""
import ctypes as ct, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loc_amd import _lib as L
from nerf_loc_amd.renderer import HipRenderer
from nerf_loc_amd.synth import SceneConfig, make_frame, make_rays, make_weights
cfg = SceneConfig("c2", R=4096, S=128, W=256, V=10, H=256, Wimg=336, seed=2)
frame, weights = make_frame(cfg), make_weights(cfg)
rays = make_rays(cfg, frame)
rnd = HipRenderer(cfg.W, cfg.C, cfg.S, "f16mx", device="cuda:0")
rnd.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
rnd.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
for _ in range(3):
    rnd.render_rays(rays["rays_o"], rays["rays_d"], frame["pose"][:3, 3])
torch.cuda.synchronize()
buf = (ct.c_ulonglong * 64)()
lib = L.load()
lib.nl_debug_tg_trace.argtypes = [ct.c_void_p]
assert lib.nl_debug_tg_trace(buf) == 0
t = np.array(buf[:], dtype=np.int64)
if len(sys.argv) > 1 and sys.argv[1] == "conv1":   # tgemm_conv1_kernel: chunks 8 .. 13 of block 0, wave 0
    tt = t[:30].reshape(6, 5)
    print("conv1, wave 0: per chunk [wait vmcnt, barrier, operand + issue of chunk g + 3's loads, 12 matrix instructions, -> next chunk] and the chunk's total")
    for i in range(5):
        ph = [int(tt[i, 1] - tt[i, 0]), int(tt[i, 2] - tt[i, 1]), int(tt[i, 3] - tt[i, 2]), int(tt[i, 4] - tt[i, 3]), int(tt[i + 1, 0] - tt[i, 4])]
        print(f"  chunk {8 + i}: {ph}  total {int(tt[i + 1, 0] - tt[i, 0])}")
    sys.exit(0)
for w, base in ((0, 0), (4, 32)):
    tt = t[base:base + 30].reshape(6, 5)
    print(f"wave {w}: per slab [convert, issue loads, matrix phase, wait vmcnt, barrier -> next slab start] and the slab's total")
    for i in range(5):
        ph = [int(tt[i, 1] - tt[i, 0]), int(tt[i, 2] - tt[i, 1]), int(tt[i, 3] - tt[i, 2]), int(tt[i, 4] - tt[i, 3]), int(tt[i + 1, 0] - tt[i, 4])]
        print(f"  slab {4 + i}: {ph}  total {int(tt[i + 1, 0] - tt[i, 0])}")
