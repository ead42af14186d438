#!/usr/bin/env python3
"""Run only the ray U-Net stage (nl_ray_unet) on a c2-sized batch — a harness for profiling the conv GEMMs in isolation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_loc_amd.synth import CONFIGS, make_weights
from nerf_loc_amd.renderer import HipRenderer

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rnd = HipRenderer(cfg.W, cfg.C, cfg.S_total, "bf16x3")
rnd.load_weights({k: torch.from_numpy(v) for k, v in make_weights(cfg).items()})
x = torch.randn(R * cfg.S_total, cfg.W, device="cuda")
for _ in range(2): y = rnd.ray_unet(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters): y = rnd.ray_unet(x)
torch.cuda.synchronize()
print("unet ms/call", (time.perf_counter() - t0) / iters * 1e3, float(y.abs().mean()))
