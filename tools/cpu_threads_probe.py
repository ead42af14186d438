import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights
from oracle import render_oracle as orc
cfg = CONFIGS["c2"]; frame = make_frame(cfg); rays = make_rays(cfg, frame, R=64); w = make_weights(cfg)
p = {k: torch.from_numpy(v) for k, v in w.items()}; fr = orc.to_torch(frame); rr = orc.to_torch(rays)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    for rep in range(2):
        t0 = time.perf_counter(); tm = {}
        with torch.no_grad(): orc.render_rays(p, fr, rr, cfg.S, knn_threads=min(os.cpu_count(), 64), timers=tm)
        dt = time.perf_counter() - t0
    print(nt, f"{64/dt:.2f} rays/s", {k: round(v, 2) for k, v in tm.items()}, flush=True)
