#!/usr/bin/env python3
"""forward_fuzz.py's checks (library vs the CPU oracle and vs fp64, three precisions, a random option per case) on the shapes that take the fused
feat_mlp.0 + compositing kernel (tgemm.hip: feat_comp_mx_kernel — W = 256, whole 32-row tiles, rays whose waves divide a workgroup of 8 or 6) and on their
neighbours that must fall back (S / 32 = 5, 7; sample counts that are not multiples of 32; hierarchical totals): python tools/feat_comp_fuzz.py [ncases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import forward_fuzz

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
worst = 0.0
for case in range(n):
    S_total = int(rng.choice([32, 64, 96, 128, 128, 160, 192, 224, 256, 40, 72]))
    NI = int(rng.choice([0, 0, 0, 32, 64])) if S_total >= 96 and S_total % 32 == 0 else 0
    S = S_total - NI
    V = int(rng.integers(1, 13)); H, Wimg = int(rng.integers(24, 73)), int(rng.integers(24, 89)); R = int(rng.integers(1, 24))
    white = int(rng.random() < 0.3); mcut = int(rng.integers(1, 8)) if rng.random() < 0.15 else 0
    os.environ["FORCE"] = f"256,{S},{NI},{V},192,{H},{Wimg},{R},{white},{mcut}"
    os.environ["FORCE_OPT"] = str(int(rng.integers(0, 4)))
    print(f"[{case}] FORCE={os.environ['FORCE']} opt={os.environ['FORCE_OPT']}", flush=True)
    w = forward_fuzz.run(1, 500 + case, verbose=False)
    worst = max(worst, w)
print(f"feat_comp fuzz: {n} scenes, worst error vs fp64 {worst:.2e}")
