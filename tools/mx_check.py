#!/usr/bin/env python3
"""f16mx (fp16 hi.hi + two MX-FP8 cross terms in the fused neural-point kernel) against bf16x3 and fp32 on the goldens + config 2: errors and times."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loc_amd.renderer import HipRenderer
from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights
from tests.golden_cases import CASES, build_case
from tests.util import load_golden, rel_err

def renderer(case, prec):
    cfg, fr = case["cfg"], case["frame"]
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, prec)
    r.load_weights({k: torch.from_numpy(v) for k, v in case["weights"].items()})
    r.set_frame(fr["topk_images"], fr["feat_fine_src"], fr["vis_featmaps"], fr["topk_Ks"], fr["topk_poses"], cfg.near, cfg.far, fr["support_fine"])
    return r

def z_of(cfg, R):
    lin = torch.linspace(0, 1, cfg.S)
    return (torch.tensor(cfg.near) * (1 - lin) + torch.tensor(cfg.far) * lin).expand(R, cfg.S).contiguous()

for name in ("w128s64", "w256s128"):
    case = build_case(name); cfg = case["cfg"]; g = load_golden(name)
    for prec in ("bf16x3", "f16mx"):
        r = renderer(case, prec)
        out = r.render_rays(case["rays"]["rays_o"], case["rays"]["rays_d"], case["frame"]["pose"][:3, 3], z_vals=z_of(cfg, cfg.R), white_bkgd=cfg.white_bkgd, intermediates=True)
        errs = {k: rel_err(out[k].cpu().numpy(), g[k]) for k in ("rgb", "depth", "weights", "depth_uncertainty", "feat", "sigma")}
        rows = g["rows"] if "rows" in g else slice(None)
        errs["feature_agg"] = rel_err(out["feature_agg"].cpu().numpy()[rows], g["feature_agg"])
        print(name, prec, {k: f"{v:.1e}" for k, v in errs.items()}, flush=True)

cfg = CONFIGS["c2"]
frame, weights = make_frame(cfg), make_weights(cfg)
rays = make_rays(cfg, frame, seed_offset=1000)
case = {"cfg": cfg, "frame": frame, "weights": weights}
outs = {}
for prec in ("fp32", "bf16x3", "f16mx"):
    r = renderer(case, prec)
    o, d = torch.from_numpy(rays["rays_o"]).cuda(), torch.from_numpy(rays["rays_d"]).cuda()
    z = z_of(cfg, cfg.R).cuda()
    qc = frame["pose"][:3, 3]
    for _ in range(3): out = r.render_rays(o, d, qc, z_vals=z)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = r.render_rays(o, d, qc, z_vals=z)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 100
    outs[prec] = {k: out[k].float().cpu().numpy() for k in ("rgb", "depth", "weights", "depth_uncertainty", "feat")}
    print(f"c2 {prec}: {ms:.2f} ms per 4096 x 128 batch = {4096 / ms:.0f} k rays/s", flush=True)
    del r
for prec in ("bf16x3", "f16mx"):
    print("c2", prec, "vs fp32:", {k: f"{rel_err(outs[prec][k], outs['fp32'][k]):.1e}" for k in outs[prec]})
