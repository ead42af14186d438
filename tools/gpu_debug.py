#!/usr/bin/env python3
"""GPU-box debugging aid: runs golden cases through the HIP path, prints per-output errors vs golden + oracle."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nerf_loc_amd.renderer import HipRenderer
from tests.golden_cases import CASES, build_case
from tests.util import load_golden, rel_err

def run(name, precision):
    cfg, full = CASES[name]
    case = build_case(name)
    g = load_golden(name)
    fr, rays = case["frame"], case["rays"]
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, precision)
    r.load_weights({k: torch.from_numpy(v) for k, v in case["weights"].items()})
    r.set_frame(fr["topk_images"], fr["feat_fine_src"], fr["vis_featmaps"], fr["topk_Ks"], fr["topk_poses"], cfg.near, cfg.far, fr["support_fine"])
    out = r.render_rays(rays["rays_o"], rays["rays_d"], fr["pose"][:3, 3], white_bkgd=cfg.white_bkgd, intermediates=True)
    torch.cuda.synchronize()
    res = {}
    o = {k: v.cpu().numpy() for k, v in out.items()}
    res["knn_d2_exact"] = bool(np.array_equal(o["knn_d2"], g["knn_d2"]))
    res["knn_idx_eq"] = float((o["knn_idx"] == g["knn_idx"]).mean())
    res["mask_eq"] = bool(np.array_equal(o["mask"], g["mask"]))
    rows = g["rows"] if "rows" in g else slice(None)
    for k in ("mv_feature_agg", "feature_agg", "geo"):
        gk = {"mv_feature_agg": "multiview_feature_agg"}.get(k, k)
        res[k] = rel_err(o[k][rows], g[gk])
    for k in ("sigma", "weights", "rgb", "depth", "depth_uncertainty", "feat"):
        res[k] = rel_err(o[k], g[k])
    return res

if __name__ == "__main__":
    names = [n for n in CASES if CASES[n][0].N_importance == 0]
    precs = sys.argv[1:] or ["fp32"]
    for p in precs:
        for n in names:
            try:
                res = run(n, p)
                print(p, n, json.dumps({k: (v if isinstance(v, bool) else float(f"{v:.3g}")) for k, v in res.items()}))
            except Exception as e:
                print(p, n, "EXC", repr(e))
