"""Debug (round 5): which K slab / layer / cross term of the fused neural-point kernel's MX-FP6 arithmetic is off — f16mx against fp32 on the w256s128 golden scene with
the wide layers' weights masked to one 64-column slab, rounded to fp16 (second cross term = 0), one layer at a time.  Found the result / scale-operand register overlap
of the fp6 packing builtins.  python tools/mx6_debug.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from nerf_loc_amd.renderer import HipRenderer
from tests.golden_cases import build_case
from tests.util import rel_err

def renderer(case, prec, weights):
    cfg, fr = case["cfg"], case["frame"]
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, prec)
    r.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
    r.set_frame(fr["topk_images"], fr["feat_fine_src"], fr["vis_featmaps"], fr["topk_Ks"], fr["topk_poses"], cfg.near, cfg.far, fr["support_fine"])
    return r
def z_of(cfg, R):
    lin = torch.linspace(0, 1, cfg.S)
    return (torch.tensor(cfg.near) * (1 - lin) + torch.tensor(cfg.far) * lin).expand(R, cfg.S).contiguous()
case = build_case("w256s128"); cfg = case["cfg"]
base = {k: v.copy() for k, v in case["weights"].items()}
names = [k for k in base if k.startswith("base_mlp") ]
print(names)
def run(weights, tag):
    outs = {}
    for prec in ("fp32", "f16mx"):
        r = renderer(case, prec, weights)
        out = r.render_rays(case["rays"]["rays_o"], case["rays"]["rays_d"], case["frame"]["pose"][:3, 3], z_vals=z_of(cfg, cfg.R), white_bkgd=cfg.white_bkgd, intermediates=True)
        outs[prec] = out["feature_agg"].cpu().numpy()
    print(f"{tag:50s} feature_agg f16mx vs fp32: {rel_err(outs['f16mx'], outs['fp32']):.2e}", flush=True)
f16 = lambda a: a.astype(np.float16).astype(np.float32)
wide = ["base_mlp.2.weight", "base_mlp.4.weight", "base_mlp_attn.w_ks.weight", "base_mlp_attn.w_vs.weight"]
run(base, "as is")
for lay in wide:
    w = dict(base)
    for k in wide:
        if k != lay:
            m = base[k].copy(); m[:, 128:192] = 0; w[k] = m
    run(w, f"slab 2 zeroed everywhere but in {lay}")
w = dict(base)
for k in wide:
    m = base[k].copy(); m[:, 128:192] = 0; w[k] = m
run(w, "slab 2 zeroed in all four")
