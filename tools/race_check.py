"""Determinism check: the same batch rendered repeatedly must give bit-identical outputs.  python tools/race_check.py [case] [precision] [reps]
case: a golden case name (w256s128, c1, ...) or a BASELINE config (c2, c3)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loc_amd.renderer import HipRenderer
case = sys.argv[1] if len(sys.argv) > 1 else "w256s128"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
want_feat = os.environ.get("RC_NOFEAT") is None
inter_on = os.environ.get("RC_NOINTER") is None
if case in ("c2", "c3", "c4"):
    from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights
    cfg = CONFIGS[case]; frame, weights = make_frame(cfg), make_weights(cfg); rays = make_rays(cfg, frame)
else:
    from tests.golden_cases import build_case
    c = build_case(case); cfg, frame, rays, weights = c["cfg"], c["frame"], c["rays"], c["weights"]
r = HipRenderer(cfg.W, cfg.C, cfg.S_total, prec)
r.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
KEYS = ("rgb", "depth", "weights", "feat", "sigma", "feature_agg", "geo")
inter = inter_on and cfg.R * cfg.S_total <= 1 << 18
def run():
    o = r.render_rays(rays["rays_o"], rays["rays_d"], frame["pose"][:3, 3], intermediates=inter, want_feat=want_feat)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in o.items() if k in KEYS}
a = run()
nbad = 0
for it in range(reps):
    b = run()
    diff = {k: (float((a[k].float() - b[k].float()).abs().max()), int((a[k] != b[k]).sum())) for k in a if not torch.equal(a[k], b[k])}
    if diff:
        nbad += 1
        print("rep", it, diff)
        if "rgb" in diff:
            print("   bad rays", (a["rgb"] != b["rgb"]).any(1).nonzero().flatten()[:16].tolist())
print(f"{case} {prec}: {nbad} of {reps} repetitions differ from the first")
