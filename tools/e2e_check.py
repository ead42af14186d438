import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from nerf_loc_amd.conditional_nerf import ConditionalNeRF
from tests.golden_cases import build_setup_case
from tests.test_dropin_module import _args
from tests.util import load_golden, rel_err
dev = torch.device("cuda:0")
for flags in ("default", "fp32conv"):
    if flags == "fp32conv":
        torch.backends.cudnn.allow_tf32 = False
        torch.backends.cuda.matmul.allow_tf32 = False
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False
    for case_name in ("setup", "setup_holes"):
        case = build_setup_case(case_name); CFG, frame, rays = case["cfg"], case["frame"], case["rays"]; g = load_golden(case_name)
        for precision in ("fp32", "bf16x3", "f16mx"):
            net = ConditionalNeRF(_args(CFG), precision=precision).to(dev).eval()
            net.load_state_dict({k: torch.from_numpy(v) for k, v in case["weights"].items()}, strict=True)
            data = {k: torch.from_numpy(frame[k]).to(dev) for k in ("topk_images", "topk_depths", "topk_Ks", "topk_poses", "feat_fine_src", "feat_coarse_src", "depth_range", "K", "pose")}
            data.update({"embedding_a": None, "H": frame["H"], "W": frame["W"], "stride_fine": 4, "stride_coarse": 8})
            rd = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in rays.items()}
            net.support_neural_points = None; net.multiview_aggregator.vis_featmaps = None
            out = net.render_rays(data, rd)
            err_cnn = rel_err(net.multiview_aggregator.vis_featmaps.cpu().numpy(), g["vis_featmaps"])
            conf = rel_err(net.support_neural_points["fine"]["confidence"].cpu().numpy(), g["fine_confidence"])
            err_b = {k: f"{rel_err(out[k].cpu().numpy(), g[k]):.1e}" for k in ("rgb", "depth", "weights", "depth_uncertainty", "feat")}
            pts = torch.from_numpy(g["query_pts"]).to(dev)
            with torch.no_grad():
                df, _, _ = net.query_fine(data, pts); dc, _, _ = net.query_coarse(data, pts)
            print(flags, case_name, precision, f"cnn {err_cnn:.1e} conf {conf:.1e}", err_b, f"desc_f {rel_err(df.cpu().numpy(), g['desc_fine']):.1e} desc_c {rel_err(dc.cpu().numpy(), g['desc_coarse']):.1e}", flush=True)
