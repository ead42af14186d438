#!/usr/bin/env python3
"""Time the per-frame setup (DepthFusionNet, back-projection, confidence pass, KNN grid, frame tables) and one
full `render_image` through the drop-in module at BASELINE config-2 sizes.  Setup = HIP library (back-projection, cross-view
consistency features, confidence aggregate, KNN grid, frame tables) + the per-frame CNN on PyTorch-ROCm."""
import os, sys, time
from types import SimpleNamespace as NS
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights, make_depth_fusion_weights, add_setup_inputs
from nerf_loc_amd.conditional_nerf import ConditionalNeRF

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
args = NS(multires=10, multires_views=4, i_embed=0, backbone2d_fpn_dim=cfg.C, model_3d_hidden_dim=cfg.W,
          render=NS(N_samples=cfg.S, N_importance=cfg.N_importance, N_rand=1024, chunk=4096, lindisp=False, white_bkgd=False,
                    use_render_uncertainty=True, render_feature=True),
          use_scene_coord_memorization=False, matcher_hidden_dim=192, use_depth_supervision=False, matching=NS(fine_num_3d_keypoints=1024))
frame = add_setup_inputs(cfg, make_frame(cfg))
rays = make_rays(cfg, frame)
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)   # inference path (under enable_grad the descriptor queries take the gradient path)
net = ConditionalNeRF(args, precision="bf16x3").to(dev).eval()
w = dict(make_weights(cfg)); w.update(make_depth_fusion_weights(cfg.seed))
net.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
data = {k: torch.from_numpy(frame[k]).to(dev) for k in ("topk_images", "topk_depths", "topk_Ks", "topk_poses", "feat_fine_src", "feat_coarse_src", "depth_range", "K", "pose")}
data.update({"embedding_a": None, "H": frame["H"], "W": frame["W"], "stride_fine": 4, "stride_coarse": 8})
rd = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in rays.items()}

def frame_setup_and_first_batch():
    net.support_neural_points = None
    net.multiview_aggregator.vis_featmaps = None
    return net.render_rays(data, rd)

frame_setup_and_first_batch(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); frame_setup_and_first_batch(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t_first = min(ts)
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); net.render_rays(data, rd); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t_batch = min(ts)
if len(sys.argv) > 2 and sys.argv[2] == 'setuponly':
    print(f'{cfg.name}: per-frame setup {1e3*(t_first-t_batch):.1f} ms'); sys.exit(0)
pts = net.support_neural_points["fine"]["xyz"][:1024].contiguous()
def tq(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"{cfg.name}: descriptor queries of 1024 points: query_fine {tq(lambda: net.query_fine(data, pts)):.2f} ms, query_coarse {tq(lambda: net.query_coarse(data, pts)):.2f} ms")
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); img = net.render_image(data); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t_img = min(ts)
print(f"{cfg.name}: setup + first {cfg.R}-ray batch {t_first*1e3:.1f} ms; warm batch {t_batch*1e3:.1f} ms -> per-frame setup {1e3*(t_first-t_batch):.1f} ms; "
      f"render_image {cfg.H}x{cfg.Wimg} = {cfg.H*cfg.Wimg} rays in {t_img*1e3:.1f} ms ({cfg.H*cfg.Wimg/t_img/1e3:.0f} k rays/s through the module)")
