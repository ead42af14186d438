"""Drop-in boundary: `nerf_loc_amd.conditional_nerf.ConditionalNeRF` vs the reference module (goldens from
tools/gen_golden.py `setup`): state_dict contract, per-frame setup (PyTorch), and — on the GPU — the end-to-end path."""
import json
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from nerf_loc_amd.synth import SceneConfig, add_setup_inputs, make_depth_fusion_weights, make_frame, make_weights
from tests.util import GOLDEN_DIR, load_golden, rel_err

CFG = SceneConfig("setup", R=24, S=16, W=32, V=3, H=32, Wimg=48, seed=21)


def _args(cfg):
    return NS(multires=10, multires_views=4, i_embed=0, backbone2d_fpn_dim=cfg.C, model_3d_hidden_dim=cfg.W,
              render=NS(N_samples=cfg.S, N_importance=cfg.N_importance, N_rand=1024, chunk=2048, lindisp=False, white_bkgd=False,
                        use_render_uncertainty=True, render_feature=True),
              use_scene_coord_memorization=False, matcher_hidden_dim=192, use_depth_supervision=False, matching=NS(fine_num_3d_keypoints=1024))


def _weights(cfg):
    w = dict(make_weights(cfg))
    w.update(make_depth_fusion_weights(cfg.seed))
    return {k: torch.from_numpy(v) for k, v in w.items()}


def test_state_dict_contract_equals_reference():
    from nerf_loc_amd.conditional_nerf import ConditionalNeRF
    want = json.load(open(os.path.join(GOLDEN_DIR, "state_dict_contract.json")))
    sd = ConditionalNeRF(_args(CFG)).state_dict()
    assert set(sd) == set(want), sorted(set(sd) ^ set(want))[:8]
    for k, shp in want.items():
        assert list(sd[k].shape) == shp, (k, tuple(sd[k].shape), shp)
    # strict load of a full reference-style checkpoint
    net = ConditionalNeRF(_args(CFG))
    net.load_state_dict(_weights(CFG), strict=True)


def test_depth_fusion_cnn_matches_reference_vis_featmaps():
    """The per-frame CNN (PyTorch) on the oracle's restatement of its hand-made input; on the GPU that input comes from
    nl_cross_view_features (tests/test_gpu_setup.py)."""
    from nerf_loc_amd.depth_fusion import DepthFusionNet
    from oracle import setup_oracle as sorc
    g = load_golden("setup")
    frame = add_setup_inputs(CFG, make_frame(CFG))
    net = DepthFusionNet().eval()
    pre = "multiview_aggregator.depth_fusion."
    net.load_state_dict({k[len(pre):]: v for k, v in _weights(CFG).items() if k.startswith(pre)}, strict=True)
    t = torch.from_numpy
    near, far = [float(x) for x in frame["depth_range"][0]]
    with torch.no_grad():
        out = net.encode(sorc.cnn_input(t(frame["topk_images"]), t(frame["topk_depths"]), t(frame["topk_Ks"]), t(frame["topk_poses"]), near, far))
    assert out.shape == g["vis_featmaps"].shape
    assert rel_err(out.numpy(), g["vis_featmaps"]) < 2e-5


def test_per_frame_setup_has_no_cpu_path():
    from nerf_loc_amd.frame_setup import backproject_support, cross_view_features
    frame = add_setup_inputs(CFG, make_frame(CFG))
    t = torch.from_numpy
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cross_view_features(t(frame["topk_images"]), t(frame["topk_depths"]), t(frame["topk_Ks"]), t(frame["topk_poses"]), 1.0, 5.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        backproject_support(t(frame["topk_images"]), t(frame["feat_fine_src"]), t(frame["topk_depths"]), t(frame["topk_Ks"]), t(frame["topk_poses"]), 4)


@pytest.mark.gpu
@pytest.mark.parametrize("case_name,precision,tol", [("setup", "fp32", 1e-4), ("setup", "bf16x3", 2e-4), ("setup_holes", "bf16x3", 2e-4)])
def test_dropin_end_to_end_matches_reference(case_name, precision, tol):
    """Same calls the reference's pose estimator makes: caches reset -> render_rays builds the frame (DepthFusionNet,
    back-projection, confidence) and renders; then descriptor queries.  Tolerance: setup (CNN on ROCm vs CPU) + renderer."""
    from nerf_loc_amd.conditional_nerf import ConditionalNeRF
    from tests.golden_cases import build_setup_case
    case = build_setup_case(case_name)
    CFG, frame, rays = case["cfg"], case["frame"], case["rays"]
    g = load_golden(case_name)
    dev = torch.device("cuda:0")
    net = ConditionalNeRF(_args(CFG), precision=precision).to(dev).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in case["weights"].items()}, strict=True)
    data = {k: torch.from_numpy(frame[k]).to(dev) for k in ("topk_images", "topk_depths", "topk_Ks", "topk_poses", "feat_fine_src", "feat_coarse_src",
                                                           "depth_range", "K", "pose")}
    data.update({"embedding_a": None, "H": frame["H"], "W": frame["W"], "stride_fine": 4, "stride_coarse": 8})
    rd = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in rays.items()}
    net.support_neural_points = None
    net.multiview_aggregator.vis_featmaps = None
    out = net.render_rays(data, rd)
    sp = net.support_neural_points["fine"]
    assert np.array_equal(sp["xyz"].cpu().numpy().shape, g["fine_xyz"].shape)
    assert rel_err(sp["xyz"].cpu().numpy(), g["fine_xyz"]) < 1e-5
    assert rel_err(sp["direction"].cpu().numpy(), g["fine_direction"]) < 1e-5
    assert rel_err(net.multiview_aggregator.vis_featmaps.cpu().numpy(), g["vis_featmaps"]) < 5e-5
    assert rel_err(sp["confidence"].cpu().numpy(), g["fine_confidence"]) < tol
    assert rel_err(net.support_neural_points["coarse"]["keypoint_score"].cpu().numpy(), g["coarse_keypoint_score"]) < 1e-5
    for k in ("rgb", "depth", "weights", "depth_uncertainty", "feat"):
        assert rel_err(out[k].cpu().numpy(), g[k]) < tol, (k, rel_err(out[k].cpu().numpy(), g[k]))
    assert np.array_equal(out["mask"].cpu().numpy(), g["mask"])
    pts = torch.from_numpy(g["query_pts"]).to(dev)
    desc_f, _, _ = net.query_fine(data, pts)
    desc_c, _, _ = net.query_coarse(data, pts)
    assert rel_err(desc_f.detach().cpu().numpy(), g["desc_fine"]) < tol
    assert rel_err(desc_c.detach().cpu().numpy(), g["desc_coarse"]) < tol
    # render_image (model.py:602-639): all pixels in one library call; a pixel's result does not depend on the batch it is in
    img = net.render_image(data)
    assert img["rgb"].shape == (CFG.H, CFG.Wimg, 3) and img["weights"].shape == (CFG.H, CFG.Wimg, CFG.S)
    assert torch.isfinite(img["rgb"]).all()
    pts2d = torch.tensor([[0., 0.], [5., 3.], [CFG.Wimg - 1., CFG.H - 1.], [17., 20.], [40., 9.]], device=dev)
    sub = net.points_2d_to_rays(pts2d, CFG.H, CFG.Wimg, data["K"], data["pose"])
    sub["depth_range"] = data["depth_range"][0]
    part = net.render_rays(data, sub)
    for k in ("rgb", "depth", "weights", "feat"):
        want = img[k][pts2d[:, 1].long(), pts2d[:, 0].long()]
        assert torch.equal(part[k].view(want.shape), want), k
    # the caller's per-frame cache reset is honoured (nerf_pose_estimator.py:289-290)
    net.support_neural_points = None
    net.multiview_aggregator.vis_featmaps = None
    out2 = net.render_rays(data, rd)
    # (MIOpen may pick another conv algorithm for the per-frame CNN on a later call: equal up to fp32 noise, not bitwise)
    assert rel_err(out2["rgb"].cpu().numpy(), out["rgb"].cpu().numpy()) < 1e-5


@pytest.mark.gpu
def test_dropin_training_paths_raise_clearly():
    from nerf_loc_amd.conditional_nerf import ConditionalNeRF
    net = ConditionalNeRF(_args(CFG)).to("cuda:0")
    with pytest.raises(NotImplementedError):
        net.compute_render_loss({})
    net.train()
    with pytest.raises(NotImplementedError):
        net.render_rays({}, {})


@pytest.mark.gpu
def test_dropin_without_feature_rendering():
    """render.render_feature=False (model.py:84-89, 594-598): no feat_mlp in the state_dict, no 'feat' in the outputs, everything else
    unchanged."""
    from nerf_loc_amd.conditional_nerf import ConditionalNeRF
    from tests.golden_cases import build_setup_case
    case = build_setup_case("setup")
    cfg, frame, rays = case["cfg"], case["frame"], case["rays"]
    dev = torch.device("cuda:0")
    data = {k: torch.from_numpy(frame[k]).to(dev) for k in ("topk_images", "topk_depths", "topk_Ks", "topk_poses", "feat_fine_src", "feat_coarse_src",
                                                           "depth_range", "K", "pose")}
    data.update({"embedding_a": None, "H": frame["H"], "W": frame["W"], "stride_fine": 4, "stride_coarse": 8})
    rd = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in rays.items()}
    outs = []
    for render_feature in (True, False):
        args = _args(cfg)
        args.render.render_feature = render_feature
        net = ConditionalNeRF(args, precision="bf16x3").to(dev).eval()
        w = {k: torch.from_numpy(v) for k, v in case["weights"].items() if render_feature or not k.startswith("feat_mlp.")}
        net.load_state_dict(w, strict=True)
        outs.append(net.render_rays(data, rd))
    assert "feat" in outs[0] and "feat" not in outs[1]
    assert torch.equal(outs[0]["mask"], outs[1]["mask"])
    for k in ("rgb", "depth", "weights", "depth_uncertainty"):   # two module instances: MIOpen may pick another algorithm for the per-frame CNN
        assert rel_err(outs[1][k].cpu().numpy(), outs[0][k].cpu().numpy()) < 1e-5, k
