"""Randomised module-level cross-check (GPU): the drop-in ConditionalNeRF in train() mode on random small scenes (views, feature width, hidden width, samples) —
compute_render_loss and the train-mode descriptor queries with the library's training nodes (hip_training = True, the default) against the all-eager fp32 graph of
the same module (hip_training = False): losses, descriptors and every parameter's / map's gradient.  python tools/module_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS
from nerf_loc_amd.conditional_nerf import ConditionalNeRF
from nerf_loc_amd.synth import SceneConfig, make_frame, make_rays, make_weights
from tests.golden_cases import add_setup_inputs, make_depth_fusion_weights


def _args(cfg):
    return NS(multires=10, multires_views=4, i_embed=0, backbone2d_fpn_dim=cfg.C, model_3d_hidden_dim=cfg.W,
              render=NS(N_samples=cfg.S, N_importance=cfg.N_importance, N_rand=1024, chunk=2048, lindisp=False, white_bkgd=False,
                        use_render_uncertainty=True, render_feature=True),
              use_scene_coord_memorization=False, matcher_hidden_dim=192, use_depth_supervision=False, matching=NS(fine_num_3d_keypoints=1024))


def run(ncases=6, seed0=0, precision="f16mx"):
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    worst_all = 0.0
    for case in range(ncases):
        rng = np.random.default_rng(9000 + 100 * seed0 + case)
        W = int(rng.choice([32, 64, 128])); S = int(8 * rng.integers(2, 9)); V = int(rng.integers(2, 7)); C = int(rng.choice([32, 64, 100, 192]))
        H, Wimg = int(16 * rng.integers(2, 5)), int(16 * rng.integers(2, 6)); R = int(rng.integers(8, 40))
        cfg = SceneConfig(f"mfuzz{case}", R=R, S=S, W=W, V=V, H=H, Wimg=Wimg, C=C, seed=9100 + 100 * seed0 + case)
        frame = add_setup_inputs(cfg, make_frame(cfg))
        weights = dict(make_weights(cfg)); weights.update(make_depth_fusion_weights(cfg.seed))
        rays = make_rays(cfg, frame)
        base = frame["support_fine"]["xyz"][::5][:32]
        pts = t((base + 0.004 * rng.standard_normal((len(base), 3))).astype(np.float32))
        tc, tf = t(rng.standard_normal((len(base), 192)).astype(np.float32)), t(rng.standard_normal((len(base), 192)).astype(np.float32))
        res = {}
        for hip in (False, True):
            data = {k: t(frame[k]) for k in ("topk_images", "topk_depths", "topk_Ks", "topk_poses", "feat_fine_src", "feat_coarse_src", "depth_range", "K", "pose")}
            data["feat_fine_src"] = data["feat_fine_src"].clone().requires_grad_(True)
            data["feat_coarse_src"] = data["feat_coarse_src"].clone().requires_grad_(True)
            trng = np.random.default_rng(cfg.seed + 2000)
            img = trng.random((3, cfg.H, cfg.Wimg)).astype(np.float32)
            pyr = trng.standard_normal((1, cfg.C, cfg.H // 2, cfg.Wimg // 2)).astype(np.float32)
            data.update({"embedding_a": None, "H": frame["H"], "W": frame["W"], "stride_fine": 4, "stride_coarse": 8,
                         "sample_coords": t(rays["pixel_coordinates"]), "img": t(img), "feat_pyramid": {"layer1": t(pyr)}})
            torch.manual_seed(1234 + case)   # sample_rays draws the pixels
            net = ConditionalNeRF(_args(cfg), precision=precision).to(dev).train()
            net.hip_training = hip
            net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()}, strict=True)
            desc_c, _, _ = net.query_coarse(data, pts)
            desc_f, _, _ = net.query_fine(data, pts)
            loss = (desc_c * tc).sum() / len(pts) + (desc_f * tf).sum() / len(pts)
            lr = net.compute_render_loss(data)[0]
            loss = loss + lr
            loss.backward()
            g = {k: v.grad.detach().clone() for k, v in net.named_parameters() if v.grad is not None}
            g["feat_fine_src"], g["feat_coarse_src"] = data["feat_fine_src"].grad.clone(), data["feat_coarse_src"].grad.clone()
            res[hip] = (float(loss.detach()), desc_c.detach(), desc_f.detach(), g, float(lr.detach()))
            if hip:   # eval mode: the inference entry points (no autograd; stage kernels without the training nodes) on the same points
                net.eval()
                with torch.no_grad():
                    res["eval"] = (net.query_coarse(data, pts)[0].detach(), net.query_fine(data, pts)[0].detach())
                    img_out = net.render_image(data)   # the whole image through the fused inference path (one library call)
                    assert all(torch.isfinite(v.float()).all() for v in img_out.values()), "render_image"
                    res["image"] = img_out
        (l0, c0, f0, g0, r0), (l1, c1, f1, g1, r1) = res[False], res[True]
        gmax = max(float(v.abs().max()) for v in g0.values())
        worst = ("", 0.0)
        for k, b in g0.items():
            a = g1.get(k)
            if a is None:
                assert float(b.abs().max()) <= 1e-5 * gmax, (case, k, "missing")
                continue
            l2 = float((a - b).norm() / max(float(b.norm()), 1e-5 * gmax))
            if l2 > worst[1]: worst = (k, l2)
        dc = float((c1 - c0).abs().max() / c0.abs().max()); df = float((f1 - f0).abs().max() / f0.abs().max())
        ec, ef = res["eval"]
        dce = float((ec - c0).abs().max() / c0.abs().max()); dfe = float((ef - f0).abs().max() / f0.abs().max())
        assert dce < 2e-3 and dfe < 2e-3, ("eval-mode descriptors", dce, dfe)
        print(f"case {case}: W={W} S={S} V={V} C={C} {H}x{Wimg} R={R}: loss {l0:.6f} vs {l1:.6f} (render {r0:.6f} vs {r1:.6f}), desc {dc:.1e} / {df:.1e}, worst gradient L2-rel {worst[1]:.2e} ({worst[0]})", flush=True)
        assert abs(l1 - l0) < 2e-3 * abs(l0) and dc < 2e-3 and df < 2e-3 and worst[1] < 5e-2, "MISMATCH"
        worst_all = max(worst_all, worst[1])
    return worst_all


if __name__ == "__main__":
    print("worst", run(int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 0, sys.argv[3] if len(sys.argv) > 3 else "bf16x3"))
