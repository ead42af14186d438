"""Randomised cross-check of the gradients (GPU): small random scenes (hidden width 32 ... 256 in steps of 32, up to 256 samples, 1 ... 16 views, feature widths
8 ... 192 — odd ones on the frozen path —, support sets also smaller than K, 1 ... 13 rays, white background); frozen weights (PoseOptimizer) or training; the
whole path as one node (keep / kept pair, or the chunking pair over a small workspace) or one node per stage.  Every gradient — rays, pose, 84 parameter
tensors, both maps, support features — of the library in the parity mode AND in the fp32 mode against autograd of the eager graph in fp64, with the same graph
in fp32 as the yardstick for what this scene's conditioning allows.  It found what no fixed case had: every fixed gradient case used the reference's
192-channel feature maps, and for C <= 123 the transposed out_fc.0 product runs on the streaming kernel whose weight stream was never packed (all gradients
through the statistics rows vanished in the non-fp32 modes).  python tools/grad_fuzz.py [cases] [seed]; tests/test_backward_kernels.py runs a few cases."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loc_amd import diff_render as dr
from nerf_loc_amd.renderer import HipRenderer
from nerf_loc_amd.synth import SceneConfig, make_frame, make_rays, make_weights

def run(ncases=20, seed0=0, verbose=True):
    """-> worst L2-relative gradient error over the cases (asserts on a mismatch)"""
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    worst_all = 0.0
    only = [int(x) for x in os.environ["ONLY"].split(",")] if os.environ.get("ONLY") else None   # re-run single cases of a sweep
    for case in range(ncases):
        if only is not None and case not in only:
            continue
        rng = np.random.default_rng(seed0 * 1000 + case)
        W = int(rng.choice([32, 64, 96, 128, 160, 192, 224, 256])); S = int(8 * rng.integers(2, 13)) if rng.random() < 0.7 else int(8 * rng.integers(13, 33)); V = int(rng.integers(1, 17)); C = int(rng.choice([8, 32, 60, 64, 100, 124, 128, 192]))
        # the path under test: frozen weights (PoseOptimizer) or training; the whole path as one node (keep / kept pair, or the chunking pair over a small
        # workspace) or one node per stage.  Frozen weights also take feature widths that are not multiples of 4
        train = bool(rng.random() < 0.6); variant = str(rng.choice(["keep", "chunk", "stages"]))
        if not train and rng.random() < 0.4: C = int(rng.choice([7, 31, 61, 101]))
        white = bool(rng.random() < 0.3)
        if os.environ.get("FORCE_VARIANT"):   # "train|frozen,keep|chunk|stages"
            tv, variant = os.environ["FORCE_VARIANT"].split(","); train = tv == "train"
        H, Wimg = int(8 * rng.integers(3, 9)), int(8 * rng.integers(3, 12)); R = int(rng.integers(1, 14))
        if os.environ.get("FORCE"):   # "W,S,V,C,H,Wimg,R"
            W, S, V, C, H, Wimg, R = [int(x) for x in os.environ["FORCE"].split(",")]
        cfg = SceneConfig(f"fuzz{case}", R=max(R, 2), S=S, W=W, V=V, H=H, Wimg=Wimg, C=C, seed=5000 + seed0 * 1000 + case)
        print(f"case {case} config: W={W} S={S} V={V} C={C} {H}x{Wimg} R={R} {'train' if train else 'frozen'} {variant}{' white' if white else ''}", flush=True)
        frame = make_frame(cfg); rays = make_rays(cfg, frame); weights = make_weights(cfg)
        if rng.random() < 0.25:   # fewer support points than K
            m = int(rng.integers(1, 8)); frame["support_fine"] = {k: np.ascontiguousarray(v[:m]) for k, v in frame["support_fine"].items()}
        M = frame["support_fine"]["xyz"].shape[0]
        o0, d0 = t(rays["rays_o"][:R]), t(rays["rays_d"][:R])
        lin = torch.linspace(0, 1, cfg.S_total, device=dev)
        z = (cfg.near * (1 - lin) + cfg.far * lin).expand(R, cfg.S_total).contiguous()
        g = torch.Generator().manual_seed(case)
        cot = {k: torch.randn(*shp, generator=g).to(dev) for k, shp in (("rgb", (R, 3)), ("depth", (R,)), ("depth_uncertainty", (R,)), ("feat", (R, cfg.C)), ("weights", (R, cfg.S_total)))}
        res = {}
        for prec in ("eager64", "eager", "fp32", "bf16x3"):
            eager = prec.startswith("eager")
            dt = torch.float64 if prec == "eager64" else torch.float32
            t = (lambda a, _dt=dt: (lambda x: x.to(_dt) if x.is_floating_point() else x)(torch.from_numpy(np.ascontiguousarray(a)).to(dev)))
            r = HipRenderer(cfg.W, cfg.C, cfg.S_total, "fp32" if eager else prec)
            r.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
            r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
            p = {k: t(v).requires_grad_(train) for k, v in weights.items()}
            fr = {k: t(frame[k]) for k in ("topk_Ks", "topk_poses", "topk_images", "feat_fine_src", "vis_featmaps")}
            sp = {k: t(v) for k, v in frame["support_fine"].items()}
            if train: fr["feat_fine_src"].requires_grad_(True); fr["vis_featmaps"].requires_grad_(True); sp["feature"].requires_grad_(True)
            fr.update({"near": float(cfg.near), "far": float(cfg.far), "support": sp})
            o, d, pose = o0.to(dt).clone().requires_grad_(True), d0.to(dt).clone().requires_grad_(True), t(frame["pose"]).clone().requires_grad_(True)
            use_beta = train and variant != "chunk"
            keep_bytes, orig_bw = dr.KEEP_BYTES, r.render_rays_backward
            if variant == "chunk" and not eager:
                dr.KEEP_BYTES = 0
                wr = int(rng.integers(1, max(2, R)))
                r.render_rays_backward = lambda *a, _o=orig_bw, _w=wr, **k: _o(*a, workspace_rays=_w, **k)
            try:
                out = dr.render_rays_diff(p, fr, o, d, z.to(dt), pose, lambda q: r.knn(q.float(), 8)[1], frozen_renderer=r if (not eager and not train) else None,
                                          train_renderer=r if (not eager and train) else None, whole_path=variant != "stages", beta=use_beta, white_bkgd=white)
                loss = sum((out[k] * cot[k].to(dt)).sum() for k in cot) + ((out["beta"] * cot["depth"].to(dt)).sum() if use_beta else 0.0)
                leaves = {"rays_o": o, "rays_d": d, "pose": pose}
                if train:
                    leaves.update({"feat_fine_src": fr["feat_fine_src"], "vis_featmaps": fr["vis_featmaps"], "support.feature": sp["feature"]})
                    leaves.update({n: p[n] for n in dr.RENDER_PARAMS})
                gs = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
            finally:
                dr.KEEP_BYTES = keep_bytes
            res[prec] = ({k: v.detach() for k, v in out.items()}, dict(zip(leaves.keys(), gs)))
            del r
        # reference: autograd of the eager graph in fp64; yardstick: the same graph in fp32.  The function is only piecewise smooth (LeakyReLU, MaxPool ties, the
        # visibility formula's |.| and clamps, in-image thresholds): a forward value that differs in the last bits puts a borderline sample on the other branch, and
        # the few tensors fed by that branch move by percents (the LayerNorm tables in front of a MaxPool; a scalar bias that is a sum of cancelling terms) — for
        # fp32 autograd as for the library.  So: every tensor within 5e-2 in the L2 norm or within 5x of fp32 autograd, EXCEPT at most four tensors per mode
        # (none above 0.5), and the median over all tensors below 2e-3.  (The bug this tool found put 1.0 on thirty tensors.)
        (o64, g64), (o32e, g32e) = res["eager64"], res["eager"]
        gmax = max(float(v.abs().max()) for v in g64.values() if v is not None)
        def l2err(ga):
            out_ = {}
            for k, b in g64.items():
                a = ga.get(k)
                if b is None or a is None:
                    assert (a is None or float(a.abs().max()) <= 1e-5 * gmax) and (b is None or float(b.abs().max()) <= 1e-5 * gmax), (case, k)
                    continue
                assert torch.isfinite(a).all(), (case, k)
                out_[k] = float((a.double() - b).norm() / max(float(b.norm()), 1e-5 * gmax))
            return out_
        yard = l2err(g32e)
        worst = ("", 0.0)
        errs = {}
        for mode in ("fp32", "bf16x3"):
            em = l2err(res[mode][1])
            big = {k: e for k, e in em.items() if e >= max(5e-2, 5 * yard.get(k, 0.0))}
            if len(em) <= 5:   # frozen weights: rays and pose only
                assert not big, (case, mode, big)
            else:
                med, ymed = float(np.median(list(em.values()))), float(np.median([yard[k] for k in em if k in yard] or [0.0]))
                if os.environ.get("VERBOSE"): print(f"    {mode}: median {med:.2e}, fp32-autograd yardstick median {ymed:.2e}; worst:", sorted(((round(v, 4), k) for k, v in em.items()), reverse=True)[:8], flush=True)
                # the median bar follows the scene's conditioning like the per-tensor bar does: long rays (S > 128) at W = 256 put fp32 autograd itself at 2-5e-3
                assert len(big) <= 4 and all(e < 0.5 for e in big.values()) and med < max(2e-3, 3 * ymed), (case, mode, big, med, ymed)
            for k, e in em.items():
                errs[f"{mode}:{k}"] = e
                if e > worst[1] and k not in big: worst = (f"{mode}:{k}", e)
        o32, o16 = res["eager64"][0], res["bf16x3"][0]
        if os.environ.get("VERBOSE"): print("   ", sorted(((round(v, 4), k) for k, v in errs.items()), reverse=True)[:14])
        fwd = max(float((o16[k].double() - o32[k].double()).abs().max() / max(float(o32[k].double().abs().max()), 1e-6)) for k in cot)
        worst_all = max(worst_all, worst[1])
        print(f"case {case}: W={W} S={S} V={V} C={C} {H}x{Wimg} R={R} M={M} {'train' if train else 'frozen'} {variant}: forward {fwd:.1e}, worst gradient L2-rel {worst[1]:.2e} ({worst[0]})", flush=True)
        # (the forward figure is reported only: tools/forward_fuzz.py holds the forward to 1e-4, border-line samples excluded; the gradient criteria are above)
    if verbose: print("all cases passed; worst gradient L2-rel", worst_all)
    return worst_all


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
