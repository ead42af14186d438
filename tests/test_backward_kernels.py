"""HIP backward kernels, first slice: nl_composite_backward and nl_knn_backward against PyTorch autograd of the same
expressions (fp32 and, as the yardstick for rounding, fp64).  End to end they run inside the gradient path: the pose / training-step
gradient tests of tests/test_diff_render.py compare that path with the reference's own autograd goldens."""
import os
import numpy as np
import pytest
import torch

from nerf_loc_amd import diff_render as dr
from tests.util import rel_err


def _composite_inputs(R, S, C, seed, dev, dt=torch.float32):
    g = torch.Generator().manual_seed(seed)
    z = torch.sort(0.3 + 4.7 * torch.rand(R, S, generator=g), -1)[0]
    sigma = torch.nn.functional.softplus(2 * torch.randn(R, S, generator=g))      # thin to opaque, some rays saturate (T underflows behind them)
    sigma[: R // 4] *= 40.0
    rgb_s, ft = torch.rand(R, S, 3, generator=g), torch.randn(R, S, C, generator=g)
    cot = [torch.randn(R, 3, generator=g), torch.randn(R, generator=g), torch.randn(R, generator=g), torch.randn(R, C, generator=g), torch.randn(R, S, generator=g)]
    mv = lambda t: t.to(dev).to(dt)
    return mv(z), mv(sigma), mv(rgb_s), mv(ft), [mv(c) for c in cot]


def test_composite_function_falls_back_to_autograd_on_the_cpu():
    """Without a GPU the gradient path uses plain autograd of the same expression (the HIP kernel is never emulated on the CPU)."""
    z, sigma, rgb_s, ft, cot = _composite_inputs(8, 16, 5, 0, "cpu")
    assert not dr._hip_ok(sigma, rgb_s, ft, z)
    outs = dr.composite_eager(sigma.requires_grad_(True), rgb_s, ft, z, True)
    torch.autograd.grad(sum((o * c).sum() for o, c in zip(outs, cot)), sigma)


@pytest.mark.gpu
@pytest.mark.parametrize("S,C,white", [(32, 192, False), (64, 192, True), (128, 192, False), (192, 192, True), (256, 7, False), (40, 0, False)])
def test_composite_backward_matches_autograd(S, C, white):
    dev = torch.device("cuda:0")
    R = 37
    z, sigma, rgb_s, ft, cot = _composite_inputs(R, S, max(C, 1), S, dev)
    if C == 0:
        ft = ft[..., :0].contiguous()
        cot[3] = cot[3][..., :0].contiguous()

    def run(fn, dt):
        ins = [t.detach().to(dt).requires_grad_(True) for t in (sigma, rgb_s, ft)]
        outs = fn(ins[0], ins[1], ins[2], z.to(dt), white)
        loss = sum((o * c.to(dt)).sum() for o, c in zip(outs, cot))
        return [o.detach() for o in outs], torch.autograd.grad(loss, ins)
    o_hip, g_hip = run(dr.CompositeFn.apply, torch.float32)
    o_ref, g_ref = run(dr.composite_eager, torch.float32)
    _, g_64 = run(dr.composite_eager, torch.float64)
    for a, b in zip(o_hip, o_ref):
        assert torch.equal(a, b)          # the forward IS the eager expression
    for name, a, b, c in zip(("sigma", "rgb_s", "ft"), g_hip, g_ref, g_64):
        if a.numel() == 0:
            continue
        e_hip, e_ref = rel_err(a.cpu().numpy(), c.cpu().numpy()), rel_err(b.cpu().numpy(), c.cpu().numpy())
        assert e_hip < max(3 * e_ref, 2e-6), (name, S, e_hip, e_ref)      # as close to fp64 as fp32 autograd is
        assert torch.isfinite(a).all()
    # single incoming gradients (the others None), as a loss on one output produces them
    ins = [t.detach().requires_grad_(True) for t in (sigma, rgb_s, ft)]
    outs = dr.CompositeFn.apply(ins[0], ins[1], ins[2], z, white)
    g1 = torch.autograd.grad((outs[1] * cot[1]).sum(), ins[0])[0]
    ins2 = [t.detach().requires_grad_(True) for t in (sigma, rgb_s, ft)]
    g2 = torch.autograd.grad((dr.composite_eager(ins2[0], ins2[1], ins2[2], z, white)[1] * cot[1]).sum(), ins2[0])[0]
    assert rel_err(g1.cpu().numpy(), g2.cpu().numpy()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,M", [(1000, 8, 500), (257, 1, 64), (64, 8, 5)])
def test_knn_backward_matches_autograd(N, K, M):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(N + K)
    xyz, sp = torch.randn(N, 3, generator=g).to(dev), torch.randn(M, 3, generator=g).to(dev)
    idx = torch.randint(0, M, (N, K), generator=g).to(dev)
    cot = torch.randn(N, K, generator=g).to(dev)
    if M < K:
        cot[:, M:] = 0      # padded slots (knn_utils.py:48-53): the kernel skips them, the reference's distances there are constants
        idx[:, M:] = 0
    a, b = xyz.clone().requires_grad_(True), sp.clone().requires_grad_(True)
    ga, gb = torch.autograd.grad((dr.KnnDist2Fn.apply(a, b, idx) * cot).sum(), [a, b])
    a2, b2 = xyz.clone().requires_grad_(True), sp.clone().requires_grad_(True)
    off = a2[:, None, :] - b2[idx]
    ra, rb = torch.autograd.grad(((off * off).sum(-1) * cot).sum(), [a2, b2])
    assert rel_err(ga.cpu().numpy(), ra.cpu().numpy()) < 1e-6
    assert rel_err(gb.cpu().numpy(), rb.cpu().numpy()) < 1e-5      # atomics: another summation order
    # the query-only case (frozen support table: PoseOptimizer) allocates no support-side gradient
    a3 = xyz.clone().requires_grad_(True)
    g3 = torch.autograd.grad((dr.KnnDist2Fn.apply(a3, sp, idx) * cot).sum(), a3)[0]
    assert torch.equal(g3, ga)


@pytest.mark.gpu
@pytest.mark.parametrize("case,precision,tol,ntrunc", [("tiny_full", "fp32", 2e-4, None), ("c1", "fp32", 2e-4, None), ("w256s128", "fp32", 2e-4, None),
                                                       ("w256s128", "bf16x3", 2e-4, None), ("fewpts", "fp32", 2e-4, None),
                                                       # ragged ends of the two fused kernels of the frozen-weight path (point_fused2 KEEP instance, point_bwd chain):
                                                       # 1003 samples = 62 full 16-sample tiles + 11 samples, the last wave's fourth sample missing
                                                       ("w256s128", "bf16x3", 2e-4, 1003), ("w128s64", "bf16x3", 2e-4, 333)])
def test_point_branch_backward_matches_autograd(case, precision, tol, ntrunc):
    """nl_point_mlp_backward (frozen weights) against autograd of the eager restatement of the branch (diff_render._point_branch, itself
    checked against the reference's autograd): gradients w.r.t. the sample positions, the viewing directions and the query features."""
    from nerf_loc_amd.renderer import HipRenderer
    from tests.golden_cases import build_case
    c = build_case(case)
    cfg, frame, rays = c["cfg"], c["frame"], c["rays"]
    dev = torch.device("cuda:0")
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, precision)
    r.load_weights({k: torch.from_numpy(v) for k, v in c["weights"].items()})
    r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
    # (fewpts: fewer support points than K — zero rows at distance 0 in the padded neighbour slots, knn_utils.py:211-220.  bf16x3: the
    # FORWARD mode; the backward pass multiplies in exact fp32 in every mode)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    p = {k: t(v) for k, v in c["weights"].items()}
    fr = {"near": float(cfg.near), "far": float(cfg.far), "support": {k: t(v) for k, v in frame["support_fine"].items()}}
    R = min(cfg.R, 12)
    o, d = t(rays["rays_o"][:R]), t(rays["rays_d"][:R])
    lin = torch.linspace(0, 1, cfg.S, device=dev)
    z = (cfg.near * (1 - lin) + cfg.far * lin).expand(R, cfg.S)
    xyz = (o[:, None, :] + d[:, None, :] * z[..., None]).reshape(-1, 3).contiguous()
    dirs = d[:, None, :].expand(R, cfg.S, 3).reshape(-1, 3).contiguous()
    if ntrunc:
        xyz, dirs = xyz[:ntrunc].contiguous(), dirs[:ntrunc].contiguous()
    g = torch.Generator().manual_seed(3)
    G = torch.randn(xyz.shape[0], cfg.W, generator=g).to(dev)
    cot = torch.randn(xyz.shape[0], cfg.W, generator=g).to(dev)
    idx = r.knn(xyz, 8)[1].long()

    def grads(fn, dt):
        a, b, cc = (v.detach().to(dt).requires_grad_(True) for v in (xyz, dirs, G))
        out = fn(a, b, cc, dt)
        return out.detach(), torch.autograd.grad((out * cot.to(dt)).sum(), [a, b, cc])
    cast = lambda tree, dt: {k: (cast(v, dt) if isinstance(v, dict) else (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v)) for k, v in tree.items()}
    eager = lambda a, b, cc, dt: dr._point_branch(cast(p, dt), cast(fr, dt), a, b, cc, idx)
    hip = lambda a, b, cc, dt: dr.PointBranchFn.apply(a, b, cc, r, 8)
    o_hip, g_hip = grads(hip, torch.float32)
    o_ref, g_ref = grads(eager, torch.float32)
    _, g_64 = grads(eager, torch.float64)
    assert rel_err(o_hip.cpu().numpy(), o_ref.cpu().numpy()) < (1e-4 if precision == "bf16x3" else 5e-5)
    for name, a, b, c64 in zip(("xyz", "dirs", "G"), g_hip, g_ref, g_64):
        e_hip, e_ref = rel_err(a.cpu().numpy(), c64.cpu().numpy()), rel_err(b.cpu().numpy(), c64.cpu().numpy())
        print(case, precision, name, "hip vs fp64", e_hip, "| fp32 autograd vs fp64", e_ref)
        if precision == "fp32":
            assert e_hip < max(tol, 3 * e_ref), (name, e_hip, e_ref)
        else:
            # The derivative of a LeakyReLU network is piecewise constant: a recomputed forward that is 2^-22 off (split-FP16) flips the sign of the
            # ~2 in 10^7 pre-activations that lie that close to zero, and each flip moves ITS row's gradient by percents (fp32 autograd against fp64
            # does the same: e_ref).  So: all but a handful of rows to `tol`, the whole tensor in the L2 norm.
            err = (a.double() - c64).abs().max(1)[0] / c64.abs().max()
            bad = int((err > tol).sum())
            l2 = float((a.double() - c64).norm() / c64.norm())
            print("   rows beyond", tol, ":", bad, "of", err.numel(), "| L2-rel", l2)
            assert bad <= max(2, err.numel() // 200) and l2 < 1e-2, (name, bad, l2)


def _mv_setup(case, precision="fp32", R_max=10):
    from nerf_loc_amd.renderer import HipRenderer
    from tests.golden_cases import build_case
    c = _view_count_case(case) if case.startswith("v") else build_case(case)
    cfg, frame, rays = c["cfg"], c["frame"], c["rays"]
    dev = torch.device("cuda:0")
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, precision)
    r.load_weights({k: torch.from_numpy(v) for k, v in c["weights"].items()})
    r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    p = {k: t(v) for k, v in c["weights"].items()}
    fr = {k: t(frame[k]) for k in ("topk_Ks", "topk_poses", "topk_images", "feat_fine_src", "vis_featmaps")}
    fr.update({"near": float(cfg.near), "far": float(cfg.far)})
    R = min(cfg.R, R_max)
    o, d = t(rays["rays_o"][:R]), t(rays["rays_d"][:R])
    lin = torch.linspace(0, 1, cfg.S, device=dev)
    z = (cfg.near * (1 - lin) + cfg.far * lin).expand(R, cfg.S)
    xyz = (o[:, None, :] + d[:, None, :] * z[..., None]).reshape(-1, 3).contiguous()
    return cfg, r, p, fr, xyz, t(frame["pose"])


def _cast(tree, dt):
    return {k: (_cast(v, dt) if isinstance(v, dict) else (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v)) for k, v in tree.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["tiny_full", "offview", "c1", "w256s128"])
def test_mv_aggregate_backward_matches_autograd(case):
    """nl_mv_aggregate_backward against autograd of diff_render._mv_aggregate (frozen weights): d(sum G . cotangent) / d xyz."""
    cfg, r, p, fr, xyz, _ = _mv_setup(case)
    g = torch.Generator().manual_seed(5)
    cot = torch.randn(xyz.shape[0], cfg.W, generator=g).to(xyz.device)

    def grads(fn, dt):
        a = xyz.detach().to(dt).requires_grad_(True)
        out = fn(a, dt)
        return out.detach(), torch.autograd.grad((out * cot.to(dt)).sum(), a)[0]
    o_hip, g_hip = grads(lambda a, dt: dr.MvAggFn.apply(a, r)[0], torch.float32)
    o_ref, g_ref = grads(lambda a, dt: dr._mv_aggregate(_cast(p, dt), _cast(fr, dt), a)[0], torch.float32)
    _, g_64 = grads(lambda a, dt: dr._mv_aggregate(_cast(p, dt), _cast(fr, dt), a)[0], torch.float64)
    assert rel_err(o_hip.cpu().numpy(), o_ref.cpu().numpy()) < 5e-5
    e_hip, e_ref = rel_err(g_hip.cpu().numpy(), g_64.cpu().numpy()), rel_err(g_ref.cpu().numpy(), g_64.cpu().numpy())
    print(case, "g_xyz hip vs fp64", e_hip, "| fp32 autograd vs fp64", e_ref)
    assert e_hip < max(2e-4, 3 * e_ref), (e_hip, e_ref)


def _blend_eager(p, fr, xyz, agg, qc):
    """The colour blend of diff_render.render_rays_diff's eager branch as a function of (xyz, feature_agg, query centre)."""
    import torch.nn.functional as F
    _, mvf, mvv, _ = dr._mv_aggregate(p, fr, xyz)
    W = agg.shape[1]
    ang = dr._view_angles(xyz, qc, fr["topk_poses"][:, :3, 3])
    w0, b0 = p["rgb_blending_mlp.0.weight"], p["rgb_blending_mlp.0.bias"]
    Fd = mvf.shape[-1]
    xb = (F.linear(agg, w0[:, :W]) + b0).unsqueeze(1) + F.linear(mvf, w0[:, W:W + Fd]) + mvv * w0[:, W + Fd] + F.linear(ang, w0[:, W + Fd + 1:])
    xb = dr._lrelu(dr._lin(p, "rgb_blending_mlp.2", dr._lrelu(xb)))
    bw = F.softmax(dr._lin(p, "rgb_blending_mlp.4", xb).masked_fill(mvv == 0, -1e9), dim=1)
    return torch.sum(mvf[:, :, :3] * bw, dim=1)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["tiny_full", "offview", "c1", "w256s128"])
def test_blend_backward_matches_autograd(case):
    """nl_blend / nl_blend_backward against autograd of the eager blend: gradients w.r.t. the sample positions, feature_agg and the query centre."""
    cfg, r, p, fr, xyz, pose = _mv_setup(case)
    g = torch.Generator().manual_seed(6)
    fa = torch.randn(xyz.shape[0], cfg.W, generator=g).to(xyz.device)
    cot = torch.randn(xyz.shape[0], 3, generator=g).to(xyz.device)
    qc0 = pose[:3, 3].clone()

    def grads(fn, dt):
        a, b, q = (v.detach().to(dt).requires_grad_(True) for v in (xyz, fa, qc0))
        out = fn(a, b, q, dt)
        return out.detach(), torch.autograd.grad((out * cot.to(dt)).sum(), [a, b, q])
    o_hip, g_hip = grads(lambda a, b, q, dt: dr.BlendFn.apply(a, b, q, r), torch.float32)
    o_ref, g_ref = grads(lambda a, b, q, dt: _blend_eager(_cast(p, dt), _cast(fr, dt), a, b, q), torch.float32)
    _, g_64 = grads(lambda a, b, q, dt: _blend_eager(_cast(p, dt), _cast(fr, dt), a, b, q), torch.float64)
    assert rel_err(o_hip.cpu().numpy(), o_ref.cpu().numpy()) < 5e-5
    for name, a, b, c64 in zip(("xyz", "feature_agg", "query_center"), g_hip, g_ref, g_64):
        e_hip, e_ref = rel_err(a.cpu().numpy(), c64.cpu().numpy()), rel_err(b.cpu().numpy(), c64.cpu().numpy())
        print(case, name, "hip vs fp64", e_hip, "| fp32 autograd vs fp64", e_ref)
        assert e_hip < max(2e-4, 3 * e_ref), (name, e_hip, e_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("case,precision", [("tiny_full", "fp32"), ("c1", "fp32"), ("w128s64", "bf16x3"), ("w256s128", "bf16x3"), ("s192out", "bf16x3")])
def test_ray_unet_backward_matches_autograd(case, precision):
    """nl_ray_unet_backward (frozen weights) against autograd of the eager U-Net (diff_render._ray_unet), for every slab height the configs use
    (S = 16 ... 192; fused and unfused forward paths)."""
    from nerf_loc_amd.renderer import HipRenderer
    from tests.golden_cases import build_case
    c = build_case(case)
    cfg = c["cfg"]
    dev = torch.device("cuda:0")
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, precision)
    r.load_weights({k: torch.from_numpy(v) for k, v in c["weights"].items()})
    p = {k: torch.from_numpy(v).to(dev) for k, v in c["weights"].items()}
    R, S, W = 6, cfg.S_total, cfg.W
    g = torch.Generator().manual_seed(9)
    x = torch.randn(R * S, W, generator=g).to(dev)
    cot = torch.randn(R * S, W, generator=g).to(dev)

    def grads(fn, dt):
        a = x.detach().to(dt).requires_grad_(True)
        out = fn(a, dt)
        return out.detach(), torch.autograd.grad((out * cot.to(dt)).sum(), a)[0]
    eager = lambda a, dt: dr._ray_unet(_cast(p, dt), a.view(R, S, W).permute(0, 2, 1)).permute(0, 2, 1).reshape(R * S, W)
    o_hip, g_hip = grads(lambda a, dt: dr.UnetFn.apply(a, r), torch.float32)
    o_ref, g_ref = grads(eager, torch.float32)
    _, g_64 = grads(eager, torch.float64)
    assert rel_err(o_hip.cpu().numpy(), o_ref.cpu().numpy()) < (1e-4 if precision == "bf16x3" else 2e-5)
    e_hip, e_ref = rel_err(g_hip.cpu().numpy(), g_64.cpu().numpy()), rel_err(g_ref.cpu().numpy(), g_64.cpu().numpy())
    print(case, precision, "g_x hip vs fp64", e_hip, "| fp32 autograd vs fp64", e_ref)
    assert e_hip < max(1e-4, 3 * e_ref), (e_hip, e_ref)


@pytest.mark.gpu
def test_backward_entry_points_walk_large_batches_in_chunks():
    """Every backward entry point processes N samples (R rays) in pieces that fit the caller's workspace: a workspace sized for a third of the batch
    must give the same gradients as one that holds it all (per-sample stages: bit-identical; the support-side atomics are not involved)."""
    cfg, r, p, fr, xyz, pose = _mv_setup("c1", "bf16x3", R_max=12)
    N, W, S = xyz.shape[0], cfg.W, cfg.S_total
    g = torch.Generator().manual_seed(11)
    rnd = lambda *shape: torch.randn(*shape, generator=g).to(xyz.device)
    G, cotW, cot3, dirs = rnd(N, W), rnd(N, W), rnd(N, 3), torch.nn.functional.normalize(rnd(N, 3), dim=1)
    third = N // 3 + 1
    a = r.point_mlp_backward(xyz, dirs, G, cotW)
    b = r.point_mlp_backward(xyz, dirs, G, cotW, workspace_samples=third)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.equal(r.mv_aggregate_backward(xyz, cotW), r.mv_aggregate_backward(xyz, cotW, workspace_samples=third))
    a = r.blend_backward(xyz, pose[:3, 3], G, cot3)
    b = r.blend_backward(xyz, pose[:3, 3], G, cot3, workspace_samples=third)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and rel_err(b[2].cpu().numpy(), a[2].cpu().numpy()) < 1e-5
    x = rnd(12 * S, W)
    assert torch.equal(r.ray_unet_backward(x, cotW[: 12 * S]), r.ray_unet_backward(x, cotW[: 12 * S], workspace_rays=5))


# --------------------------------------------------------------------------------------------- training: weight gradients (nl_*_backward_train)
POINT_PARAMS = ["ray_diff_fc.0.weight", "ray_diff_fc.0.bias", "ray_diff_fc.2.weight", "ray_diff_fc.2.bias",
                "base_mlp.0.weight", "base_mlp.0.bias", "base_mlp.2.weight", "base_mlp.2.bias", "base_mlp.4.weight", "base_mlp.4.bias",
                "base_mlp_attn.w_qs.weight", "base_mlp_attn.w_ks.weight", "base_mlp_attn.w_vs.weight", "base_mlp_attn.fc.weight",
                "base_mlp_attn.layer_norm.weight", "base_mlp_attn.layer_norm.bias"]


def _assert_param_grads(got, ref32, ref64, tol, what, exact_forward):
    """Every tensor against the fp64 autograd gradient, relative to that tensor's largest entry.  exact_forward (fp32 mode on a small network): to `tol`,
    or 5x what fp32 autograd manages.  Otherwise the comparison has to live with the piecewise-constant derivative of a LeakyReLU network: the
    rounding of the forward pass decides the sign of the few pre-activations within ~1e-7 of zero (for fp32 autograd as well: e_ref), and one
    flipped unit changes that sample's whole contribution to its layer's weight gradient and to everything upstream (percents of single entries
    at test sizes, where a gradient sums ~10^4 rows) — so: every tensor in the L2 norm, and no entry off by more than a few percent."""
    gmax = max(float(v.abs().max()) for v in ref64.values())
    for n in got:
        a, c64 = got[n].double().cpu(), ref64[n].double().cpu()
        # (a gradient that is mathematically zero — the bias of the softmax logits — holds rounding noise in every implementation: measured against
        # 1e-6 of the step's largest gradient instead of against itself)
        den = max(float(c64.abs().max()), 1e-6 * gmax)
        e_hip, e_ref = float((a - c64).abs().max()) / den, float((ref32[n].double().cpu() - c64).abs().max()) / den
        rows = (a - c64).abs().reshape(a.shape[0], -1).max(1)[0] / den
        bad, l2 = int((rows > tol).sum()), float((a - c64).norm() / max(float(c64.norm()), 1e-6 * gmax))
        print(what, n, tuple(a.shape), "hip vs fp64", e_hip, "| fp32 autograd vs fp64", e_ref, "| rows beyond tol", bad, "of", rows.numel(), "| L2-rel", l2)
        if exact_forward:
            assert e_hip < max(tol, 5 * e_ref), (n, e_hip, e_ref)
        else:
            assert l2 < 1e-2 and e_hip < 5e-2, (n, bad, l2, e_hip)


@pytest.mark.gpu
@pytest.mark.parametrize("case,precision,chunk", [("tiny_full", "fp32", None), ("c1", "fp32", None), ("c1", "bf16x3", 100), ("w256s128", "bf16x3", None),
                                                  ("w128s64", "bf16x3", None)])
def test_point_branch_weight_gradients_match_autograd(case, precision, chunk):
    """nl_point_mlp_backward_train: the gradients of the branch's 16 parameter tensors and of the support table's features, against autograd of the
    eager restatement (fp64).  chunk: a workspace for fewer samples than N — the weight gradients accumulate over the chunks."""
    from nerf_loc_amd.renderer import HipRenderer
    from tests.golden_cases import build_case
    c = build_case(case)
    cfg, frame, rays = c["cfg"], c["frame"], c["rays"]
    dev = torch.device("cuda:0")
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, precision)
    r.load_weights({k: torch.from_numpy(v) for k, v in c["weights"].items()})
    r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    p = {k: t(v) for k, v in c["weights"].items()}
    fr = {"near": float(cfg.near), "far": float(cfg.far), "support": {k: t(v) for k, v in frame["support_fine"].items()}}
    R = min(cfg.R, 12)
    o, d = t(rays["rays_o"][:R]), t(rays["rays_d"][:R])
    lin = torch.linspace(0, 1, cfg.S, device=dev)
    z = (cfg.near * (1 - lin) + cfg.far * lin).expand(R, cfg.S)
    xyz = (o[:, None, :] + d[:, None, :] * z[..., None]).reshape(-1, 3).contiguous()
    dirs = d[:, None, :].expand(R, cfg.S, 3).reshape(-1, 3).contiguous()
    g = torch.Generator().manual_seed(5)
    G = torch.randn(xyz.shape[0], cfg.W, generator=g).to(dev)
    cot = torch.randn(xyz.shape[0], cfg.W, generator=g).to(dev)
    d2, idx = r.knn(xyz, 8)

    def eager(dt):
        pp = {k: v.detach().to(dt).requires_grad_(k in POINT_PARAMS) for k, v in p.items()}
        sp = {k: v.to(dt) for k, v in fr["support"].items()}
        sp["feature"] = sp["feature"].detach().requires_grad_(True)
        out = dr._point_branch(pp, {"near": fr["near"], "far": fr["far"], "support": sp}, xyz.to(dt), dirs.to(dt), G.to(dt), idx.long())
        gs = torch.autograd.grad((out * cot.to(dt)).sum(), [pp[n] for n in POINT_PARAMS] + [sp["feature"]])
        return dict(zip(POINT_PARAMS + ["support.feature"], gs))
    ref32, ref64 = eager(torch.float32), eager(torch.float64)
    tg = r.train_grads(POINT_PARAMS, support_feature=True)
    gx, gd, gg = r.point_mlp_backward(xyz, dirs, G, cot, K=8, knn=(d2, idx), train=tg, workspace_samples=chunk)
    gx0, gd0, gg0 = r.point_mlp_backward(xyz, dirs, G, cot, K=8, knn=(d2, idx), workspace_samples=chunk)
    # the input gradients do not depend on the training outputs.  (Not bit for bit: with frozen weights the encode columns' way back is the lane-per-row kernel
    # (fp64 sin / cos recurrence, another summation order), and for W = 128 / 256, K = 8 the forward is the fused keep kernel and the four row products are one chain
    # launch: pt_forward_keep_fused, point_bwd.hip)
    lim = 1e-4 if case in ("w256s128", "w128s64") else 2e-5
    for a, b in ((gx, gx0), (gd, gd0), (gg, gg0)):
        assert float((a - b).norm() / b.norm()) < lim, "the input gradients do not depend on the training outputs"
    got = {k: v.clone() for k, v in tg.weights.items()}
    got["support.feature"] = tg.support_feature.clone()
    _assert_param_grads(got, ref32, ref64, 3e-4, f"{case}/{precision}", exact_forward=case not in ("w256s128", "w128s64"))
    # a second call ADDS
    r.point_mlp_backward(xyz, dirs, G, cot, K=8, knn=(d2, idx), train=tg, workspace_samples=chunk)
    assert rel_err(tg.weights["base_mlp.2.weight"].cpu().numpy(), 2 * got["base_mlp.2.weight"].cpu().numpy()) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("case,precision,chunk", [("tiny_full", "fp32", None), ("offview", "fp32", None), ("c1", "fp32", 70), ("c1", "bf16x3", None),
                                                  ("v16", "bf16x3", None), ("v12", "bf16x3", 50), ("w128s64", "bf16x3", None), ("v5c64", "bf16x3", None),
                                                  ("v4c100", "bf16x3", None), ("v3c8", "bf16x3", None)])
def test_mv_aggregate_weight_gradients_match_autograd(case, precision, chunk):
    """nl_mv_aggregate_backward_train: gradients of out_fc, of the four NeuRay decoders (24 tensors), of the support feature maps (grid_sample's backward)
    and of the DepthFusionNet maps against autograd of diff_render._mv_aggregate in fp64."""
    cfg, r, p, fr, xyz, _ = _mv_setup(case, precision)
    g = torch.Generator().manual_seed(11)
    cot = torch.randn(xyz.shape[0], cfg.W, generator=g).to(xyz.device)
    names = list(dr.MV_PARAMS)

    def eager(dt):
        pp = {k: v.detach().to(dt).requires_grad_(k in names) for k, v in p.items()}
        ff = _cast(fr, dt)
        ff["feat_fine_src"] = ff["feat_fine_src"].detach().requires_grad_(True)
        ff["vis_featmaps"] = ff["vis_featmaps"].detach().requires_grad_(True)
        out = dr._mv_aggregate(pp, ff, xyz.to(dt))[0]
        gs = torch.autograd.grad((out * cot.to(dt)).sum(), [pp[n] for n in names] + [ff["feat_fine_src"], ff["vis_featmaps"]])
        return dict(zip(names + ["feat_fine_src", "vis_featmaps"], gs))
    ref32, ref64 = eager(torch.float32), eager(torch.float64)
    tg = r.train_grads(names, feat_maps=True, vis_featmaps=True)
    gx = r.mv_aggregate_backward(xyz, cot, train=tg, workspace_samples=chunk)
    # (training scatters into the maps with a wave per sample; frozen maps take eight samples per wave — mv_geom_backward8_kernel: the same sums in another order)
    gx0 = r.mv_aggregate_backward(xyz, cot, workspace_samples=chunk)
    assert float((gx - gx0).norm() / gx0.norm()) < 2e-5, "the input gradient does not depend on the training outputs"
    got = {k: v.clone() for k, v in tg.weights.items()}
    got["feat_fine_src"], got["vis_featmaps"] = tg.feat_maps.clone(), tg.vis_featmaps.contiguous().clone()
    _assert_param_grads(got, ref32, ref64, 3e-4, f"{case}/{precision}", exact_forward=True)


@pytest.mark.gpu
@pytest.mark.parametrize("case,precision,chunk", [("tiny_full", "fp32", None), ("offview", "fp32", None), ("c1", "fp32", 70), ("c1", "bf16x3", None),
                                                  ("v16", "bf16x3", None), ("w128s64", "bf16x3", None)])
def test_blend_weight_gradients_match_autograd(case, precision, chunk):
    """nl_blend_backward_train: gradients of rgb_blending_mlp (layer 1's feature columns through the blend-projected maps, whose gradient the library returns
    as a map), the decoders, the feature maps and the DepthFusionNet maps against autograd of the eager blend in fp64."""
    import torch.nn.functional as F
    cfg, r, p, fr, xyz, pose = _mv_setup(case, precision)
    g = torch.Generator().manual_seed(12)
    fa = torch.randn(xyz.shape[0], cfg.W, generator=g).to(xyz.device)
    cot = torch.randn(xyz.shape[0], 3, generator=g).to(xyz.device)
    qc = pose[:3, 3].clone()
    names = list(dr.BLEND_PARAMS)
    W, C = cfg.W, cfg.C

    def eager(dt):
        pp = {k: v.detach().to(dt).requires_grad_(k in names) for k, v in p.items()}
        ff = _cast(fr, dt)
        ff["feat_fine_src"] = ff["feat_fine_src"].detach().requires_grad_(True)
        ff["vis_featmaps"] = ff["vis_featmaps"].detach().requires_grad_(True)
        out = _blend_eager(pp, ff, xyz.to(dt), fa.to(dt), qc.to(dt))
        gs = torch.autograd.grad((out * cot.to(dt)).sum(), [pp[n] for n in names] + [ff["feat_fine_src"], ff["vis_featmaps"]])
        return dict(zip(names + ["feat_fine_src", "vis_featmaps"], gs))
    ref32, ref64 = eager(torch.float32), eager(torch.float64)
    # the HIP node + the per-frame projection's graph, as diff_render.render_rays_diff builds them
    pp = {k: v.detach().clone().requires_grad_(k in names) for k, v in p.items()}
    feat, vis = fr["feat_fine_src"].detach().clone().requires_grad_(True), fr["vis_featmaps"].detach().clone().requires_grad_(True)
    pmaps = F.linear(feat, pp["rgb_blending_mlp.0.weight"][:, W + 3:W + 3 + C])
    out = dr.BlendTrainFn.apply(xyz, fa, qc, pmaps, vis, r, *[pp[n] for n in names])
    gs = torch.autograd.grad((out * cot).sum(), [pp[n] for n in names] + [feat, vis])
    got = dict(zip(names + ["feat_fine_src", "vis_featmaps"], gs))
    _assert_param_grads(got, ref32, ref64, 3e-4, f"{case}/{precision}", exact_forward=True)


@pytest.mark.gpu
@pytest.mark.parametrize("case,precision,chunk", [("tiny_full", "fp32", None), ("c1", "fp32", 2), ("w128s64", "bf16x3", None), ("s192out", "bf16x3", None)])
def test_ray_unet_weight_gradients_match_autograd(case, precision, chunk):
    """nl_ray_unet_backward_train: the 28 U-Net tensors (convolutions: one split-K product per tap with the input rows shifted inside each ray; transposed
    convolutions: the three taps from the even / odd output phases; LayerNorm([C, L]) tables: sums over the rays) against autograd of the eager U-Net in fp64."""
    from nerf_loc_amd.renderer import HipRenderer
    from tests.golden_cases import build_case
    c = build_case(case)
    cfg = c["cfg"]
    dev = torch.device("cuda:0")
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, precision)
    r.load_weights({k: torch.from_numpy(v) for k, v in c["weights"].items()})
    p = {k: torch.from_numpy(v).to(dev) for k, v in c["weights"].items()}
    R, S, W = 6, cfg.S_total, cfg.W
    g = torch.Generator().manual_seed(19)
    x = torch.randn(R * S, W, generator=g).to(dev)
    cot = torch.randn(R * S, W, generator=g).to(dev)
    names = list(dr.UNET_PARAMS)

    def eager(dt):
        pp = {k: v.detach().to(dt).requires_grad_(k in names) for k, v in p.items()}
        out = dr._ray_unet(pp, x.to(dt).view(R, S, W).permute(0, 2, 1)).permute(0, 2, 1).reshape(R * S, W)
        return dict(zip(names, torch.autograd.grad((out * cot.to(dt)).sum(), [pp[n] for n in names])))
    ref32, ref64 = eager(torch.float32), eager(torch.float64)
    tg = r.train_grads(names)
    gx = r.ray_unet_backward(x, cot, train=tg, workspace_rays=chunk)
    assert torch.equal(gx, r.ray_unet_backward(x, cot, workspace_rays=chunk))
    _assert_param_grads({k: v.clone() for k, v in tg.weights.items()}, ref32, ref64, 3e-4, f"{case}/{precision}", exact_forward=True)


def _view_count_case(name):
    """tiny scenes with 16 / 12 / 3 support views ("v16", "v12", "v3w64": the 16- and 4-view instantiations of the kernels whose shared-memory footprint and
    view loops depend on the bucket — the training scatter of mv_geom_backward_kernel keeps one slot per view in LDS: 118 KB at 16)."""
    from nerf_loc_amd.synth import make_frame, make_rays, make_weights
    from tests.golden_cases import CASES
    V = int("".join(ch for ch in name[1:3] if ch.isdigit()))
    C = int(name.split("c")[1]) if "c" in name[1:] else 192      # "v5c64": feature maps narrower than the reference's 192 channels
    cfg = CASES["tiny_full"][0].replace(name=name, V=V, W=64 if name.endswith("w64") else 32, C=C, seed=900 + V)
    frame = make_frame(cfg)
    return {"cfg": cfg, "frame": frame, "rays": make_rays(cfg, frame), "weights": make_weights(cfg)}


@pytest.mark.gpu
@pytest.mark.parametrize("case,precision,train,chunk", [("tiny_full", "fp32", False, None), ("tiny_full", "fp32", False, 4), ("fewpts", "fp32", True, None),
                                                        ("tiny_full", "fp32", True, 5), ("offview", "bf16x3", False, 2), ("c1", "fp32", True, 3),
                                                        ("c1", "fp32", True, None), ("c1", "bf16x3", True, None), ("w128s64", "bf16x3", False, None), ("w128s64", "fp32", True, None),
                                                        ("s192out", "fp32", True, 2), ("w256s128", "fp32", True, None),
                                                        ("v16", "bf16x3", True, None), ("v12", "bf16x3", True, 3), ("v3w64", "bf16x3", True, None)])
def test_whole_path_backward_matches_the_stage_nodes(case, precision, train, chunk):
    """nl_render_rays_backward (one call for the whole path: RenderFn) against the chain of per-stage autograd nodes + eager heads: the same
    gradients w.r.t. the rays, the query pose and — train — every parameter tensor, the maps and the support features."""
    from nerf_loc_amd.renderer import HipRenderer
    from tests.golden_cases import build_case
    c = _view_count_case(case) if case.startswith("v") else build_case(case)
    cfg, frame, rays = c["cfg"], c["frame"], c["rays"]
    dev = torch.device("cuda:0")
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, precision)
    r.load_weights({k: torch.from_numpy(v) for k, v in c["weights"].items()})
    r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    R = min(cfg.R, 10)
    o0, d0 = t(rays["rays_o"][:R]), t(rays["rays_d"][:R])
    lin = torch.linspace(0, 1, cfg.S_total, device=dev)
    z = (cfg.near * (1 - lin) + cfg.far * lin).expand(R, cfg.S_total).contiguous()
    g = torch.Generator().manual_seed(23)
    cot = {k: torch.randn(*shp, generator=g).to(dev) for k, shp in (("rgb", (R, 3)), ("depth", (R,)), ("depth_uncertainty", (R,)), ("feat", (R, cfg.C)),
                                                                    ("weights", (R, cfg.S_total)))}
    knn = lambda q: r.knn(q, 8)[1]

    def run(whole):
        p = {k: t(v).requires_grad_(train) for k, v in c["weights"].items()}
        fr = {k: t(frame[k]) for k in ("topk_Ks", "topk_poses", "topk_images", "feat_fine_src", "vis_featmaps")}
        sp = {k: t(v) for k, v in frame["support_fine"].items()}
        if train:
            fr["feat_fine_src"].requires_grad_(True); fr["vis_featmaps"].requires_grad_(True); sp["feature"].requires_grad_(True)
        fr.update({"near": float(cfg.near), "far": float(cfg.far), "support": sp})
        o, d, pose = o0.clone().requires_grad_(True), d0.clone().requires_grad_(True), t(frame["pose"]).clone().requires_grad_(True)
        use_beta = train and chunk is None    # (the uncertainty head exists in the keep / kept pair and in the stage nodes' eager tail)
        out = dr.render_rays_diff(p, fr, o, d, z, pose, knn, frozen_renderer=None if train else r, train_renderer=r if train else None, whole_path=whole,
                                  beta=use_beta)
        loss = sum((out[k] * cot[k]).sum() for k in cot)
        if use_beta:
            loss = loss + (out["beta"] * cot["depth"]).sum()
        leaves = {"rays_o": o, "rays_d": d, "pose": pose}
        if train:
            leaves.update({n: p[n] for n in dr.RENDER_PARAMS + (("beta_mlp.0.weight", "beta_mlp.0.bias") if use_beta else ())})
            leaves.update({"feat_fine_src": fr["feat_fine_src"], "vis_featmaps": fr["vis_featmaps"], "support.feature": sp["feature"]})
        gs = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
        return {k: v.detach() for k, v in out.items()}, dict(zip(leaves.keys(), gs))
    keep_bytes = dr.KEEP_BYTES
    if chunk is not None:   # the chunking pair (fused forward + nl_render_rays_backward over a small workspace) instead of the keep / kept pair
        orig = r.render_rays_backward
        r.render_rays_backward = lambda *a, **k: orig(*a, workspace_rays=chunk, **k)
        dr.KEEP_BYTES = 0
    try:
        o_w, g_w = run(True)
    finally:
        dr.KEEP_BYTES = keep_bytes
    o_n, g_n = run(False)
    for k in cot:   # fused inference kernels against the stage entry points: the configured precision's tolerance
        assert rel_err(o_w[k].cpu().numpy(), o_n[k].cpu().numpy()) < (2e-4 if precision == "bf16x3" else 2e-5), k
    gmax = max(float(v.abs().max()) for v in g_n.values() if v is not None)
    worst, errs = ("", 0.0), {}
    for k, b in g_n.items():
        a = g_w[k]
        if b is None:
            assert a is None or float(a.abs().max()) <= 1e-6 * gmax, k
            continue
        assert a is not None, k
        e = float((a - b).abs().max() / max(float(b.abs().max()), 1e-6 * gmax))
        if e > worst[1]: worst = (k, e)
        errs[k] = e
    # fp32 mode: both paths recompute exactly the same forward — summation orders only.  bf16x3: the stage nodes recompute each stage from the
    # split-bf16 forward's saved inputs, the whole-path call recomputes everything from the rays in split-FP16: inputs 1e-5 apart flip the odd
    # LeakyReLU sign, single entries move by a percent
    assert worst[1] < (2e-3 if precision == "fp32" else 3e-2), worst
    assert float(np.median(list(errs.values()))) < (1e-5 if precision == "fp32" else 2e-3)
    print(case, precision, "train" if train else "frozen", "worst", worst, "median", float(np.median(list(errs.values()))))


@pytest.mark.gpu
def test_nodes_refuse_a_backward_pass_against_replaced_state():
    """The library nodes recompute from the renderer's current weights / frame: replacing either between forward and backward raises instead of
    silently differentiating another function."""
    cfg, r, p, fr, xyz, pose = _mv_setup("tiny_full")
    from tests.golden_cases import build_case
    frame = build_case("tiny_full")["frame"]
    a = xyz.clone().requires_grad_(True)
    out = dr.MvAggFn.apply(a, r)[0]
    r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
    with pytest.raises(RuntimeError, match="replaced between the forward and the backward"):
        out.sum().backward()


@pytest.mark.gpu
@pytest.mark.parametrize("case,precision", [("tiny_full", "fp32"), ("tiny_full", "bf16x3"), ("fewpts", "bf16x3"), ("c1", "bf16x3"), ("w128s64", "bf16x3"), ("s192out", "bf16x3")])
def test_no_call_writes_past_its_workspace(case, precision):
    """Every workspace is carved by the library from a caller buffer of exactly the size its *_workspace_bytes query returned: with a canary region
    behind each workspace AND between the buffers carved from it (nl_debug_bump_gap), the forward stages, every backward / training entry point and the
    whole-path pairs leave all of them intact (the buffer that the U-Net's LayerNorm rows overflowed at W = 32 was the last one of its workspace in the
    stage call and an inner one in the whole-path call)."""
    from nerf_loc_amd.renderer import HipRenderer
    from tests.golden_cases import build_case
    c = build_case(case)
    cfg, frame, rays = c["cfg"], c["frame"], c["rays"]
    dev = torch.device("cuda:0")
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, precision)
    r.set_guard(1 << 20, gap_bytes=4096)   # a canary behind every workspace and between the buffers carved from it
    try:
        _guarded_calls(r, c, cfg, frame, rays, dev)
    finally:
        r.set_guard(0)


def _guarded_calls(r, c, cfg, frame, rays, dev):
    r.load_weights({k: torch.from_numpy(v) for k, v in c["weights"].items()})
    r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    R, S, W = min(cfg.R, 9), cfg.S_total, cfg.W
    o, d = t(rays["rays_o"][:R]), t(rays["rays_d"][:R])
    lin = torch.linspace(0, 1, S, device=dev)
    z = (cfg.near * (1 - lin) + cfg.far * lin).expand(R, S).contiguous()
    xyz = (o[:, None, :] + d[:, None, :] * z[..., None]).reshape(-1, 3).contiguous()
    dirs = d[:, None, :].expand(R, S, 3).reshape(-1, 3).contiguous()
    N = R * S
    g = torch.Generator().manual_seed(41)
    rnd = lambda *shp: torch.randn(*shp, generator=g).to(dev)
    qc = t(frame["pose"])[:3, 3]
    names = list(dr.RENDER_PARAMS)
    calls = 0
    r.render_rays(o, d, qc, z_vals=z); calls += r.check_guards()
    mv, _, _, _ = r.mv_aggregate(xyz, qc, want_raw=False); calls += r.check_guards()
    fa, d2, idx = r.point_mlp(xyz, dirs, mv, K=8); calls += r.check_guards()
    geo = r.ray_unet(fa); calls += r.check_guards()
    r.blend(xyz, qc, fa); calls += r.check_guards()
    for chunk in (None, 3):
        tg = r.train_grads(names, support_feature=True, feat_maps=True, vis_featmaps=True, blend_feat_maps=True)
        r.mv_aggregate_backward(xyz, rnd(N, W), train=tg, workspace_samples=None if chunk is None else chunk * S); calls += r.check_guards()
        r.point_mlp_backward(xyz, dirs, mv, rnd(N, W), K=8, knn=(d2, idx), train=tg, workspace_samples=None if chunk is None else chunk * S); calls += r.check_guards()
        r.ray_unet_backward(fa, rnd(N, W), train=tg, workspace_rays=chunk); calls += r.check_guards()
        r.blend_backward(xyz, qc, fa, rnd(N, 3), train=tg, workspace_samples=None if chunk is None else chunk * S); calls += r.check_guards()
        r.render_rays_backward(o, d, z, qc, g_rgb=rnd(R, 3), g_feat=rnd(R, cfg.C), g_depth=rnd(R), g_weights=rnd(R, S), want_g_query_center=True, train=tg,
                               workspace_rays=chunk); calls += r.check_guards()
        r.render_rays_backward(o, d, z, qc, g_rgb=rnd(R, 3), workspace_rays=chunk); calls += r.check_guards()
    for train in (False, True):
        out, state = r.render_rays_keep(o, d, z, qc, train=train)
        tg = r.train_grads(names, support_feature=True, feat_maps=True, vis_featmaps=True, blend_feat_maps=True) if train else None
        r.render_rays_backward_kept(state, g_rgb=rnd(R, 3), g_feat=rnd(R, cfg.C), train=tg); calls += r.check_guards()
    assert calls >= 15 and r.gaps_checked > 300   # (guarded workspaces / gap regions between their buffers checked)


@pytest.mark.gpu
def test_graphed_refinement_steps_equal_the_ungraphed_ones():
    """RenderFn replays the keep / kept pair of a repeated batch shape as two HIP graphs (renderer.GraphedKeep) from the second step against one frame on:
    the same outputs and gradients, bit for bit, as the ungraphed pair — over steps with different rays and cotangents, and dropped when the frame changes."""
    from nerf_loc_amd.renderer import HipRenderer
    from tests.golden_cases import build_case
    c = build_case("c1")
    cfg, frame, rays = c["cfg"], c["frame"], c["rays"]
    dev = torch.device("cuda:0")
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, "bf16x3")
    r.load_weights({k: torch.from_numpy(v) for k, v in c["weights"].items()})
    setf = lambda: r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far,
                               frame["support_fine"])
    setf()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    R = 24
    lin = torch.linspace(0, 1, cfg.S_total, device=dev)
    z = (cfg.near * (1 - lin) + cfg.far * lin).expand(R, cfg.S_total).contiguous()
    p = {k: t(v) for k, v in c["weights"].items()}
    fr = {k: t(frame[k]) for k in ("topk_Ks", "topk_poses", "topk_images", "feat_fine_src", "vis_featmaps")}
    fr.update({"near": float(cfg.near), "far": float(cfg.far), "support": {k: t(v) for k, v in frame["support_fine"].items()}})
    g = torch.Generator().manual_seed(51)

    def step(i, graphs):
        dr.USE_GRAPHS = graphs
        o = t(rays["rays_o"][i * R:(i + 1) * R]).clone().requires_grad_(True)
        d = t(rays["rays_d"][i * R:(i + 1) * R]).clone().requires_grad_(True)
        pose = t(frame["pose"]).clone().requires_grad_(True)
        out = dr.render_rays_diff(p, fr, o, d, z, pose, lambda q: r.knn(q, 8)[1], frozen_renderer=r)
        cf, cr = torch.randn(R, cfg.C, generator=torch.Generator().manual_seed(100 + i)).to(dev), torch.randn(R, 3, generator=torch.Generator().manual_seed(200 + i)).to(dev)
        loss = (out["feat"] * cf).sum() + (out["rgb"] * cr).sum() + out["depth"].sum()
        gs = torch.autograd.grad(loss, [o, d, pose])
        return [out[k].detach().clone() for k in ("rgb", "feat", "depth", "weights", "mask")] + [x.clone() for x in gs]
    try:
        ref = [step(i, False) for i in range(4)]
        got = [step(i, True) for i in range(4)]      # step 0: first sight of the shape (ungraphed), steps 1..3: capture + replays
        assert r._graphs[(R, False)].fwd is not None and r._graphs[(R, False)].bwd is not None, "the graphs were captured"
        names = ("rgb", "feat", "depth", "weights", "mask", "g_o", "g_d", "g_pose")
        for i, (a, b) in enumerate(zip(ref, got)):
            for nm, x, y in zip(names, a, b):
                assert torch.equal(x, y), (i, nm, float((x.float() - y.float()).abs().max()), float(x.float().abs().max()))
        setf()                                        # a new frame: the graphs of the old one are dropped
        again = step(1, True)
        assert (R, False) in r._graphs and r._graphs[(R, False)] == "seen"
        for x, y in zip(ref[1], again):
            assert torch.equal(x, y)
    finally:
        dr.USE_GRAPHS = True


@pytest.mark.gpu
def test_random_scenes_training_gradients_match_eager_autograd():
    """tools/grad_fuzz.py on a few random scenes: widths 32 ... 256, 1 ... 16 views, feature maps of 8 ... 192 channels, support sets smaller than K, odd
    ray counts — every gradient of the whole-path training node in the parity mode and in the fp32 mode against autograd of the eager graph (the fixed cases
    all use 192-channel maps; for C <= 123 a transposed product changes kernels, and its weight stream was missing until this check existed)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("grad_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "grad_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # round 4: 60 seeded scenes in the driver-run suite (was 8)
    assert mod.run(int(os.environ.get("NERFLOC_FUZZ_GRAD", "60")), 7, verbose=False) < 0.5    # (the criteria themselves — per tensor, against fp64 autograd with fp32 autograd as the yardstick — are asserted inside;
    # the worst tensor of 60 scenes is a LayerNorm table in front of a MaxPool at 7.8e-2, exactly where fp32 autograd of the same graph is: scene 48, S = 144)


@pytest.mark.gpu
def test_training_with_feature_widths_that_are_not_multiples_of_four_takes_the_eager_graph():
    """The weight-gradient products read their operands as 16-byte rows: with a feature width C, C % 4 != 0 (the forward and the frozen-weight gradients take it)
    the library's training entry points answer NL_ERR_UNSUPPORTED.  Round 4: that refusal used to surface inside `loss.backward()`, with no graph
    left to fall back to; now `HipRenderer.train_capable()` is asked BEFORE the forward chooses its autograd nodes and the step runs on the eager graph — finite
    gradients for every parameter — while a direct call of the training entry point still refuses loudly."""
    from nerf_loc_amd.renderer import HipRenderer
    from nerf_loc_amd.synth import SceneConfig, make_frame, make_rays, make_weights
    cfg = SceneConfig("c31", R=8, S=32, W=64, V=5, H=32, Wimg=56, C=31, seed=77)
    frame, weights = make_frame(cfg), make_weights(cfg)
    rays = make_rays(cfg, frame)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, "bf16x3")
    assert not r.train_capable()
    assert HipRenderer(64, 32, 32, "bf16x3").train_capable()
    r.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
    r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
    fr = {k: t(frame[k]) for k in ("topk_Ks", "topk_poses", "topk_images", "feat_fine_src", "vis_featmaps")}
    fr.update({"near": float(cfg.near), "far": float(cfg.far), "support": {k: t(v) for k, v in frame["support_fine"].items()}})
    lin = torch.linspace(0, 1, cfg.S_total, device=dev)
    z = (cfg.near * (1 - lin) + cfg.far * lin).expand(8, cfg.S_total).contiguous()
    for train in (False, True):
        p = {k: t(v).requires_grad_(train) for k, v in weights.items()}
        o, d, pose = t(rays["rays_o"][:8]).requires_grad_(True), t(rays["rays_d"][:8]).requires_grad_(True), t(frame["pose"]).clone().requires_grad_(True)
        out = dr.render_rays_diff(p, fr, o, d, z, pose, lambda q: r.knn(q, 8)[1], frozen_renderer=None if train else r, train_renderer=r if train else None, whole_path=True)
        loss = out["rgb"].sum() + out["feat"].sum()
        leaves = [o, d, pose] + ([p[n] for n in dr.RENDER_PARAMS] if train else [])
        g = torch.autograd.grad(loss, leaves, allow_unused=True)
        assert all(x is None or torch.isfinite(x).all() for x in g)
        assert all(x is not None for x in g[:3])
        if train:
            assert sum(x is not None for x in g[3:]) >= 80   # the eager graph reached the parameters
    # the library itself still says so when asked directly (the keep / kept pair a training step takes)
    tg = r.train_grads(list(dr.RENDER_PARAMS), support_feature=True, feat_maps=True, vis_featmaps=True, blend_feat_maps=True)
    with pytest.raises(RuntimeError, match="unsupported shape or option"):
        kept = r.render_rays_keep(t(rays["rays_o"][:8]), t(rays["rays_d"][:8]), z, t(frame["pose"])[:3, 3], train=True)
        assert kept is not None
        r.render_rays_backward_kept(kept[1], g_rgb=torch.ones(8, 3, device=dev), g_feat=torch.ones(8, cfg.C, device=dev), train=tg)


@pytest.mark.gpu
def test_render_node_backward_can_run_twice():
    """`RenderFn.backward` consumed its kept activations / graph and then fell through to a recompute branch that indexed tensors the node had
    never saved (IndexError on the second `autograd.grad` of one forward, e.g. retain_graph=True).  The second pass now recomputes from the saved rays
    (the library searches the neighbours again when none were saved): same gradients as the first pass to the recompute's rounding — for the kept pair,
    for the graphed pair and for the chunking path."""
    from nerf_loc_amd.renderer import HipRenderer
    from tests.golden_cases import build_case
    c = build_case("c1")
    cfg, frame, rays = c["cfg"], c["frame"], c["rays"]
    dev = torch.device("cuda:0")
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, "fp32")
    r.load_weights({k: torch.from_numpy(v) for k, v in c["weights"].items()})
    r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    R = 16
    lin = torch.linspace(0, 1, cfg.S_total, device=dev)
    z = (cfg.near * (1 - lin) + cfg.far * lin).expand(R, cfg.S_total).contiguous()
    p = {k: t(v) for k, v in c["weights"].items()}
    fr = {k: t(frame[k]) for k in ("topk_Ks", "topk_poses", "topk_images", "feat_fine_src", "vis_featmaps")}
    fr.update({"near": float(cfg.near), "far": float(cfg.far), "support": {k: t(v) for k, v in frame["support_fine"].items()}})
    cf = torch.randn(R, cfg.C, generator=torch.Generator().manual_seed(3)).to(dev)

    def twice():
        o = t(rays["rays_o"][:R]).clone().requires_grad_(True)
        d = t(rays["rays_d"][:R]).clone().requires_grad_(True)
        pose = t(frame["pose"]).clone().requires_grad_(True)
        out = dr.render_rays_diff(p, fr, o, d, z, pose, lambda q: r.knn(q, 8)[1], frozen_renderer=r)
        loss = (out["feat"] * cf).sum() + out["depth"].sum()
        g1 = torch.autograd.grad(loss, [o, d, pose], retain_graph=True)
        g2 = torch.autograd.grad(loss, [o, d, pose])
        return g1, g2
    keep, graphs = dr.KEEP_BYTES, dr.USE_GRAPHS
    try:
        for mode in ("kept", "graph", "chunking"):
            dr.KEEP_BYTES = 0 if mode == "chunking" else keep
            dr.USE_GRAPHS = mode == "graph"
            if mode == "graph":
                twice()   # first sight of the shape: the next call replays graphs
            g1, g2 = twice()
            for nm, a, b in zip(("g_o", "g_d", "g_pose"), g1, g2):
                assert torch.isfinite(b).all()
                assert rel_err(b.cpu().numpy(), a.cpu().numpy()) < 1e-4, (mode, nm)
    finally:
        dr.KEEP_BYTES, dr.USE_GRAPHS = keep, graphs


@pytest.mark.gpu
def test_huge_feature_maps_give_finite_outputs_and_gradients():
    """the split-FP16 arithmetic (the neural-point kernel of f16mx, the decoders, the backward passes' recomputed forward) has no range handling of
    its own — a value beyond fp16's 65504 used to become inf and the gradient NaN.  The kernels now run with MODE.FP16_OVFL (conversions saturate): support features
    and feature maps scaled to ~1e5 (table rows and activations far outside fp16's range) must give FINITE renders and pose gradients in every mode — accuracy
    is lost there by design."""
    from nerf_loc_amd.renderer import HipRenderer
    from nerf_loc_amd.synth import SceneConfig, make_frame, make_rays, make_weights
    cfg = SceneConfig("huge", R=16, S=32, W=128, V=4, H=48, Wimg=64, seed=9)
    frame, weights = make_frame(cfg), make_weights(cfg)
    rays = make_rays(cfg, frame)
    frame["feat_fine_src"] = frame["feat_fine_src"] * 1e5
    frame["support_fine"]["feature"] = frame["support_fine"]["feature"] * 1e5
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    lin = torch.linspace(0, 1, cfg.S_total, device=dev)
    z = (cfg.near * (1 - lin) + cfg.far * lin).expand(cfg.R, cfg.S_total).contiguous()
    outs = {}
    for prec in ("fp32", "bf16x3", "f16mx"):
        r = HipRenderer(cfg.W, cfg.C, cfg.S_total, prec)
        r.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
        r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
        fr = {k: t(frame[k]) for k in ("topk_Ks", "topk_poses", "topk_images", "feat_fine_src", "vis_featmaps")}
        fr.update({"near": float(cfg.near), "far": float(cfg.far), "support": {k: t(v) for k, v in frame["support_fine"].items()}})
        p = {k: t(v) for k, v in weights.items()}
        o, d, pose = t(rays["rays_o"]).requires_grad_(True), t(rays["rays_d"]).requires_grad_(True), t(frame["pose"]).clone().requires_grad_(True)
        out = dr.render_rays_diff(p, fr, o, d, z, pose, lambda q: r.knn(q, 8)[1], frozen_renderer=r)
        for k in ("rgb", "depth", "weights", "feat"):
            assert torch.isfinite(out[k]).all(), (prec, k)
        g = torch.autograd.grad(out["rgb"].sum() + out["depth"].sum(), [o, d, pose])
        assert all(torch.isfinite(x).all() for x in g), prec
        outs[prec] = {k: out[k].detach().cpu().numpy() for k in ("rgb", "depth", "weights")}
    # (no accuracy claim at this magnitude — the colour blend's softmax sees logits of order 1e5, LayerNorms divide differences of 1e5-sized numbers: only
    #  finiteness is asserted; the parity of the modes at ordinary magnitudes is what every other test holds)
