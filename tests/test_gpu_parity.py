"""GPU parity tests (run with `pytest -m gpu` on the MI355X box): everything goes through the C-ABI.

Tolerances (max |a-b| / max |b|, SURVEY.md App. B metric): fp32 MFMA mode 5e-5; bf16x3 split mode 1e-4
(BASELINE.json's bar); integer/index work (KNN distances, indices, ray mask) bit-exact."""
import os

import numpy as np
import pytest
import torch

from tests.golden_cases import CASES, build_case
from tests.util import l2_rel, load_golden, oracle_inputs, rel_err

pytestmark = pytest.mark.gpu

TOL = {"fp32": 5e-5, "bf16x3": 1e-4, "f16mx": 1e-4}
PLAIN = [n for n in CASES if CASES[n][0].N_importance == 0]


def _renderer(case, precision, **kw):
    from nerf_loc_amd.renderer import HipRenderer
    cfg, fr = case["cfg"], case["frame"]
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, precision, **kw)
    r.load_weights({k: torch.from_numpy(v) for k, v in case["weights"].items()})
    r.set_frame(fr["topk_images"], fr["feat_fine_src"], fr["vis_featmaps"], fr["topk_Ks"], fr["topk_poses"], cfg.near, cfg.far, fr["support_fine"])
    return r


def _z(cfg, R):
    from oracle.render_oracle import sample_depths
    return sample_depths(cfg.S, torch.tensor(cfg.near), torch.tensor(cfg.far), cfg.lindisp).expand(R, cfg.S).contiguous()


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16mx"])
@pytest.mark.parametrize("name", PLAIN)
def test_render_rays_matches_reference_golden(name, precision):
    cfg, full = CASES[name]
    case = build_case(name)
    g = load_golden(name)
    r = _renderer(case, precision)
    out = r.render_rays(case["rays"]["rays_o"], case["rays"]["rays_d"], case["frame"]["pose"][:3, 3], z_vals=_z(cfg, cfg.R),
                        white_bkgd=cfg.white_bkgd, intermediates=True)
    o = {k: v.cpu().numpy() for k, v in out.items()}
    assert np.array_equal(o["knn_d2"], g["knn_d2"]), "squared distances are bit-exact"
    assert np.array_equal(o["mask"], g["mask"])
    if name != "ties":
        assert np.array_equal(o["knn_idx"], g["knn_idx"])
    tol = TOL[precision]
    for k in ("rgb", "depth", "weights", "depth_uncertainty", "feat", "sigma"):
        assert rel_err(o[k], g[k]) < tol, (k, rel_err(o[k], g[k]))
        # max-rel-to-max says nothing about the small entries (most `weights`, `depth_uncertainty`): an L2-relative bound as well
        assert l2_rel(o[k], g[k]) < tol, (k, "l2", l2_rel(o[k], g[k]))
    rows = g["rows"] if "rows" in g else slice(None)
    for k, gk in (("mv_feature_agg", "multiview_feature_agg"), ("feature_agg", "feature_agg"), ("geo", "geo")):
        assert rel_err(o[k][rows], g[gk]) < tol, (k, rel_err(o[k][rows], g[gk]))
        assert l2_rel(o[k][rows], g[gk]) < tol, (k, "l2", l2_rel(o[k][rows], g[gk]))


@pytest.mark.parametrize("name", ["c1", "w256s128"])
def test_bf16_throughput_mode_error_is_reported_not_hidden(name):
    """Plain bf16 does NOT meet 1e-4 (SURVEY §7 'hard parts'); keep its error bounded and known."""
    case = build_case(name)
    g = load_golden(name)
    cfg = case["cfg"]
    r = _renderer(case, "bf16")
    out = r.render_rays(case["rays"]["rays_o"], case["rays"]["rays_d"], case["frame"]["pose"][:3, 3], z_vals=_z(cfg, cfg.R))
    errs = {k: rel_err(out[k].cpu().numpy(), g[k]) for k in ("rgb", "depth", "weights", "feat")}
    assert all(e < 3e-2 for e in errs.values()), errs
    assert max(errs.values()) > 1e-4, "if this ever passes 1e-4 the headline mode should switch to bf16"


@pytest.mark.parametrize("n,m,K,kind", [(5000, 3000, 8, "cloud"), (4096, 20000, 8, "sheets"), (2000, 5, 8, "few"), (3000, 700, 1, "cloud"),
                                         (2000, 900, 8, "lattice"), (1500, 1, 8, "one"), (1000, 2000, 8, "far"), (1003, 4000, 8, "cloud"), (37, 500, 1, "cloud"),
                                         (4096, 6000, 8, "rays"), (600, 300, 8, "dupes"), (1500, 4000, 8, "aniso"), (1200, 9, 8, "rays"),
                                         # ~190 exact ties per location: the deferred-selection list (64 entries) overflows into the selection loop
                                         (600, 1300, 8, "dupes"),
                                         # a render-sized batch: the 16-queries-per-wave instance (N >= 2^18)
                                         (262144 + 37, 2500, 8, "rays")])
def test_knn_exact_vs_oracle(n, m, K, kind):
    from nerf_loc_amd.renderer import HipRenderer
    from oracle import render_oracle as orc
    rng = np.random.default_rng(n + m + K)
    p = rng.standard_normal((m, 3)).astype(np.float32)
    q = (rng.standard_normal((n, 3)) * 1.5).astype(np.float32)
    if kind == "sheets":
        p[:, 2] = (0.05 * np.sin(3 * p[:, 0]) + np.round(p[:, 2])).astype(np.float32)
    if kind == "lattice":
        p = (np.round(p * 3) / 3).astype(np.float32)
        q = (np.round(q * 3) / 3).astype(np.float32)
    if kind == "far":
        q[::2] += np.array([40.0, -25.0, 10.0], np.float32)
    if kind == "rays":      # consecutive queries march along rays: the path that seeds a query's bound from its predecessor's neighbours
        o = rng.standard_normal((n // 64 + 1, 3)) * 0.3
        d = rng.standard_normal((n // 64 + 1, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        t = np.linspace(-2.5, 2.5, 64)
        q = (o[:, None, :] + d[:, None, :] * t[None, :, None]).reshape(-1, 3)[:n].astype(np.float32)
    if kind == "dupes":     # every point is one of 7 locations: ties everywhere, resolved by index
        p = p[rng.integers(0, 7, m)]
        q[: n // 2] = p[rng.integers(0, m, n // 2)]
    if kind == "aniso":     # a slab: 1e3 units wide, 1e-3 thick
        p = (p * np.array([300.0, 20.0, 3e-4], np.float32)).astype(np.float32)
        q = (q * np.array([200.0, 15.0, 5e-4], np.float32)).astype(np.float32)
    case = build_case("tiny_full")
    case["frame"]["support_fine"] = {"xyz": p, "feature": np.zeros((m, 195), np.float32), "confidence": np.ones((m, 1), np.float32),
                                     "direction": np.zeros((m, 4), np.float32)}
    r = _renderer(case, "fp32")
    d2, idx = r.knn(q, K)
    d2o, idxo = orc.knn_points(torch.from_numpy(q), torch.from_numpy(p), K)
    assert np.array_equal(d2.cpu().numpy(), d2o.numpy())
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), idxo.numpy()), "ties resolve by index exactly like knn_cpu.cpp's heap"


@pytest.mark.parametrize("name", ["tiny_full", "offview", "w256s128"])
def test_stage_entry_points_match_oracle(name):
    from oracle import render_oracle as orc
    cfg, _ = CASES[name]
    case = build_case(name)
    params, frame, rays = oracle_inputs(case)
    r = _renderer(case, "fp32")
    z = _z(cfg, cfg.R)
    zo, xyz = r.sample_points(rays["rays_o"], rays["rays_d"], z)
    xyz_ref = (rays["rays_o"][:, None] + rays["rays_d"][:, None] * z[..., None]).view(-1, 3)
    assert np.array_equal(xyz.cpu().numpy(), xyz_ref.numpy()), "a3 is bit-exact (no fma contraction)"
    with torch.no_grad():
        mv_o, rgbf_o, vis_o, mask1_o = orc.mv_aggregate(params, frame, xyz_ref)
    mv, rgbf, vis_ang, valid = r.mv_aggregate(xyz, case["frame"]["pose"][:3, 3])
    assert rel_err(mv.cpu().numpy(), mv_o.numpy()) < 5e-5
    assert rel_err(rgbf[:, :, :195].cpu().numpy(), rgbf_o.numpy()) < 1e-5
    assert rel_err(vis_ang[:, :, 0].cpu().numpy(), vis_o.squeeze(-1).numpy()) < 2e-5
    assert np.array_equal(valid.cpu().numpy() > 0, (mask1_o.squeeze(-1).sum(1) > 1).numpy())
    ang_o = orc.compute_angle(xyz_ref, frame["pose"], frame["topk_poses"]).permute(1, 0, 2)
    # unit(difference of two nearly parallel unit vectors) is ill-conditioned: fp32 rounding (~1e-7 on each unit
    # vector) is amplified by 1/|diff|, |diff| = sqrt(2 - 2 dot).  Bound the error accordingly.
    ang_h, ang_r = vis_ang[:, :, 1:5].cpu().numpy(), ang_o.numpy()
    nd = np.sqrt(np.maximum(2.0 - 2.0 * ang_r[..., 3:4].astype(np.float64), 1e-10))
    assert (np.abs(ang_h[..., :3] - ang_r[..., :3]) <= 1e-6 / nd + 2e-6).all()
    assert np.abs(ang_h[..., 3] - ang_r[..., 3]).max() < 1e-6
    d = torch.cat([rays["rays_d"][:, None].repeat(1, cfg.S, 1).view(-1, 3), z.reshape(-1, 1)], -1)
    with torch.no_grad():
        qd = orc.query(params, frame, xyz_ref, d)
    fa, d2, idx = r.point_mlp(xyz, d[:, :3].contiguous(), mv_o)
    assert np.array_equal(d2.cpu().numpy(), qd["knn_d2"].numpy())
    assert rel_err(fa.cpu().numpy(), qd["feature_agg"].numpy()) < 5e-5
    W = cfg.W
    with torch.no_grad():
        geo_o = orc.ray_unet(params, qd["feature_agg"].view(cfg.R, cfg.S, W).permute(0, 2, 1)).permute(0, 2, 1).reshape(-1, W)
    geo = r.ray_unet(qd["feature_agg"])
    assert rel_err(geo.cpu().numpy(), geo_o.numpy()) < 5e-5


def test_descriptor_query_direction_falls_back_to_nearest_neighbour():
    """query() with direction=None (model.py:391-392), as query_coarse/query_fine call it."""
    from oracle import render_oracle as orc
    case = build_case("tiny_full")
    params, frame, rays = oracle_inputs(case)
    r = _renderer(case, "fp32")
    xyz = frame["support_fine"]["xyz"][::7][:200].clone() + 0.01
    with torch.no_grad():
        qd = orc.query(params, frame, xyz, None)
    fa, _, _ = r.point_mlp(xyz, None, qd["multiview_feature_agg"])
    assert rel_err(fa.cpu().numpy(), qd["feature_agg"].numpy()) < 5e-5


@pytest.mark.parametrize("name", ["w128s64", "w256s128"])
def test_descriptor_query_direction_fallback_on_the_fused_kernels(name):
    """The same fallback through the fused neural-point kernels (W = 128 / 256, bf16x3): without a direction array every lane takes the
    direction of its sample's nearest neighbour (first of the 8 lanes of the sample), and k >= M rows read as zero."""
    from oracle import render_oracle as orc
    case = build_case(name)
    params, frame, rays = oracle_inputs(case)
    r = _renderer(case, "bf16x3")
    xyz = frame["support_fine"]["xyz"][::5][:333].clone() + 0.007     # 333: not a multiple of the 16-sample tile
    with torch.no_grad():
        qd = orc.query(params, frame, xyz, None)
    fa, d2, idx = r.point_mlp(xyz, None, qd["multiview_feature_agg"])
    assert rel_err(fa.cpu().numpy(), qd["feature_agg"].numpy()) < 1e-4


HIER = [n for n in CASES if CASES[n][0].N_importance > 0]


@pytest.mark.parametrize("name", HIER)
def test_hierarchical_branch_matches_reference_golden(name):
    """a20: coarse NeuRay weights -> sample_pdf (recipe uniforms) -> merge/sort -> render, vs the reference golden, at BASELINE's
    1e-4.  Two stages own the error and are bounded separately: (1) the depths z — the HIP coarse pass against the oracle's,
    relative to the depth span; (2) the renderer alone, fed the ORACLE's depths.  The end-to-end path (HIP depths into the HIP
    renderer) is then held to 1e-4 as well, and if it ever exceeds it the message says which stage moved."""
    from oracle import render_oracle as orc
    cfg, _ = CASES[name]
    case = build_case(name)
    g = load_golden(name)
    params, frame, rays = oracle_inputs(case)
    u = torch.from_numpy(case["u"])
    with torch.no_grad():
        zb = orc.sample_depths(cfg.S, torch.tensor(cfg.near), torch.tensor(cfg.far), cfg.lindisp).expand(cfg.R, cfg.S).contiguous()
        zc = orc.sample_depths(64, torch.tensor(cfg.near), torch.tensor(cfg.far), cfg.lindisp).expand(cfg.R, 64).contiguous()
        wc_o = orc.predict_weights_from_neuray(params, frame, rays, zc)
        zf = orc.sample_pdf(0.5 * (zc[:, :-1] + zc[:, 1:]), wc_o[:, 1:-1], cfg.N_importance, u)
        z_o = torch.sort(torch.cat([zb, zf], -1), -1)[0]
    for precision in ("fp32", "bf16x3", "f16mx"):
        r = _renderer(case, precision)
        z, depth_coarse, wc = r.hierarchical_depths(case["rays"]["pixel_coordinates"], case["frame"]["K"], case["frame"]["pose"], zb, case["u"], lindisp=cfg.lindisp)
        assert rel_err(wc.cpu().numpy(), wc_o.numpy()) < 2e-5
        assert rel_err(depth_coarse.cpu().numpy(), g["depth_coarse"]) < 2e-5
        z_err = float((z.cpu() - z_o).abs().max()) / (cfg.far - cfg.near)
        assert z_err < 2e-5, ("stage 1 (depths)", precision, z_err)
        tol = TOL[precision]
        alone = r.render_rays(case["rays"]["rays_o"], case["rays"]["rays_d"], case["frame"]["pose"][:3, 3], z_vals=z_o, intermediates=True)
        e_alone = {k: rel_err(alone[k].cpu().numpy(), g[k]) for k in ("rgb", "depth", "weights", "depth_uncertainty", "feat", "sigma")}
        assert max(e_alone.values()) < tol, ("stage 2 (renderer on the oracle's depths)", precision, e_alone)
        out = r.render_rays(case["rays"]["rays_o"], case["rays"]["rays_d"], case["frame"]["pose"][:3, 3], z_vals=z, intermediates=True)
        e_e2e = {k: rel_err(out[k].cpu().numpy(), g[k]) for k in e_alone}
        assert max(e_e2e.values()) < 1e-4, ("end to end (BASELINE's bar)", precision, e_e2e, "renderer alone", e_alone, "depths", z_err)
        assert np.array_equal(out["mask"].cpu().numpy(), g["mask"])


@pytest.mark.parametrize("seed", range(8))
def test_random_hierarchical_configs_match_oracle(seed):
    """a20 at seeded random sizes (base / importance sample counts, 1-16 views, image size): coarse weights -> sample_pdf with given
    uniforms -> merge / sort -> render, against the oracle's hierarchical render."""
    from oracle import render_oracle as orc
    from nerf_loc_amd.synth import make_frame, make_rays, make_weights
    rng = np.random.default_rng(7000 + seed)
    cfg = CASES["tiny_full"][0].replace(name=f"hrand{seed}", seed=700 + seed, W=int(rng.choice([32, 64, 128])), S=int(8 * rng.integers(1, 5)),
                                        N_importance=int(8 * rng.integers(1, 5)), V=int(rng.integers(1, 17)), R=int(rng.integers(1, 25)),
                                        H=int(rng.integers(24, 81)), Wimg=int(rng.integers(24, 81)))
    frame = make_frame(cfg)
    case = {"cfg": cfg, "frame": frame, "rays": make_rays(cfg, frame), "weights": make_weights(cfg)}
    u = rng.random((cfg.R, cfg.N_importance), dtype=np.float32)
    params = {k: torch.from_numpy(v) for k, v in case["weights"].items()}
    rays_t = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in case["rays"].items()}
    with torch.no_grad():
        ref = orc.render_rays(params, orc.to_torch(frame), rays_t, cfg.S, cfg.N_importance, u=torch.from_numpy(u))
    for precision in ("fp32", "bf16x3"):
        r = _renderer(case, precision)
        zb = orc.sample_depths(cfg.S, torch.tensor(cfg.near), torch.tensor(cfg.far)).expand(cfg.R, cfg.S).contiguous()
        z, depth_coarse, _ = r.hierarchical_depths(case["rays"]["pixel_coordinates"], frame["K"], frame["pose"], zb, u)
        assert rel_err(depth_coarse.cpu().numpy(), ref["depth_coarse"].numpy()) < 2e-5, cfg
        out = r.render_rays(case["rays"]["rays_o"], case["rays"]["rays_d"], frame["pose"][:3, 3], z_vals=z)
        assert np.array_equal(out["mask"].cpu().numpy(), ref["mask"].numpy()), cfg
        for k in ("rgb", "depth", "weights", "depth_uncertainty", "feat"):
            assert rel_err(out[k].cpu().numpy(), ref[k].numpy()) < 1e-4, (cfg, precision, k, rel_err(out[k].cpu().numpy(), ref[k].numpy()))


# ------------------------------------------------------------------ full-size (BASELINE config 2) properties
@pytest.fixture(scope="module")
def c2():
    from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights
    cfg = CONFIGS["c2"]
    frame = make_frame(cfg)
    return {"cfg": cfg, "frame": frame, "rays": make_rays(cfg, frame), "weights": make_weights(cfg)}


@pytest.mark.parametrize("prec", ["f16mx", "bf16x3"])
def test_full_size_properties_c2(c2, prec):
    cfg = c2["cfg"]
    r = _renderer(c2, prec)
    o, d = c2["rays"]["rays_o"], c2["rays"]["rays_d"]
    qc = c2["frame"]["pose"][:3, 3]
    z = _z(cfg, cfg.R)
    a = r.render_rays(o, d, qc, z_vals=z)
    torch.cuda.synchronize()
    for k in ("rgb", "depth", "weights", "depth_uncertainty", "feat"):
        assert torch.isfinite(a[k]).all(), k
    # last delta = 1e2 makes every ray opaque: weights sum to 1 (model.py:544-553)
    assert float((a["weights"].sum(1) - 1).abs().max()) < 1e-4
    assert float(a["depth"].min()) >= cfg.near - 1e-4 and float(a["depth"].max()) <= cfg.far + 1e-4
    # determinism
    b = r.render_rays(o, d, qc, z_vals=z)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # chunk invariance: a much smaller workspace (more, smaller ray chunks) gives bit-identical results
    from nerf_loc_amd import _lib
    import ctypes as ct
    small = _lib.load().nl_render_rays_workspace_bytes(ct.byref(r.cfg), r.V, 96)
    r2 = _renderer(c2, prec, workspace_bytes=small)
    c = r2.render_rays(o, d, qc, z_vals=z)
    for k in a:
        assert torch.equal(a[k], c[k]), k
    # ray-permutation equivariance
    perm = torch.randperm(cfg.R, generator=torch.Generator().manual_seed(0))
    e = r.render_rays(o[perm.numpy()], d[perm.numpy()], qc, z_vals=z)
    for k in a:
        assert torch.equal(a[k][perm.to(a[k].device)], e[k]), k


def test_full_size_sampled_rays_match_oracle_c2(c2):
    """64 of the 4096 rays of the headline workload against the CPU oracle (both parity modes, 1e-4)."""
    from oracle import render_oracle as orc
    cfg = c2["cfg"]
    sel = np.arange(0, cfg.R, cfg.R // 64)[:64]
    r = _renderer(c2, "bf16x3")
    z = _z(cfg, len(sel))
    outs = {}
    for prec in ("bf16x3", "f16mx"):
        r.set_precision(prec)
        outs[prec] = r.render_rays(c2["rays"]["rays_o"][sel], c2["rays"]["rays_d"][sel], c2["frame"]["pose"][:3, 3], z_vals=z)
    out = outs["bf16x3"]
    params = {k: torch.from_numpy(v) for k, v in c2["weights"].items()}
    sub = {k: (torch.from_numpy(v[sel]) if k in ("rays_o", "rays_d", "pixel_coordinates") else (torch.from_numpy(v) if isinstance(v, np.ndarray) else v))
           for k, v in c2["rays"].items()}
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = orc.render_rays(params, orc.to_torch(c2["frame"]), sub, cfg.S, knn_threads=16)
    for prec, out in outs.items():
        assert np.array_equal(out["mask"].cpu().numpy(), ref["mask"].numpy())
        for k in ("rgb", "depth", "weights", "depth_uncertainty", "feat"):
            assert rel_err(out[k].cpu().numpy(), ref[k].numpy()) < 1e-4, (prec, k, rel_err(out[k].cpu().numpy(), ref[k].numpy()))


def test_every_ray_of_the_timed_c2_batch_matches_oracle_f16mx(c2):
    """ALL 4096 rays of the headline workload (BASELINE config 2, the batch bench.py times) in the headline mode against the CPU oracle (VERDICT r5 item 2:
    the sampled-rays test above sees 64 of them, rendered as their own batch).  Bar: 1e-4 max-rel-to-max AND L2-relative over the rays that have no sample
    within 1e-3 pixel of a support view's image border; a ray that has one is a hard-threshold case of the reference's in-image test (synth.borderline_rays:
    two correct fp32 evaluations may differ) — those are compared too, and at most a handful of them may actually flip."""
    from oracle import render_oracle as orc
    from nerf_loc_amd.synth import borderline_rays
    cfg = c2["cfg"]
    o, d = c2["rays"]["rays_o"], c2["rays"]["rays_d"]
    r = _renderer(c2, "f16mx")
    z = _z(cfg, cfg.R)
    out = r.render_rays(o, d, c2["frame"]["pose"][:3, 3], z_vals=z)
    out = {k: v.cpu().numpy() for k, v in out.items()}
    params = {k: torch.from_numpy(v) for k, v in c2["weights"].items()}
    fr = orc.to_torch(c2["frame"])
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    parts = []
    for lo in range(0, cfg.R, 256):   # (a 256-ray c2 chunk of the oracle peaks around 15 GB, like the reference's `render.chunk` loop)
        sub = {k: (torch.from_numpy(v[lo:lo + 256]) if k in ("rays_o", "rays_d", "pixel_coordinates") else (torch.from_numpy(v) if isinstance(v, np.ndarray) else v))
               for k, v in c2["rays"].items()}
        with torch.no_grad():
            parts.append({k: v.numpy() for k, v in orc.render_rays(params, fr, sub, cfg.S, knn_threads=threads).items() if k != "z_vals"})
    ref = {k: np.concatenate([p_[k] for p_ in parts], 0) for k in parts[0]}
    border = borderline_rays(cfg, c2["frame"], o, d, z.numpy())
    core = ~border
    assert core.sum() > 0.9 * cfg.R
    keys = ("rgb", "depth", "weights", "depth_uncertainty", "feat")
    errs = {k: rel_err(out[k][core], ref[k][core]) for k in keys}
    l2s = {k: l2_rel(out[k][core], ref[k][core]) for k in keys}
    print("c2 all rays f16mx: max-rel", errs, "l2", l2s, "borderline rays", int(border.sum()))
    assert np.array_equal(out["mask"][core].astype(bool), ref["mask"][core].astype(bool))
    assert max(errs.values()) < 1e-4, errs
    assert max(l2s.values()) < 1e-4, l2s
    # the borderline rays: same scale (the batch's max |oracle|); a flip moves a ray by percents, everything else stays at the core's level
    flipped = np.zeros(cfg.R, bool)
    for k in keys:
        den = max(np.abs(ref[k].astype(np.float64)).max(), 1e-30)
        e = np.abs(out[k].astype(np.float64) - ref[k].astype(np.float64)).reshape(cfg.R, -1).max(1) / den
        flipped |= border & (e >= 1e-4)
    print("borderline rays that differ by >= 1e-4:", int(flipped.sum()), "of", int(border.sum()))
    assert flipped.sum() <= max(4, border.sum() // 10), (int(flipped.sum()), int(border.sum()))


# ------------------------------------------------------------------ empty and ragged ray batches
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_empty_and_ragged_ray_batches(precision):
    """R = 0 returns empty outputs; R = 1, 3 and a size that is not a multiple of any tile (37) reproduce the same rays
    rendered inside a larger batch (the renderer treats rays independently: rows of a 32/128-row tile that do not exist
    must not leak into the ones that do)."""
    name = "c1"
    cfg, _ = CASES[name]
    case = build_case(name)
    r = _renderer(case, precision)
    o, d = case["rays"]["rays_o"], case["rays"]["rays_d"]
    qc = case["frame"]["pose"][:3, 3]
    full = r.render_rays(o, d, qc, z_vals=_z(cfg, cfg.R))
    empty = r.render_rays(o[:0], d[:0], qc, z_vals=_z(cfg, 0))
    for k, v in empty.items():
        assert v.shape[0] == 0, k
    for n in (1, 3, 37):
        n = min(n, cfg.R)
        part = r.render_rays(o[:n], d[:n], qc, z_vals=_z(cfg, n))
        for k in ("rgb", "depth", "weights", "depth_uncertainty", "feat"):
            assert rel_err(part[k].cpu().numpy(), full[k][:n].cpu().numpy()) < (2e-6 if precision == "fp32" else 2e-5), (n, k)
        assert torch.equal(part["mask"], full["mask"][:n])


# ------------------------------------------------------------------ view-count specialisations of the multi-view gather
@pytest.mark.parametrize("V", [8, 12, 16])
def test_view_count_variants_match_oracle(V):
    """mv_stats is specialised per view-count bucket (4 / 8 / 10 / 16, exact or guarded); the golden cases cover V = 3, 4, 5 and
    10 — this checks the remaining kernels (exact 8, guarded 16 via V = 12, exact 16) against the oracle on a tiny scene."""
    from oracle import render_oracle as orc
    from nerf_loc_amd.synth import make_frame, make_rays, make_weights
    cfg = CASES["tiny_full"][0].replace(name=f"tiny_v{V}", V=V, seed=100 + V)
    frame = make_frame(cfg)
    case = {"cfg": cfg, "frame": frame, "rays": make_rays(cfg, frame), "weights": make_weights(cfg)}
    params = {k: torch.from_numpy(v) for k, v in case["weights"].items()}
    rays_t = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in case["rays"].items()}
    with torch.no_grad():
        ref = orc.render_rays(params, orc.to_torch(frame), rays_t, cfg.S)
    for precision in ("fp32", "bf16x3"):
        r = _renderer(case, precision)
        out = r.render_rays(case["rays"]["rays_o"], case["rays"]["rays_d"], frame["pose"][:3, 3], z_vals=_z(cfg, cfg.R))
        assert np.array_equal(out["mask"].cpu().numpy(), ref["mask"].numpy())
        for k in ("rgb", "depth", "weights", "depth_uncertainty", "feat"):
            assert rel_err(out[k].cpu().numpy(), ref[k].numpy()) < TOL[precision], (V, precision, k, rel_err(out[k].cpu().numpy(), ref[k].numpy()))


@pytest.mark.parametrize("C,W", [(64, 32), (128, 64), (61, 32), (100, 128), (61, 256)])
def test_feature_width_variants_match_oracle(C, W):
    """backbone2d_fpn_dim is 192 in every shipped reference config, but it is a config field: narrower maps, and a width that is
    not a multiple of 4 (scalar-load path of the multi-view gather, unaligned GEMM segments), with the staged (W = 32) and the fused
    (W = 64 / 128 / 256) neural-point kernels, against the oracle on a tiny scene."""
    from oracle import render_oracle as orc
    from nerf_loc_amd.synth import make_frame, make_rays, make_weights
    cfg = CASES["tiny_full"][0].replace(name=f"tiny_c{C}", C=C, W=W, seed=200 + C)
    frame = make_frame(cfg)
    case = {"cfg": cfg, "frame": frame, "rays": make_rays(cfg, frame), "weights": make_weights(cfg)}
    params = {k: torch.from_numpy(v) for k, v in case["weights"].items()}
    rays_t = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in case["rays"].items()}
    with torch.no_grad():
        ref = orc.render_rays(params, orc.to_torch(frame), rays_t, cfg.S)
    for precision in ("fp32", "bf16x3"):
        r = _renderer(case, precision)
        out = r.render_rays(case["rays"]["rays_o"], case["rays"]["rays_d"], frame["pose"][:3, 3], z_vals=_z(cfg, cfg.R))
        assert np.array_equal(out["mask"].cpu().numpy(), ref["mask"].numpy())
        for k in ("rgb", "depth", "weights", "depth_uncertainty", "feat"):
            assert rel_err(out[k].cpu().numpy(), ref[k].numpy()) < TOL[precision], (C, W, precision, k, rel_err(out[k].cpu().numpy(), ref[k].numpy()))


@pytest.mark.parametrize("W,S,R", [(96, 24, 16), (160, 40, 9), (224, 8, 33), (32, 256, 5), (192, 72, 7)])
def test_hidden_width_and_sample_count_variants_match_oracle(W, S, R):
    """nl_config accepts every hidden width that is a multiple of 32 up to 256 and every sample count that is a multiple of 8 up to
    256 (three stride-2 U-Net levels); the goldens cover W = 32 / 64 / 128 / 256 and S = 16 / 32 / 64 / 128 only."""
    from oracle import render_oracle as orc
    from nerf_loc_amd.synth import make_frame, make_rays, make_weights
    cfg = CASES["tiny_full"][0].replace(name=f"tiny_w{W}s{S}", W=W, S=S, R=R, seed=300 + W + S)
    frame = make_frame(cfg)
    case = {"cfg": cfg, "frame": frame, "rays": make_rays(cfg, frame), "weights": make_weights(cfg)}
    params = {k: torch.from_numpy(v) for k, v in case["weights"].items()}
    rays_t = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in case["rays"].items()}
    with torch.no_grad():
        ref = orc.render_rays(params, orc.to_torch(frame), rays_t, cfg.S)
    for precision in ("fp32", "bf16x3"):
        r = _renderer(case, precision)
        out = r.render_rays(case["rays"]["rays_o"], case["rays"]["rays_d"], frame["pose"][:3, 3], z_vals=_z(cfg, cfg.R))
        assert np.array_equal(out["mask"].cpu().numpy(), ref["mask"].numpy())
        for k in ("rgb", "depth", "weights", "depth_uncertainty", "feat"):
            assert rel_err(out[k].cpu().numpy(), ref[k].numpy()) < TOL[precision], (W, S, precision, k, rel_err(out[k].cpu().numpy(), ref[k].numpy()))


@pytest.mark.parametrize("H,Wimg,V", [(50, 70, 3), (31, 45, 2), (120, 160, 6)])
def test_image_size_variants_match_oracle(H, Wimg, V):
    """Image sizes that are not multiples of 4 (feature maps at floor(H/4) x floor(W/4)): projection, in-bounds tests and the three
    bilinear conventions with odd extents."""
    from oracle import render_oracle as orc
    from nerf_loc_amd.synth import make_frame, make_rays, make_weights
    cfg = CASES["tiny_full"][0].replace(name=f"tiny_{H}x{Wimg}", H=H, Wimg=Wimg, V=V, seed=400 + H)
    frame = make_frame(cfg)
    case = {"cfg": cfg, "frame": frame, "rays": make_rays(cfg, frame), "weights": make_weights(cfg)}
    params = {k: torch.from_numpy(v) for k, v in case["weights"].items()}
    rays_t = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in case["rays"].items()}
    with torch.no_grad():
        ref = orc.render_rays(params, orc.to_torch(frame), rays_t, cfg.S)
    for precision in ("fp32", "bf16x3"):
        r = _renderer(case, precision)
        out = r.render_rays(case["rays"]["rays_o"], case["rays"]["rays_d"], frame["pose"][:3, 3], z_vals=_z(cfg, cfg.R))
        assert np.array_equal(out["mask"].cpu().numpy(), ref["mask"].numpy())
        for k in ("rgb", "depth", "weights", "depth_uncertainty", "feat"):
            assert rel_err(out[k].cpu().numpy(), ref[k].numpy()) < TOL[precision], (H, Wimg, precision, k, rel_err(out[k].cpu().numpy(), ref[k].numpy()))


@pytest.mark.parametrize("N", [1, 7, 37, 1003])
def test_multiview_gather_kernels_agree_on_ragged_point_sets(N):
    """The render path's gather kernel (eight samples per wave) against the stage-API kernel (one sample per wave, pinned on the
    goldens) on point sets whose size is not a multiple of 8 and whose points are unrelated to each other."""
    name = "w128s64"
    case = build_case(name)
    r = _renderer(case, "fp32")
    rng = np.random.default_rng(N)
    sp = case["frame"]["support_fine"]["xyz"]
    pts = (sp[rng.integers(0, len(sp), N)] + 0.05 * rng.standard_normal((N, 3))).astype(np.float32)
    qc = case["frame"]["pose"][:3, 3]
    mv_a, _, _, valid_a = r.mv_aggregate(pts, qc, want_raw=True)
    mv_b, _, _, valid_b = r.mv_aggregate(pts, qc, want_raw=False)
    assert torch.equal(valid_a, valid_b)
    assert rel_err(mv_b.cpu().numpy(), mv_a.cpu().numpy()) < 1e-5


def _random_cfg(seed):
    rng = np.random.default_rng(1000 + seed)
    return CASES["tiny_full"][0].replace(
        name=f"rand{seed}", seed=500 + seed, W=int(rng.choice([32, 64, 96, 128, 160, 192, 224, 256])), S=int(8 * rng.integers(1, 9)),
        V=int(rng.integers(1, 17)), C=int(rng.choice([8, 20, 36, 64, 100, 132, 192, 33, 77])), R=int(rng.integers(1, 41)),
        H=int(rng.integers(24, 101)), Wimg=int(rng.integers(24, 101)), white_bkgd=bool(rng.integers(0, 2)))


@pytest.mark.parametrize("seed", range(24))
def test_random_configs_match_oracle(seed):
    """Seeded random (W, S, V, C, R, H, Wimg, white_bkgd) combinations — one view, 16 views, tiny feature maps, odd channel counts,
    single rays — against the oracle; which kernels run (fused / staged point MLP, 8-samples-per-wave / per-sample gather, fused /
    separate LayerNorm) depends on the draw."""
    from oracle import render_oracle as orc
    from nerf_loc_amd.synth import make_frame, make_rays, make_weights
    cfg = _random_cfg(seed)
    frame = make_frame(cfg)
    case = {"cfg": cfg, "frame": frame, "rays": make_rays(cfg, frame), "weights": make_weights(cfg)}
    params = {k: torch.from_numpy(v) for k, v in case["weights"].items()}
    rays_t = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in case["rays"].items()}
    with torch.no_grad():
        ref = orc.render_rays(params, orc.to_torch(frame), rays_t, cfg.S, white_bkgd=cfg.white_bkgd)
    for precision in ("fp32", "bf16x3"):
        r = _renderer(case, precision)
        out = r.render_rays(case["rays"]["rays_o"], case["rays"]["rays_d"], frame["pose"][:3, 3], z_vals=_z(cfg, cfg.R), white_bkgd=cfg.white_bkgd)
        assert np.array_equal(out["mask"].cpu().numpy(), ref["mask"].numpy()), cfg
        for k in ("rgb", "depth", "weights", "depth_uncertainty", "feat"):
            assert rel_err(out[k].cpu().numpy(), ref[k].numpy()) < 1e-4, (cfg, precision, k, rel_err(out[k].cpu().numpy(), ref[k].numpy()))


def test_repacking_weights_in_place_refreshes_per_frame_tables():
    """The per-frame tables (T = sp_feature . W1, blend-projected maps) are derived from the weights: loading other weights into the
    SAME packed buffer while a frame is set must rebuild them."""
    name = "c1"
    cfg, _ = CASES[name]
    case = build_case(name)
    r = _renderer(case, "bf16x3")
    o, d, qc = case["rays"]["rays_o"], case["rays"]["rays_d"], case["frame"]["pose"][:3, 3]
    a = r.render_rays(o, d, qc, z_vals=_z(cfg, cfg.R))
    w2 = {k: (v * 1.25 if k in ("base_mlp.0.weight", "rgb_blending_mlp.0.weight") else v) for k, v in case["weights"].items()}
    r.load_weights({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w2.items()})    # same packed buffer, frame untouched
    b = r.render_rays(o, d, qc, z_vals=_z(cfg, cfg.R))
    case2 = dict(case); case2["weights"] = w2
    fresh = _renderer(case2, "bf16x3").render_rays(o, d, qc, z_vals=_z(cfg, cfg.R))
    assert rel_err(a["rgb"].cpu().numpy(), b["rgb"].cpu().numpy()) > 1e-4, "the weight change must be visible"
    for k in ("rgb", "depth", "feat"):
        assert torch.equal(b[k], fresh[k]), k


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_repeated_renders_are_bit_identical(precision):
    """The same batch rendered repeatedly gives the same bits in every mode.  (The chain kernels count outstanding memory operations
    instead of draining them: a row store the compiler scheduled in front of the weight pieces it was counted behind once made the
    blend projection read a chunk that had not landed — only in bf16 mode, where such a chunk is one piece per wave — and colours
    differed from run to run.  tools/race_check.py is the long version of this test.)"""
    from nerf_loc_amd.renderer import HipRenderer
    case = build_case("w256s128")
    cfg, frame, rays = case["cfg"], case["frame"], case["rays"]
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, precision)
    r.load_weights({k: torch.from_numpy(v) for k, v in case["weights"].items()})
    r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
    first = None
    for _ in range(12):
        out = r.render_rays(rays["rays_o"], rays["rays_d"], frame["pose"][:3, 3])
        torch.cuda.synchronize()
        cur = {k: out[k].clone() for k in ("rgb", "depth", "weights", "feat")}
        if first is None:
            first = cur
        else:
            for k in cur:
                assert torch.equal(cur[k], first[k]), (precision, k)


def test_random_scenes_forward_matches_the_cpu_oracle():
    """tools/forward_fuzz.py on a few random scenes: hidden widths 32 ... 256 in steps of 32, 8 ... 64 samples with and without the hierarchical branch, 1 ... 16
    views, feature widths 5 ... 192 (odd ones too), support sets smaller than K, white background, single rays — fp32 and the parity mode against the CPU
    oracle at 1e-4 (hierarchical: the renderer on the oracle's resampled depths at 1e-4, the resampled depths themselves at 2e-4 of the range)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("forward_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "forward_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # round 4: 100 seeded scenes in the driver-run suite (was 12; the builder-side runs of tools/forward_fuzz.py cover 1 400)
    assert mod.run(int(os.environ.get("NERFLOC_FUZZ_FORWARD", "100")), 9, verbose=False) < 1e-4


# ------------------------------------------------------------------ inputs that are NOT the O(1) synthetic recipe (round 5)
SWEEP = {
    "w256s128": [("normal", 1.0 / 64, 1.0), ("normal", 8.0, 1.0), ("normal", 64.0, 1.0), ("student_t3", 1.0 / 64, 1.0), ("student_t3", 8.0, 8.0),
                 ("student_t3", 64.0, 1.0), ("normal+off4", 1.0, 1.0), ("normal+off32", 0.25, 1.0)],   # +offX: views agree on a large common value (mean^2 >> variance)
    "c2": [("student_t3", 1.0 / 64, 1.0), ("student_t3", 8.0, 1.0), ("normal", 64.0, 1.0)],
}


@pytest.mark.parametrize("case", sorted(SWEEP))
def test_parity_modes_hold_on_scaled_features_and_heavy_tailed_weights(case):
    """tools/scale_sweep.py: feature maps + support features x {1/64, 8, 64}, Student-t (nu = 3) weights of the same fan-in variance, DepthFusionNet maps x 8, feature maps
    with a large common offset — the golden-case scene w256s128 and 64 rays of BASELINE config 2, every precision mode against the CPU oracle in max-rel AND L2-rel.
    Each mode is held to BASELINE's 1e-4 (3 x the fp32 oracle's own distance to the fp64 result where that is larger: tools/forward_fuzz.py's bar) on every scene
    INSIDE its validated conditioning range — max |attention logit| <= 100 for f16mx, <= 500 for bf16x3 (ConditionalNeRF.LOGIT_LIMIT), any for fp32 — and on the
    scenes beyond it the mode the module's precision guard selects must hold that bar.  f16mx's fp8 images carry per-row block scales since round 5: no range
    assumption is left in it besides fp16's own (|activation| < 65504), which this sweep exercises up to ~2e3."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("scale_sweep", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "scale_sweep.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = mod.sweep(case, combos=SWEEP[case], verbose=True)
    assert len(rows) == 3 * len(SWEEP[case])
    print(f"{case}: {sum(not mod.in_range(r) for r in rows)} of {len(rows)} rows are outside their mode's validated range (the guard escalates there); "
          f"{sum(mod.relaxed(r) for r in rows)} are held to 3 x (oracle vs fp64) instead of 1e-4")
    assert mod.check(rows) == []
    assert any(not mod.in_range(r) for r in rows) and any(mod.in_range(r) and r["precision"] == "f16mx" for r in rows), "the sweep must cross the validated range"
