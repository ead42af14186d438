"""Scale-sweep parity (VERDICT r4 weak 1 / ADVICE r4): do the parity modes hold 1e-4 when the inputs are NOT the O(1) synthetic recipe?

Every golden / fuzz case draws N(0, 2/fan_in) weights and N(0, 1) feature maps; a trained checkpoint does not (the reference ships none, README.md:74).
This tool renders the same rays with
  * the feature maps AND the support points' feature columns multiplied by `fscale` (1/64 ... 64): activations of the neural-point MLP, the statistics
    rows of the aggregator and the blend layer move with it;
  * weights from a heavy-tailed recipe: Student-t (nu = 3) scaled to the SAME fan-in variance (single entries at 10-100 sigma);
  * optionally the DepthFusionNet maps (`vis_featmaps`) multiplied by `vscale`
in every precision mode against (a) the CPU oracle (fp32, the reference's op formulation) and (b) the same function in fp64 (the eager restatement on
the GPU) — the second says how well-conditioned the scene is — and records the conditioning indicator the library reports (max |attention logit|).

    python tools/scale_sweep.py [case ...]        cases: w256s128 (golden-case scene), c2 (64 sampled rays of BASELINE config 2), w128s64
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loc_amd.renderer import HipRenderer
from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights, weight_shapes

FSCALES = (1.0 / 64, 1.0, 8.0, 64.0)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def l2_rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def heavy_tailed_weights(cfg, seed=None):
    """make_weights' recipe with the Gaussian draws of every Linear / Conv weight replaced by Student-t (nu = 3) draws of the same variance
    (t_3 has variance 3: divide by sqrt 3); biases and LayerNorm tables as in make_weights."""
    base = make_weights(cfg, seed)
    rng = np.random.default_rng((cfg.seed if seed is None else seed) + 104723)
    out = {}
    shapes = weight_shapes(cfg)
    for name in sorted(shapes):
        shp = shapes[name]
        is_ln = (".1.weight" in name or ".1.bias" in name or "layer_norm" in name)
        if is_ln or name.endswith("bias"):
            out[name] = base[name]
            continue
        if len(shp) == 3:
            fan_in = shp[1] * 3 if "trans_conv" not in name else shp[0] * 1.5
        else:
            fan_in = shp[1]
        out[name] = (rng.standard_t(3, shp) / np.sqrt(3.0) * np.sqrt(2.0 / fan_in)).astype(np.float32)
    return out


def scaled_frame(frame, fscale=1.0, vscale=1.0, foffset=0.0):
    """foffset: a constant added to every feature-map channel (views then agree on a large common value: mean^2 >> variance, the regime in which a one-pass
    weighted variance cancels — mv_front_kernel)"""
    fr = dict(frame)
    fr["feat_fine_src"] = (frame["feat_fine_src"] * np.float32(fscale) + np.float32(foffset)).astype(np.float32)
    sp = dict(frame["support_fine"])
    f = sp["feature"].copy()
    f[:, 3:] = f[:, 3:] * np.float32(fscale) + np.float32(foffset)      # columns 0-2 are the colours (images in [0, 1]); 3.. are gathered from the feature maps (model.py:203-265)
    sp["feature"] = f
    fr["support_fine"] = sp
    fr["vis_featmaps"] = (frame["vis_featmaps"] * np.float32(vscale)).astype(np.float32)
    return fr


def eager64(cfg, frame, weights, rays, z, white=False):
    """render_rays in fp64 on the GPU (diff_render's eager restatement; exact KNN indices from the library)"""
    from nerf_loc_amd import diff_render as dr
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cast = lambda x: x.double() if torch.is_tensor(x) and x.is_floating_point() else x
    p = {k: cast(t(v)) for k, v in weights.items()}
    fr = {k: cast(t(frame[k])) for k in ("topk_Ks", "topk_poses", "topk_images", "feat_fine_src", "vis_featmaps")}
    fr.update({"near": float(cfg.near), "far": float(cfg.far), "support": {k: cast(t(v)) for k, v in frame["support_fine"].items()}})
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, "fp32")
    r.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
    r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
    outs = []
    with torch.no_grad():
        for i in range(0, rays["rays_o"].shape[0], 8):   # 8 rays at a time: the eager graph materialises (N, V, 195) / (N, 8, 285) tensors in fp64
            o = dr.render_rays_diff(p, fr, cast(t(rays["rays_o"][i:i + 8])), cast(t(rays["rays_d"][i:i + 8])), z[i:i + 8].to(dev).double(), cast(t(frame["pose"])),
                                    lambda q: r.knn(q.float(), 8)[1], white_bkgd=white)
            outs.append({k: v.cpu().numpy() for k, v in o.items() if torch.is_tensor(v)})
    return {k: np.concatenate([o[k] for o in outs]) for k in outs[0]}


KEYS = ("rgb", "depth", "weights", "depth_uncertainty", "feat")


def build(case):
    from tests.golden_cases import CASES
    if case == "c2":
        cfg = CONFIGS["c2"]
        frame = make_frame(cfg)
        rays = make_rays(cfg, frame)
        sel = np.arange(0, cfg.R, cfg.R // 64)[:64]
        rays = {k: (v[sel] if k in ("rays_o", "rays_d", "pixel_coordinates") else v) for k, v in rays.items()}
        return cfg, frame, rays
    cfg = CASES[case][0]
    frame = make_frame(cfg)
    return cfg, frame, make_rays(cfg, frame)


def run_one(cfg, frame0, rays, weights, fscale, vscale=1.0, precisions=("bf16x3", "f16mx"), threads=16, foffset=0.0):
    """-> {precision: {key: (max-rel vs oracle, l2-rel vs oracle, max-rel vs fp64)}}, cond {key: oracle vs fp64}"""
    from oracle import render_oracle as orc
    frame = scaled_frame(frame0, fscale, vscale, foffset)
    R = rays["rays_o"].shape[0]
    z = orc.sample_depths(cfg.S, torch.tensor(cfg.near), torch.tensor(cfg.far)).expand(R, cfg.S).contiguous()
    params = {k: torch.from_numpy(v) for k, v in weights.items()}
    rays_t = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in rays.items()}
    torch.set_num_threads(threads)
    with torch.no_grad():
        ref = orc.render_rays(params, orc.to_torch(frame), rays_t, cfg.S, knn_threads=threads)
    e64 = eager64(cfg, frame, weights, rays, z)
    cond = {k: rel_err(ref[k].numpy(), e64[k]) for k in KEYS}
    res = {}
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, precisions[0])
    r.load_weights(params)
    r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
    for prec in precisions:
        r.set_precision(prec)
        out = r.render_rays(rays["rays_o"], rays["rays_d"], frame["pose"][:3, 3], z_vals=z)
        torch.cuda.synchronize()
        ok_mask = bool(np.array_equal(out["mask"].cpu().numpy(), ref["mask"].numpy()))
        res[prec] = {k: (rel_err(out[k].cpu().numpy(), ref[k].numpy()), l2_rel(out[k].cpu().numpy(), ref[k].numpy()), rel_err(out[k].cpu().numpy(), e64[k])) for k in KEYS}
        res[prec]["mask_equal"] = ok_mask
    cond["logit_absmax"] = r.diagnostics()["logit_absmax"]
    return res, cond


def sweep(case, fscales=FSCALES, recipes=("normal", "student_t3"), vscales=(1.0,), precisions=("bf16x3", "f16mx", "fp32"), verbose=True, combos=None):
    """combos: explicit [(recipe, fscale, vscale), ...] instead of the product of the three lists"""
    cfg, frame, rays = build(case)
    rows = []
    if combos is None:
        combos = [(rc, fs, vs) for rc in recipes for fs in fscales for vs in vscales]
    wcache = {}
    for recipe, fs, vs in combos:
        wrec, _, off = recipe.partition("+off")   # "normal+off4": the normal weights, feature maps + 4
        if wrec not in wcache:
            wcache[wrec] = make_weights(cfg) if wrec == "normal" else heavy_tailed_weights(cfg)
        res, cond = run_one(cfg, frame, rays, wcache[wrec], fs, vs, precisions, foffset=float(off) if off else 0.0)
        for prec in precisions:
            worst_k = max(KEYS, key=lambda k: res[prec][k][0])
            row = {"case": case, "weights": recipe, "fscale": fs, "vscale": vs, "precision": prec, "worst_key": worst_k,
                   "max_rel": res[prec][worst_k][0], "l2_rel": max(res[prec][k][1] for k in KEYS), "vs_fp64": max(res[prec][k][2] for k in KEYS),
                   "oracle_vs_fp64": max(cond[k] for k in KEYS), "logit_absmax": cond["logit_absmax"], "mask_equal": res[prec]["mask_equal"],
                   "per_key": {k: res[prec][k] for k in KEYS}, "cond": cond}
            rows.append(row)
            if verbose:
                print(f"{case:9s} {recipe:15s} fscale {fs:8.4f} vscale {vs:5.2f} {prec:7s}: max-rel {row['max_rel']:.1e} ({worst_k}) l2 {row['l2_rel']:.1e} "
                      f"vs fp64 {row['vs_fp64']:.1e} | oracle vs fp64 {row['oracle_vs_fp64']:.1e} max|logit| {row['logit_absmax']:.3g} mask {'ok' if row['mask_equal'] else 'DIFFERS'}"
                      f"{'' if in_range(row) else '   (outside the validated range of ' + prec + ')'}"
                      f"{'' if row['max_rel'] < bar(row) and row['l2_rel'] < bar(row) else '   <-- ABOVE ITS BAR ' + format(bar(row), '.1e')}", flush=True)
    return rows


# What the sweep established (profiles/r5_scale_sweep.txt): the amplifier on this path is the attention over a sample's 8 neighbours — a logit error is
# (relative product error) x |logit|, and a softmax over nearly tied neighbours hands it on undamped.  The fused neural-point kernel reports the largest
# |logit| it scored (nl_frame_diagnostics); per mode there is a |logit| up to which the mode stays within 1e-4 of the CPU oracle on every scene of the sweep:
#   f16mx  (2^-16 per product): <= 100   (at 142: 3.5e-5 ... 8.1e-5, at 271: 1.2e-4)
#   bf16x3 (2^-17):             <= 500   (at 475: 4.2e-5, at ~1000: 8.5e-5 ... 1.7e-4)
#   fp32   (2^-24):             everywhere (<= 1.1e-5; one scene 8.8e-5 where the oracle itself is 1.8e-5 from fp64)
# The drop-in module's precision guard (ConditionalNeRF.LOGIT_LIMIT) escalates f16mx -> bf16x3 -> fp32 with exactly these limits, so the statement the tests
# make is: on EVERY scene of the sweep the mode the guard selects is within BASELINE's 1e-4 of the oracle (or 3 x the oracle's own distance to fp64, the bar
# of tools/forward_fuzz.py), and every mode is inside its validated range.
from nerf_loc_amd.conditional_nerf import ConditionalNeRF as _Module  # noqa: E402

LOGIT_LIMIT = dict(_Module.LOGIT_LIMIT)
LOGIT_LIMIT["fp32"] = float("inf")


def in_range(row):
    return row["logit_absmax"] <= LOGIT_LIMIT[row["precision"]]


def selected_mode(logit_absmax, start="f16mx"):
    mode = start
    while logit_absmax > LOGIT_LIMIT[mode]:
        mode = _Module._SAFER[mode]
    return mode


def bar(row):
    """BASELINE's 1e-4 against the fp32 oracle (3 x the oracle's own distance to fp64 where that is larger) for a mode inside its validated |logit| range;
    infinity outside it (such a row is reported, the mode the guard selects instead is what must hold)."""
    return max(1e-4, 3 * row["oracle_vs_fp64"]) if in_range(row) else float("inf")


def relaxed(row):
    return in_range(row) and 3 * row["oracle_vs_fp64"] > 1e-4


def check(rows):
    """-> list of failure descriptions: a mode above its bar inside its range, a mask mismatch, or a scene whose guard-selected mode misses the bar"""
    bad = []
    for r in rows:
        if not r["mask_equal"]:
            bad.append(("mask", r["case"], r["weights"], r["fscale"], r["precision"]))
        if r["max_rel"] >= bar(r) or r["l2_rel"] >= bar(r):
            bad.append(("above its bar", r["case"], r["weights"], r["fscale"], r["vscale"], r["precision"], r["max_rel"], r["l2_rel"], bar(r), r["logit_absmax"]))
    by_scene = {}
    for r in rows:
        by_scene.setdefault((r["case"], r["weights"], r["fscale"], r["vscale"]), {})[r["precision"]] = r
    for key, modes in by_scene.items():
        any_row = next(iter(modes.values()))
        sel = selected_mode(any_row["logit_absmax"])
        if sel in modes and not in_range(modes[sel]):
            bad.append(("selected mode out of range", key, sel))
    return bad


if __name__ == "__main__":
    calibration = "--calibration" in sys.argv   # part 2 of profiles/r5_scale_sweep.txt: the |logit| ranges the precision guard's limits come from
    cases = [a for a in sys.argv[1:] if not a.startswith("--")] or (["w256s128"] if calibration else ["w256s128", "c2"])
    rows = []
    for c in cases:
        if calibration:
            rows += sweep(c, fscales=(1.5, 2.0, 2.5, 3.0, 3.5, 4.0, 5.0, 6.0))
            rows += sweep(c, combos=[("normal+off4", 1.0, 1.0), ("normal+off32", 1.0, 1.0), ("normal+off100", 1.0, 1.0)])
        else:
            rows += sweep(c, vscales=(1.0,) if c == "c2" else (1.0, 8.0))
    bad = check(rows)
    for b in bad:
        print("FAIL", b)
    outside = sum(not in_range(r) for r in rows)
    print(f"rows: {len(rows)}; failures: {len(bad)}; rows outside their mode's validated |logit| range (the guard escalates there): {outside}; "
          f"rows held to 3 x (oracle vs fp64) > 1e-4: {sum(relaxed(r) for r in rows)}")
