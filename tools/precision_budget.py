#!/usr/bin/env python3
"""Precision budget of the GEMM-shaped stages, measured on the CPU.

For every GEMM group of the render path and every candidate arithmetic the render of a small ray sample is repeated with ONLY that
group's products emulated in the candidate arithmetic (everything else exact fp32), and the error of the outputs against the all-fp32
render is recorded (max-rel-to-max, the metric of the parity tests).  A second pass evaluates whole assignments.
Arithmetics (a = activation, w = weight; hi = round-to-nearest 16-bit value, lo = rounded remainder; fp32 accumulate):
  bf16x3   a_hi w_hi + a_lo w_hi + a_hi w_lo   (the parity mode, 3 MFMAs)     f16x3  the same split in fp16
  bf16x2a  (a_hi + a_lo) w_hi                   (2 MFMAs, weights single)      f16x2a
  bf16x2w  a_hi (w_hi + w_lo)                   (2 MFMAs, activations single)  f16x2w
  bf16x1   a_hi w_hi                            (1 MFMA)                       f16x1
  f16mx8   a_hi w_hi (fp16) + q8(a_lo) q8(w_hi) + q8(a_hi) q8(w_lo)   (round 4, VERDICT r3 item 2: the two cross terms on MX-scaled FP8 —
           e4m3 elements, one power-of-two scale per 32 consecutive k, the format of gfx950's v_mfma_scale_f32_32x32x64_f8f6f4 at twice the
           bf16 rate: 1 + 2 x 0.5 = 2.0 MFMA-equivalents per product instead of 3)
Uses nerf_loc_amd.diff_render's functional forward (the restatement checked against the reference's autograd goldens) with its
Linear / conv calls intercepted by name.  CPU only; nothing here is on the product path.

  python tools/precision_budget.py [c2|c1] [rays]      -> prints the table, writes profiles/r3_precision_budget.json
"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_loc_amd import diff_render as dr  # noqa: E402
from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights  # noqa: E402
from tests.util import knn_bruteforce  # noqa: E402


def split(x, dt):
    hi = x.to(dt).float()
    lo = (x - hi).to(dt).float()
    return hi, lo


def mx8(x):
    """x (..., K) -> its MX-FP8 image, dequantised: blocks of 32 along K share a power-of-two scale chosen so the block's largest magnitude
    lands at or below e4m3's 448 (no saturation), elements rounded to nearest e4m3."""
    K = x.shape[-1]
    pad = (-K) % 32
    xp = F.pad(x, (0, pad)) if pad else x
    b = xp.reshape(*xp.shape[:-1], -1, 32)
    m = b.abs().amax(-1, keepdim=True)
    e = torch.ceil(torch.log2(m.clamp_min(1e-38) / 448.0))
    sc = torch.exp2(e)
    q = (b / sc).to(torch.float8_e4m3fn).float() * sc
    q = torch.where(m > 0, q, torch.zeros_like(q))
    return q.reshape(xp.shape)[..., :K]


def mx6(x, fmt="e2m3", ref=None, shift=0.0):
    """x (..., K) -> its MX-FP6 image, dequantised: blocks of 32 along K share the power-of-two scale 2^(floor(log2 max) - 2) (the block's largest magnitude lands in
    [4, 8): e2m3's 7.5 saturates at most one half-ulp), elements rounded to nearest e2m3 (grid 0.125 below 2, 0.25 below 4, 0.5 up to 7.5)."""
    K = x.shape[-1]
    pad = (-K) % 32
    xp = F.pad(x, (0, pad)) if pad else x
    b = xp.reshape(*xp.shape[:-1], -1, 32)
    m = b.abs().amax(-1, keepdim=True)
    if ref is not None:   # the scale of another tensor's blocks, shifted (the kernel: the residual image takes the hi image's scale x 2^-11)
        rp = F.pad(ref, (0, pad)) if pad else ref
        m = rp.reshape(*rp.shape[:-1], -1, 32).abs().amax(-1, keepdim=True)
    sc = torch.exp2(torch.floor(torch.log2(m.clamp_min(1e-38))) - 2.0 + shift)
    a = (b / sc).abs()
    if fmt == "e2m3":
        step = torch.where(a < 2.0, torch.full_like(a, 0.125), torch.where(a < 4.0, torch.full_like(a, 0.25), torch.full_like(a, 0.5)))
        q = (torch.round(a / step) * step).clamp_max(7.5)
    else:   # e2m1 (fp4): 0, .5, 1, 1.5, 2, 3, 4, 6
        step = torch.where(a < 2.0, torch.full_like(a, 0.5), torch.where(a < 4.0, torch.full_like(a, 1.0), torch.full_like(a, 2.0)))
        q = (torch.round(a / step) * step).clamp_max(6.0)
    q = torch.sign(b) * q * sc
    q = torch.where(m > 0, q, torch.zeros_like(q))
    return q.reshape(xp.shape)[..., :K]


def emu_linear(x, w, mode):
    """x (..., K) @ w (N, K)^T in the emulated arithmetic, fp32 accumulate (torch's fp32 matmul on the rounded operands)."""
    if mode == "fp32":
        return x @ w.t()
    if mode == "f16mx8":
        xh, xl = split(x, torch.float16)
        wh, wl = split(w, torch.float16)
        return xh @ wh.t() + mx8(xl) @ mx8(wh).t() + mx8(xh) @ mx8(wl).t()
    if mode in ("f16mx6", "f16mx4"):
        f = "e2m3" if mode == "f16mx6" else "e2m1"
        xh, xl = split(x, torch.float16)
        wh, wl = split(w, torch.float16)
        if mode == "f16mx6":   # as the kernel does it: the activations' residual image on the hi image's block scale x 2^-11
            return xh @ wh.t() + mx6(x - xh, f, ref=xh, shift=-11.0) @ mx6(wh, f).t() + mx6(xh, f) @ mx6(w - wh, f).t()
        return xh @ wh.t() + mx6(x - xh, f) @ mx6(wh, f).t() + mx6(xh, f) @ mx6(w - wh, f).t()
    dt = torch.bfloat16 if mode.startswith("bf16") else torch.float16
    kind = mode[mode.index("x"):]
    xh, xl = split(x, dt)
    wh, wl = split(w, dt)
    if kind == "x1":
        return xh @ wh.t()
    if kind == "x2a":
        return xh @ wh.t() + xl @ wh.t()
    if kind == "x2w":
        return xh @ wh.t() + xh @ wl.t()
    if kind == "x3":
        return xh @ wh.t() + xl @ wh.t() + xh @ wl.t()
    raise ValueError(mode)


# GEMM groups = what one kernel (or one chunk program of a kernel) multiplies; name prefixes of the Linear layers / conv blocks
GROUPS = {
    "decoders (mv_vis)": ["multiview_aggregator.dist_decoder."],
    "out_fc.0": ["multiview_aggregator.out_fc.0"],
    "out_fc.2": ["multiview_aggregator.out_fc.2"],
    "ray_diff_fc": ["ray_diff_fc."],
    "base_mlp.0": ["base_mlp.0"],
    "base_mlp.2": ["base_mlp.2"],
    "base_mlp.4": ["base_mlp.4"],
    "w_qs": ["base_mlp_attn.w_qs"],
    "w_ks": ["base_mlp_attn.w_ks"],
    "w_vs": ["base_mlp_attn.w_vs"],
    "attn fc": ["base_mlp_attn.fc"],
    "unet conv1": ["ray_unet.conv1"],
    "unet inner (conv2..trans_conv1)": ["ray_unet.conv2", "ray_unet.conv3", "ray_unet.trans_conv3", "ray_unet.trans_conv2", "ray_unet.trans_conv1"],
    "unet conv_out": ["ray_unet.conv_out"],
    "feat_mlp.0": ["feat_mlp.0"],
    "feat_mlp.2": ["feat_mlp.2"],
    "blend.0": ["rgb_blending_mlp.0"],
}
MODES = ["bf16x1", "bf16x2a", "bf16x2w", "bf16x3", "f16x1", "f16x2a", "f16x2w", "f16mx8"]
if os.environ.get("BUDGET_MODES"):   # e.g. BUDGET_MODES=f16mx8: only these columns (one render per group and mode)
    MODES = os.environ["BUDGET_MODES"].split(",")


class Emu:
    def __init__(self):
        self.assign = {}

    def mode_of(self, name):
        for pre, m in self.assign.items():
            if name.startswith(pre):
                return m
        return "fp32"


EMU = Emu()


def patched_lin(p, name, x, bias=True):
    m = EMU.mode_of(name)
    y = emu_linear(x, p[f"{name}.weight"], m)
    return y + p[f"{name}.bias"] if bias else y


def patched_unet(p, x):
    """diff_render._ray_unet with the convolutions as im2col products so they can be emulated (conv == linear over the taps)."""
    def conv(name, t, transposed):
        w, b = p[f"ray_unet.{name}.0.weight"], p[f"ray_unet.{name}.0.bias"]
        m = EMU.mode_of(f"ray_unet.{name}")
        if m == "fp32":
            return F.conv_transpose1d(t, w, b, stride=2, padding=1, output_padding=1) if transposed else F.conv1d(t, w, b, stride=1, padding=1)
        R, Ci, L = t.shape
        if not transposed:
            cols = F.pad(t, (1, 1)).unfold(2, 3, 1)                      # (R, Ci, L, 3)
            y = emu_linear(cols.permute(0, 2, 1, 3).reshape(R * L, Ci * 3), w.reshape(w.shape[0], -1), m)
            return y.view(R, L, -1).permute(0, 2, 1) + b[None, :, None]
        # stride-2 transposed conv (k = 3, pad 1, output_padding 1): y[2m] = W[:, :, 1]^T x[m];  y[2m+1] = W[:, :, 2]^T x[m] + W[:, :, 0]^T x[m+1]
        xt = t.permute(0, 2, 1)                                           # (R, L, Ci)
        xn = F.pad(xt, (0, 0, 0, 1))[:, 1:]
        even = emu_linear(xt.reshape(R * L, Ci), w[:, :, 1].t(), m)
        odd = emu_linear(torch.cat([xt, xn], -1).reshape(R * L, 2 * Ci), torch.cat([w[:, :, 2].t(), w[:, :, 0].t()], 1), m)
        y = torch.stack([even, odd], 1).view(R, L, 2, -1).reshape(R, 2 * L, -1)
        return y.permute(0, 2, 1) + b[None, :, None]

    def block(name, t, transposed=False):
        g, be = p[f"ray_unet.{name}.1.weight"], p[f"ray_unet.{name}.1.bias"]
        return F.elu(F.layer_norm(conv(name, t, transposed), tuple(g.shape), g, be, eps=1e-5))
    c1 = F.max_pool1d(block("conv1", x), 2)
    c2 = F.max_pool1d(block("conv2", c1), 2)
    c3 = F.max_pool1d(block("conv3", c2), 2)
    x0 = block("trans_conv3", c3, True)
    x1 = block("trans_conv2", torch.cat([c2, x0], 1), True)
    x2 = block("trans_conv1", torch.cat([c1, x1], 1), True)
    return block("conv_out", torch.cat([x, x2], 1))


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    torch.set_num_threads(os.cpu_count() or 8)
    if name in CONFIGS:
        cfg = CONFIGS[name]
    else:   # a golden-case scene (tests/golden_cases.py), e.g. w256s128
        from tests.golden_cases import CASES
        cfg = CASES[name][0]
        n_rays = min(n_rays, cfg.R)
    frame = make_frame(cfg)
    rays = make_rays(cfg, frame, R=n_rays)
    weights = make_weights(cfg)
    # round 5: the same budget on inputs that are not O(1) — BUDGET_FSCALE multiplies the feature maps and the support features,
    # BUDGET_VSCALE the DepthFusionNet maps, BUDGET_WEIGHTS=student_t3 draws heavy-tailed weights (tools/scale_sweep.py holds the recipes)
    if os.environ.get("BUDGET_FSCALE") or os.environ.get("BUDGET_VSCALE") or os.environ.get("BUDGET_WEIGHTS"):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import scale_sweep as ss
        frame = ss.scaled_frame(frame, float(os.environ.get("BUDGET_FSCALE", "1")), float(os.environ.get("BUDGET_VSCALE", "1")))
        if os.environ.get("BUDGET_WEIGHTS") == "student_t3":
            weights = ss.heavy_tailed_weights(cfg)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    p = {k: t(v) for k, v in weights.items()}
    fr = {k: t(frame[k]) for k in ("topk_Ks", "topk_poses", "topk_images", "feat_fine_src", "vis_featmaps")}
    fr.update({"near": float(cfg.near), "far": float(cfg.far), "support": {k: t(v) for k, v in frame["support_fine"].items()}})
    o, d = t(rays["rays_o"]), t(rays["rays_d"])
    lin = torch.linspace(0, 1, cfg.S)
    z = (cfg.near * (1 - lin) + cfg.far * lin).expand(n_rays, cfg.S).contiguous()
    pose = t(frame["pose"])
    with torch.no_grad():
        xyz = (o[:, None, :] + d[:, None, :] * z[..., None]).reshape(-1, 3)
        idx = knn_bruteforce(fr["support"]["xyz"], 8, chunk=1024)(xyz)
    dr._lin = patched_lin
    dr._ray_unet = patched_unet
    keys = ("rgb", "depth", "weights", "depth_uncertainty", "feat")

    def render(assign):
        EMU.assign = assign
        with torch.no_grad():
            out = dr.render_rays_diff(p, fr, o, d, z, pose, lambda q: idx)
        return {k: out[k] for k in keys}

    t0 = time.time()
    ref = render({})
    print(f"{name}: {n_rays} rays x {cfg.S} samples, W={cfg.W}, V={cfg.V}; one render {time.time() - t0:.1f} s", flush=True)
    table = {}
    print(f"{'group':34s} " + " ".join(f"{m:>9s}" for m in MODES) + "    (worst output, max-rel-to-max)")
    for g, pres in GROUPS.items():
        if os.environ.get("BUDGET_GROUPS") and not any(t in g for t in os.environ["BUDGET_GROUPS"].split(",")):
            continue
        row = {}
        for m in MODES:
            out = render({pre: m for pre in pres})
            errs = {k: rel(out[k], ref[k]) for k in keys}
            row[m] = {"worst": max(errs.values()), "by_output": errs}
        table[g] = row
        print(f"{g:34s} " + " ".join(f"{row[m]['worst']:9.1e}" for m in MODES), flush=True)
    # whole assignments
    combos = {
        "all bf16x3 (parity mode)": {pre: "bf16x3" for pres in GROUPS.values() for pre in pres},
        "all bf16x1": {pre: "bf16x1" for pres in GROUPS.values() for pre in pres},
        "all f16x1": {pre: "f16x1" for pres in GROUPS.values() for pre in pres},
        "all f16x2a": {pre: "f16x2a" for pres in GROUPS.values() for pre in pres},
        "all bf16x2a": {pre: "bf16x2a" for pres in GROUPS.values() for pre in pres},
        "all f16mx8": {pre: "f16mx8" for pres in GROUPS.values() for pre in pres},
        "f16mx8, decoders f16x3-like (bf16x3)": {pre: ("bf16x3" if pre.startswith("multiview_aggregator.dist_decoder") else "f16mx8") for pres in GROUPS.values() for pre in pres},
    }
    if os.environ.get("BUDGET_MODES"):
        combos = {k: v for k, v in combos.items() if any(m in k for m in MODES) or "parity" in k}
    base = dict(combos["all bf16x3 (parity mode)"])
    for label, pres, m in (("x3, w_ks bf16x1", ["base_mlp_attn.w_ks"], "bf16x1"), ("x3, w_ks f16x1", ["base_mlp_attn.w_ks"], "f16x1"),
                           ("x3, w_ks+w_qs bf16x1", ["base_mlp_attn.w_ks", "base_mlp_attn.w_qs"], "bf16x1"),
                           ("x3, w_ks+w_qs f16x1", ["base_mlp_attn.w_ks", "base_mlp_attn.w_qs"], "f16x1"),
                           ("x3, w_ks+w_qs f16x2a", ["base_mlp_attn.w_ks", "base_mlp_attn.w_qs"], "f16x2a"),
                           ("x3, point branch f16mx8", ["base_mlp.", "base_mlp_attn.w_ks", "base_mlp_attn.w_vs"], "f16mx8"),
                           ("x3, conv_out f16mx8", ["ray_unet.conv_out"], "f16mx8"),
                           ("x3, point branch f16mx6", ["base_mlp.2", "base_mlp.4", "base_mlp_attn.w_ks", "base_mlp_attn.w_vs"], "f16mx6"),
                           ("x3, point branch f16mx8 (L2, L3, k, v)", ["base_mlp.2", "base_mlp.4", "base_mlp_attn.w_ks", "base_mlp_attn.w_vs"], "f16mx8"),
                           ("x3, point branch f16mx4", ["base_mlp.2", "base_mlp.4", "base_mlp_attn.w_ks", "base_mlp_attn.w_vs"], "f16mx4")):
        if os.environ.get("BUDGET_MODES") and m not in MODES:
            continue
        a = dict(base)
        a.update({pre: m for pre in pres})
        combos[label] = a
    res = {}
    for label, a in combos.items():
        out = render(a)
        errs = {k: rel(out[k], ref[k]) for k in keys}
        res[label] = errs
        print(f"{label:34s} worst {max(errs.values()):.1e}  " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()), flush=True)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", os.environ.get("BUDGET_OUT", f"r3_precision_budget_{name}.json")), "w") as fh:
        json.dump({"config": name, "rays": n_rays, "metric": "max |x - fp32| / max |fp32| over the worst of rgb, depth, weights, depth_uncertainty, feat",
                   "single_group": table, "assignments": res}, fh, indent=1)


if __name__ == "__main__":
    main()
