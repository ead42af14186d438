"""feat_mlp.0 + compositing as one kernel (tgemm.hip: feat_comp_mx_kernel; f16mx, W = 256): ray lengths for both workgroup shapes (8 waves: S = 32, 64, 128, 256;
6 waves: S = 96, 192) with ray counts that leave the last workgroup partly empty (waves past the end, rays that must not be written), against the CPU oracle at
BASELINE's 1e-4 and against the staged path of the same library (the chain kernel's feat_mlp.0 + composite_kernel, which `intermediates=True` selects)."""
import numpy as np
import pytest
import torch

from tests.util import l2_rel, rel_err

pytestmark = pytest.mark.gpu
KEYS = ("rgb", "depth", "weights", "depth_uncertainty", "feat")


def _case(S, R):
    from nerf_loc_amd.synth import SceneConfig, make_frame, make_rays, make_weights
    cfg = SceneConfig(f"fc{S}", R=R, S=S, W=256, V=4, H=48, Wimg=64, seed=100 + S)
    frame = make_frame(cfg)
    return cfg, frame, make_rays(cfg, frame), make_weights(cfg)


def _renderer(cfg, frame, weights):
    from nerf_loc_amd.renderer import HipRenderer
    r = HipRenderer(cfg.W, cfg.C, cfg.S, "f16mx")
    r.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
    r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
    return r


@pytest.mark.parametrize("S,R", [(32, 13), (64, 7), (96, 5), (128, 3), (128, 8), (192, 5), (256, 3)])
def test_fused_feature_compositing_matches_oracle_and_staged_path(S, R):
    from oracle import render_oracle as orc
    cfg, frame, rays, weights = _case(S, R)
    r = _renderer(cfg, frame, weights)
    z = orc.sample_depths(cfg.S, torch.tensor(cfg.near), torch.tensor(cfg.far)).expand(R, cfg.S).contiguous()
    qc = frame["pose"][:3, 3]
    out = r.render_rays(rays["rays_o"], rays["rays_d"], qc, z_vals=z)
    staged = r.render_rays(rays["rays_o"], rays["rays_d"], qc, z_vals=z, intermediates=True)   # fp32 feature_agg rows wanted: the chain kernel runs feat_mlp.0
    noweights = r.render_rays(rays["rays_o"], rays["rays_d"], qc, z_vals=z, want_weights=False)   # the samples' weights go through the library's scratch rows
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = orc.render_rays({k: torch.from_numpy(v) for k, v in weights.items()}, orc.to_torch(frame), orc.to_torch(rays), cfg.S)
    assert bool((out["mask"].cpu() == ref["mask"]).all())
    for k in KEYS:
        a, b = out[k].cpu().numpy(), ref[k].numpy()
        assert np.isfinite(a).all(), k
        assert rel_err(a, b) < 1e-4, (k, rel_err(a, b))
        assert l2_rel(a, b) < 1e-4, (k, "l2", l2_rel(a, b))
    # the same library, feat_mlp.0 in the chain kernel (three-term split-bf16) + composite_kernel: every ray, not just the largest entries
    fa, fb = out["feat"].cpu().numpy().astype(np.float64), staged["feat"].cpu().numpy().astype(np.float64)
    per_ray = np.abs(fa - fb).max(axis=1) / max(np.abs(fb).max(), 1e-30)
    assert per_ray.max() < 5e-5, per_ray
    assert not torch.equal(out["feat"], staged["feat"]), "the fused kernel (f16mx product) did not run: the features are bit-identical to the staged path's"
    assert "weights" not in noweights
    for k in ("rgb", "depth", "depth_uncertainty", "feat"):
        assert torch.equal(noweights[k], out[k]), k


def test_fused_feature_compositing_is_batch_invariant():
    """A ray's features do not depend on the batch it is rendered in (its group, its wave's slot in the group, the workgroup that walks it)."""
    from oracle import render_oracle as orc
    cfg, frame, rays, weights = _case(128, 37)
    r = _renderer(cfg, frame, weights)
    z = orc.sample_depths(cfg.S, torch.tensor(cfg.near), torch.tensor(cfg.far)).expand(cfg.R, cfg.S).contiguous()
    qc = frame["pose"][:3, 3]
    full = r.render_rays(rays["rays_o"], rays["rays_d"], qc, z_vals=z)
    for sel in (slice(0, 1), slice(5, 18), slice(30, 37)):
        part = r.render_rays(rays["rays_o"][sel], rays["rays_d"][sel], qc, z_vals=z[sel])
        for k in KEYS:
            assert torch.equal(part[k], full[k][sel]), (k, sel)


def test_misaligned_weights_buffer_takes_the_old_path():
    """The fused kernel reads the samples' weights as 16-byte rows; the C-ABI never asked for an aligned `weights` buffer, so a misaligned one must still render
    (the library keeps the chain kernel's feat_mlp.0 + composite_kernel for that call) and agree with the aligned call."""
    from oracle import render_oracle as orc
    cfg, frame, rays, weights = _case(128, 6)
    r = _renderer(cfg, frame, weights)
    R, S, C = cfg.R, cfg.S, cfg.C
    z = orc.sample_depths(S, torch.tensor(cfg.near), torch.tensor(cfg.far)).expand(R, S).contiguous()
    qc = frame["pose"][:3, 3]
    ref = r.render_rays(rays["rays_o"], rays["rays_d"], qc, z_vals=z)
    dev = ref["rgb"].device
    wbuf = torch.zeros(R * S + 4, device=dev)
    bufs = {"rgb": torch.empty(R, 3, device=dev), "depth": torch.empty(R, device=dev), "weights": wbuf[1:1 + R * S].view(R, S),
            "mask": torch.empty(R, dtype=torch.uint8, device=dev), "depth_uncertainty": torch.empty(R, device=dev), "feat": torch.empty(R, C, device=dev)}
    assert bufs["weights"].data_ptr() % 16 == 4 and bufs["weights"].is_contiguous()
    out = r.render_rays(rays["rays_o"], rays["rays_d"], qc, z_vals=z, out_buffers=bufs)
    torch.cuda.synchronize()
    # (not bit-equal: the aligned call hands feature_agg to the ray U-Net as split-FP16 fragments, the fallback as split-bf16 ones — two 16-bit-plus roundings of the
    # same rows)
    for k in ("rgb", "depth", "weights", "depth_uncertainty", "feat"):
        a, b = out[k].cpu().numpy().astype(np.float64), ref[k].cpu().numpy().astype(np.float64)
        assert np.abs(a - b).max() / np.abs(b).max() < 5e-5, k
    assert float(wbuf[0]) == 0.0 and float(wbuf[-1]) == 0.0   # nothing written outside the view
