"""HIP backward kernels, first slice: nl_composite_backward and nl_knn_backward against PyTorch autograd of the same
expressions (fp32 and, as the yardstick for rounding, fp64).  End to end they run inside the gradient path: the pose / training-step
gradient tests of tests/test_diff_render.py compare that path with the reference's own autograd goldens."""
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
