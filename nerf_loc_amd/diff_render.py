"""Gradient path of the conditional-NeRF renderer.

The HIP renderer (`renderer.HipRenderer`) is forward-only.  The one caller of the reference that differentiates through `render_rays`
is `PoseOptimizer` (pose_optimizer.py:131-168): 512 rays x <= 50 Adam steps on an se(3) vector, loss = masked MSE of the rendered
feature (or colour) against the query image's, gradient to the camera pose only — every network weight is frozen.  At that size the
render is ~65 k samples, so the gradient path is written as fp32 PyTorch ops with autograd on the tensors' device (the GPU in
production), NOT as hand-written backward kernels; the exact K-nearest-neighbour search, which is not differentiable (indices) and
is the one step eager PyTorch cannot do at this size, stays the HIP kernel (`HipRenderer.knn`).  `ConditionalNeRF.render_rays`
routes here only when autograd is enabled and an input requires grad; the inference path never touches this module.

Every function cites the reference lines it follows (paths relative to /root/reference/nerf_loc/models/).  Parity: the gradients of
both PoseOptimizer losses w.r.t. the pose and the rays are checked against the reference's own autograd (tests/golden/grad_*.npz,
made by tools/gen_golden.py) in tests/test_diff_render.py.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import ctypes as ct

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------- HIP backward kernels (first slice)
def _hip_ok(*tensors) -> bool:
    return all(t is not None and t.is_cuda and t.dtype == torch.float32 for t in tensors)


def composite_eager(sigma: Tensor, rgb_s: Tensor, ft: Tensor, z_vals: Tensor, white_bkgd: bool):
    """conditional_nerf/model.py:544-560, 597 in plain torch ops: -> (rgb, depth, depth_uncertainty, feat, weights); last interval 1e2."""
    deltas = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], 1e2 * torch.ones_like(z_vals[:, :1])], -1)
    alphas = 1 - torch.exp(-deltas * sigma)
    T = torch.cumprod(torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas], -1)[:, :-1], -1)
    wts = alphas * T
    rgb = (wts[..., None] * rgb_s).sum(1)
    if white_bkgd:
        rgb = rgb + (1 - wts.sum(1)[:, None])
    depth = (wts * z_vals).sum(1)
    feat = (wts[..., None] * ft).sum(1)
    unc = (wts * (z_vals - depth[:, None]) ** 2).sum(1)
    return rgb, depth, unc, feat, wts


class CompositeFn(torch.autograd.Function):
    """Alpha compositing with the HIP backward kernel `nl_composite_backward` (the forward is the plain torch expression above; nothing but
    the inputs is saved — the kernel recomputes the transmittance).  Inputs: sigma (R,S), rgb_s (R,S,3), ft (R,S,C), z_vals (R,S) [constant]."""

    @staticmethod
    def forward(ctx, sigma, rgb_s, ft, z_vals, white_bkgd):
        ctx.white = bool(white_bkgd)
        sigma, rgb_s, ft, z_vals = sigma.contiguous(), rgb_s.contiguous(), ft.contiguous(), z_vals.contiguous()
        ctx.save_for_backward(sigma, rgb_s, ft, z_vals)
        return composite_eager(sigma, rgb_s, ft, z_vals, ctx.white)

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_unc, g_feat, g_wts):
        from . import _lib as L
        sigma, rgb_s, ft, z = ctx.saved_tensors
        R, S = sigma.shape
        C = ft.shape[-1]
        gs, grs, gft = torch.empty_like(sigma), torch.empty_like(rgb_s), torch.empty_like(ft)
        ptr = lambda t: None if t is None else t.contiguous().data_ptr()   # noqa: E731
        keep = [None if g is None else g.contiguous() for g in (g_rgb, g_depth, g_unc, g_feat, g_wts)]
        st = torch.cuda.current_stream(sigma.device).cuda_stream
        with torch.cuda.device(sigma.device):
            L.check(L.load().nl_composite_backward(z.data_ptr(), sigma.data_ptr(), rgb_s.data_ptr(), ft.data_ptr(), R, S, C, int(ctx.white),
                                                   *[ptr(g) for g in keep], gs.data_ptr(), grs.data_ptr(), gft.data_ptr(), st), "nl_composite_backward")
        return gs, grs, gft, None, None


class KnnDist2Fn(torch.autograd.Function):
    """Squared distances to the given neighbours with the HIP backward kernel `nl_knn_backward` (= the reference's
    KNearestNeighborBackwardKernel, ops/knn/src/knn.cu:449-490)."""

    @staticmethod
    def forward(ctx, xyz, sp_xyz, idx):
        xyz, sp_xyz = xyz.contiguous(), sp_xyz.contiguous()
        idx32 = idx.to(torch.int32).contiguous()
        ctx.save_for_backward(xyz, sp_xyz, idx32)
        off = xyz[:, None, :] - sp_xyz[idx.long()]
        return (off * off).sum(-1)

    @staticmethod
    def backward(ctx, g_d2):
        from . import _lib as L
        xyz, sp, idx32 = ctx.saved_tensors
        N, K = idx32.shape
        M = sp.shape[0]
        g = g_d2.contiguous()
        gx = torch.empty_like(xyz)
        gsp = torch.zeros_like(sp) if ctx.needs_input_grad[1] else None
        st = torch.cuda.current_stream(xyz.device).cuda_stream
        with torch.cuda.device(xyz.device):
            L.check(L.load().nl_knn_backward(xyz.data_ptr(), sp.data_ptr(), idx32.data_ptr(), g.data_ptr(), N, K, M, gx.data_ptr(),
                                             None if gsp is None else gsp.data_ptr(), st), "nl_knn_backward")
        return gx, gsp, None


class PointBranchFn(torch.autograd.Function):
    """The neural-point branch (model.py:372-427) with FROZEN weights as one autograd node on the HIP library: forward = the fused kernels
    of `nl_point_mlp`, backward = `nl_point_mlp_backward` (staged recompute + transposed-weight GEMMs; nothing is saved but the inputs).
    Used by the pose-refinement gradient path (eval mode: the weights and the support table are constants, pose_optimizer.py:131-168);
    training keeps the eager graph, which also reaches the weights.  `renderer`: a HipRenderer holding the same weights and frame."""

    @staticmethod
    def forward(ctx, xyz, dirs, G, renderer, K):
        xyz, G = xyz.contiguous(), G.contiguous()
        dirs = None if dirs is None else dirs.contiguous()
        ctx.r, ctx.K, ctx.has_dir, ctx.gen = renderer, int(K), dirs is not None, renderer.state_gen
        fa, d2, idx = renderer.point_mlp(xyz, dirs, G, K=int(K))
        ctx.save_for_backward(xyz, G, d2, idx, *([dirs] if dirs is not None else []))
        return fa

    @staticmethod
    def backward(ctx, g_fa):
        _same_state(ctx)
        xyz, G, d2, idx = ctx.saved_tensors[:4]
        dirs = ctx.saved_tensors[4] if ctx.has_dir else None
        gx, gd, gg = ctx.r.point_mlp_backward(xyz, dirs, G, g_fa.contiguous(), K=ctx.K, knn=(d2, idx))
        return gx, gd, gg, None, None


def _same_state(ctx):
    """The library nodes keep no activations: their backward recomputes from the renderer's CURRENT weights and frame tables, which therefore must be
    the forward call's (one training / refinement step = forward and backward against one frame)."""
    if ctx.r.state_gen != ctx.gen:
        raise RuntimeError("the HipRenderer's weights or frame were replaced between the forward and the backward pass of this autograd node")


# RenderFn keeps the staged forward's activations for the backward call when they fit this many bytes (one chunk: ~90 KB per sample at W = 256, i.e. a
# PoseOptimizer or training batch); 0 = never (fused forward + recompute)
KEEP_BYTES = 12 << 30
# ... and replays the pair as HIP graphs from the second step of a shape against one frame on (frozen weights; renderer.GraphedKeep)
USE_GRAPHS = True


class RenderFn(torch.autograd.Function):
    """ConditionalNeRF.render_rays (model.py:472-600) as ONE autograd node on the HIP library: forward = the fused inference path
    (`nl_render_rays`), backward = `nl_render_rays_backward` — the whole path backwards in one call, every per-sample quantity recomputed once
    (DESIGN.md §5.15).  Inputs: rays_o, rays_d (R,3), query centre (3,), z_vals (R,S) constants, renderer, white_bkgd, then — a training step —
    the frame tensors (feature maps, DepthFusionNet maps, support features) and the RENDER_PARAMS tensors whose gradients the call also
    produces (the renderer must hold their current values).  Returns (rgb, depth, depth_uncertainty, feat, weights, mask)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, qc, z_vals, renderer, white_bkgd, feat_maps=None, vis_maps=None, sp_feature=None, *params):
        o, d, z = rays_o.contiguous(), rays_d.contiguous(), z_vals.contiguous()
        ctx.r, ctx.white, ctx.gen = renderer, bool(white_bkgd), renderer.state_gen
        ctx.kept = None
        train = len(params) > 0
        # params = the RENDER_PARAMS tensors [+ beta_mlp.0.weight, beta_mlp.0.bias: the training-mode uncertainty head, keep / kept pair only]
        ctx.has_beta = len(params) == len(RENDER_PARAMS) + 2
        if ctx.has_beta:
            kept = renderer.render_rays_keep(o, d, z, qc, white_bkgd=bool(white_bkgd), train=True, max_bytes=KEEP_BYTES or None,
                                             beta_head=(params[-2].detach(), params[-1].detach()))
            if kept is None:
                raise RuntimeError("RenderFn with the uncertainty head needs the batch to fit one workspace chunk (diff_render.KEEP_BYTES)")
            out, ctx.kept = kept
            ctx.save_for_backward(o, d, qc, z)
            ctx.mark_non_differentiable(out["mask"])
            return out["rgb"], out["depth"], out["depth_uncertainty"], out["feat"], out["weights"], out["mask"], out["beta"]
        # frozen weights, a shape this frame has seen before (a refinement loop): the pair replayed as two HIP graphs
        ctx.graph = None
        if not train and USE_GRAPHS and KEEP_BYTES and qc.is_cuda:
            gk = renderer.graphed_keep(o.shape[0], bool(white_bkgd), max_bytes=KEEP_BYTES)
            lease = gk.acquire() if gk is not None else None
            if lease is not None:
                out = gk.forward(o, d, z, qc.detach().float())
                ctx.graph, ctx.lease, ctx.graph_stamp = gk, lease, gk.fwd_count
                ctx.save_for_backward(o, d, qc, z)
                ctx.mark_non_differentiable(out["mask"])
                return out["rgb"], out["depth"], out["depth_uncertainty"], out["feat"], out["weights"], out["mask"]
        # small batches (a PoseOptimizer / training step): the staged forward whose activations the backward call reuses, when they fit KEEP_BYTES
        kept = renderer.render_rays_keep(o, d, z, qc, white_bkgd=bool(white_bkgd), train=train, max_bytes=KEEP_BYTES) if KEEP_BYTES else None
        if kept is not None:
            out, ctx.kept = kept
            ctx.save_for_backward(o, d, qc, z)
        else:
            # (a query centre on the device goes down as per-ray rows: no device-to-host copy, no synchronisation point in the loop)
            qarg = qc.detach().float().reshape(1, 3).expand(o.shape[0], 3).contiguous() if qc.is_cuda else qc
            out = renderer.render_rays(o, d, qarg, z_vals=z, white_bkgd=bool(white_bkgd), want_knn=True)
            ctx.save_for_backward(o, d, qc, z, out["knn_d2"], out["knn_idx"])
        ctx.mark_non_differentiable(out["mask"])
        return out["rgb"], out["depth"], out["depth_uncertainty"], out["feat"], out["weights"], out["mask"]

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_unc, g_feat, g_wts, _g_mask, g_beta=None):
        _same_state(ctx)
        o, d, qc, z = ctx.saved_tensors[:4]
        need = ctx.needs_input_grad
        names = [n for n, nd in zip(RENDER_PARAMS, need[9:]) if nd]
        train = bool(names) or any(need[6:9])
        tg = None
        if train:
            # rgb_blending_mlp.0's feature columns act on taps of the projected maps: their gradient and the maps' share of it come back as a map
            tg = ctx.r.train_grads(names, support_feature=need[8], feat_maps=need[6], vis_featmaps=need[7],
                                   blend_feat_maps=need[6] or "rgb_blending_mlp.0.weight" in names)
        gbeta_w = gbeta_b = None
        if getattr(ctx, "graph", None) is not None:   # (between this node's forward and backward the graphs' static buffers were not touched: one step at a time)
            if ctx.graph.fwd_count != ctx.graph_stamp:
                raise RuntimeError("RenderFn: the graphed keep buffers were overwritten by a later forward before this node's backward ran")
            go, gd, gq = ctx.graph.backward(g_rgb, g_depth, g_unc, g_feat, g_wts)
            ctx.graph = ctx.lease = None
            return (go if need[0] else None, gd if need[1] else None, gq.to(qc.dtype) if need[2] else None, None, None, None, None, None, None)
        if ctx.kept is not None and ctx.has_beta:
            go, gd, gq, (gbeta_w, gbeta_b) = ctx.r.render_rays_backward_kept(ctx.kept, g_rgb, g_depth, g_unc, g_feat, g_wts, want_g_query_center=need[2], train=tg,
                                                                            g_beta=g_beta, want_beta_grads=True)
            ctx.kept = None
        elif ctx.kept is not None:
            go, gd, gq = ctx.r.render_rays_backward_kept(ctx.kept, g_rgb, g_depth, g_unc, g_feat, g_wts, want_g_query_center=need[2], train=tg)
            ctx.kept = None
        else:
            # the recompute path — also what a SECOND backward of this node takes (retain_graph=True, autograd.grad twice): the graph / kept branches above
            # consume their activations and run once; after that the forward is recomputed from the saved rays (the library searches the neighbours again
            # when the node saved none).  The uncertainty head exists only in the keep / kept pair: no second pass for it.
            if ctx.has_beta:
                raise RuntimeError("RenderFn: the backward of a training node with the uncertainty head (beta) can run once (its kept activations were consumed)")
            saved = ctx.saved_tensors
            go, gd, gq = ctx.r.render_rays_backward(o, d, z, qc, g_rgb, g_depth, g_unc, g_feat, g_wts, white_bkgd=ctx.white, want_g_query_center=need[2], train=tg,
                                                    knn=(saved[4], saved[5]) if len(saved) >= 6 else None)
        gmaps = gvis = gsp = None
        gw = {}
        if tg is not None:
            gw = dict(tg.weights)
            gmaps, gvis, gsp = tg.feat_maps, tg.vis_featmaps, tg.support_feature
            if tg.blend_feat_maps is not None:   # P = maps . W0f^T  ->  d W0f = P_grad^T maps, d maps += P_grad . W0f
                fm, w0 = ctx.r._frame_keep[1], ctx.r.weight_tensor("rgb_blending_mlp.0.weight")
                W, Cf = ctx.r.W, fm.shape[-1]
                pg = tg.blend_feat_maps.reshape(-1, 32)
                if "rgb_blending_mlp.0.weight" in gw:
                    gw["rgb_blending_mlp.0.weight"][:, W + 3:W + 3 + Cf] += pg.t() @ fm.reshape(-1, Cf)
                if gmaps is not None:
                    gmaps += (pg @ w0[:, W + 3:W + 3 + Cf]).view_as(gmaps)
        return (go if need[0] else None, gd if need[1] else None, None if gq is None else gq.to(qc.dtype), None, None, None, gmaps, gvis, gsp) + \
            tuple(gw.get(n) for n in RENDER_PARAMS) + ((gbeta_w, gbeta_b) if ctx.has_beta else ())


POINT_PARAMS = ("ray_diff_fc.0.weight", "ray_diff_fc.0.bias", "ray_diff_fc.2.weight", "ray_diff_fc.2.bias",
                "base_mlp.0.weight", "base_mlp.0.bias", "base_mlp.2.weight", "base_mlp.2.bias", "base_mlp.4.weight", "base_mlp.4.bias",
                "base_mlp_attn.w_qs.weight", "base_mlp_attn.w_ks.weight", "base_mlp_attn.w_vs.weight", "base_mlp_attn.fc.weight",
                "base_mlp_attn.layer_norm.weight", "base_mlp_attn.layer_norm.bias")


class PointBranchTrainFn(torch.autograd.Function):
    """PointBranchFn for a TRAINING step (compute_render_loss, model.py:641-685): the same HIP forward, and a backward that also delivers the
    gradients of the branch's 16 parameter tensors and of the support table's features (nl_point_mlp_backward_train: weight gradients as
    split-K products over the recomputed activations, in the same pass as the input gradients).  `renderer` holds the CURRENT values of
    `params` (POINT_PARAMS order) and of the support table; the tensors are passed so that autograd knows where the gradients go.
    base_mlp_agg_weight and the neighbours' confidences get no gradient: it is identically zero (the softmax / the normalisation run over K
    identical rows, model.py:415-427)."""

    @staticmethod
    def forward(ctx, xyz, dirs, G, sp_feature, renderer, K, *params):
        xyz, G = xyz.contiguous(), G.contiguous()
        dirs = None if dirs is None else dirs.contiguous()
        ctx.r, ctx.K, ctx.has_dir, ctx.gen = renderer, int(K), dirs is not None, renderer.state_gen
        fa, d2, idx = renderer.point_mlp(xyz, dirs, G, K=int(K))
        ctx.save_for_backward(xyz, G, d2, idx, *([dirs] if dirs is not None else []))
        return fa

    @staticmethod
    def backward(ctx, g_fa):
        _same_state(ctx)
        xyz, G, d2, idx = ctx.saved_tensors[:4]
        dirs = ctx.saved_tensors[4] if ctx.has_dir else None
        names = [n for n, need in zip(POINT_PARAMS, ctx.needs_input_grad[6:]) if need]
        tg = ctx.r.train_grads(names, support_feature=ctx.needs_input_grad[3])
        gx, gd, gg = ctx.r.point_mlp_backward(xyz, dirs, G, g_fa.contiguous(), K=ctx.K, knn=(d2, idx), train=tg)
        return (gx, gd, gg, tg.support_feature, None, None) + tuple(tg.weights.get(n) for n in POINT_PARAMS)


_DEC = [f"multiview_aggregator.dist_decoder.{d}_decoder.{i}.{t}" for d in ("mean", "var", "aw", "vis") for i in (0, 2, 4) for t in ("weight", "bias")]
MV_PARAMS = tuple(f"multiview_aggregator.out_fc.{i}.{t}" for i in (0, 2) for t in ("weight", "bias")) + tuple(_DEC)
BLEND_PARAMS = tuple(f"rgb_blending_mlp.{i}.{t}" for i in (0, 2, 4) for t in ("weight", "bias")) + tuple(_DEC)
HEAD_PARAMS = ("sigma_mlp.0.weight", "sigma_mlp.0.bias", "feat_mlp.0.weight", "feat_mlp.0.bias", "feat_mlp.2.weight", "feat_mlp.2.bias")


UNET_PARAMS = tuple(f"ray_unet.{blk}.{i}.{t}" for blk in ("conv1", "conv2", "conv3", "trans_conv3", "trans_conv2", "trans_conv1", "conv_out")
                    for i in (0, 1) for t in ("weight", "bias"))


class UnetTrainFn(torch.autograd.Function):
    """UnetFn for a training step: (x (R*S, W), renderer, *UNET_PARAMS) -> geo; backward = nl_ray_unet_backward_train: also the gradients of the seven
    blocks' convolutions (one split-K product per tap) and LayerNorm([C, L]) tables (sums over the rays)."""

    @staticmethod
    def forward(ctx, x, renderer, *params):
        x = x.contiguous()
        ctx.r, ctx.gen = renderer, renderer.state_gen
        ctx.save_for_backward(x)
        return renderer.ray_unet(x)

    @staticmethod
    def backward(ctx, g_geo):
        _same_state(ctx)
        x, = ctx.saved_tensors
        names = [n for n, need in zip(UNET_PARAMS, ctx.needs_input_grad[2:]) if need]
        tg = ctx.r.train_grads(names)
        gx = ctx.r.ray_unet_backward(x, g_geo.contiguous(), train=tg)
        return (gx, None) + tuple(tg.weights.get(n) for n in UNET_PARAMS)


class MvAggTrainFn(torch.autograd.Function):
    """MvAggFn for a training step: (xyz, feature maps (V,h,w,C), DepthFusionNet maps (V,32,vh,vw), renderer, *MV_PARAMS tensors) -> (G, valid_s);
    backward = nl_mv_aggregate_backward_train: also d/d out_fc, d/d the four decoders, d/d both maps."""

    @staticmethod
    def forward(ctx, xyz, feat_maps, vis_maps, renderer, *params):
        xyz = xyz.contiguous()
        ctx.r, ctx.gen = renderer, renderer.state_gen
        ctx.save_for_backward(xyz)
        mv, _, _, valid = renderer.mv_aggregate(xyz, torch.zeros(3), want_raw=False)
        ctx.mark_non_differentiable(valid)
        return mv, valid

    @staticmethod
    def backward(ctx, g_mv, _g_valid):
        _same_state(ctx)
        xyz, = ctx.saved_tensors
        names = [n for n, need in zip(MV_PARAMS, ctx.needs_input_grad[4:]) if need]
        tg = ctx.r.train_grads(names, feat_maps=ctx.needs_input_grad[1], vis_featmaps=ctx.needs_input_grad[2])
        gx = ctx.r.mv_aggregate_backward(xyz, g_mv.contiguous(), train=tg)
        return (gx if ctx.needs_input_grad[0] else None, tg.feat_maps, tg.vis_featmaps, None) + tuple(tg.weights.get(n) for n in MV_PARAMS)


class BlendTrainFn(torch.autograd.Function):
    """BlendFn for a training step: (xyz, feature_agg, query centre, blend-projected maps P (V,h,w,32) = feature maps . the feature columns of
    rgb_blending_mlp.0 [its graph carries those columns' and the maps' gradient], DepthFusionNet maps, renderer, *BLEND_PARAMS) -> rgb_s (N,3);
    backward = nl_blend_backward_train."""

    @staticmethod
    def forward(ctx, xyz, fa, qc, pmaps, vis_maps, renderer, *params):
        xyz, fa = xyz.contiguous(), fa.contiguous()
        ctx.r, ctx.gen = renderer, renderer.state_gen
        ctx.save_for_backward(xyz, fa, qc)
        return renderer.blend(xyz, qc, fa)

    @staticmethod
    def backward(ctx, g_rgb_s):
        _same_state(ctx)
        xyz, fa, qc = ctx.saved_tensors
        names = [n for n, need in zip(BLEND_PARAMS, ctx.needs_input_grad[6:]) if need]
        tg = ctx.r.train_grads(names, vis_featmaps=ctx.needs_input_grad[4], blend_feat_maps=ctx.needs_input_grad[3])
        gx, gfa, gq = ctx.r.blend_backward(xyz, qc, fa, g_rgb_s.contiguous(), want_g_query_center=ctx.needs_input_grad[2], train=tg)
        return (gx if ctx.needs_input_grad[0] else None, gfa, None if gq is None else gq.to(qc.dtype), tg.blend_feat_maps, tg.vis_featmaps, None) + \
            tuple(tg.weights.get(n) for n in BLEND_PARAMS)


RENDER_PARAMS = POINT_PARAMS + MV_PARAMS + BLEND_PARAMS[:6] + UNET_PARAMS + HEAD_PARAMS   # the 84 tensors of the ray path


class MvAggFn(torch.autograd.Function):
    """Multi-view aggregation (multiview_aggregator.py:156-222) with frozen weights / support maps as one autograd node on the HIP library:
    xyz (N,3) -> (G (N,W), valid_s (N) int32 [non-differentiable: #views that see the sample > 1]); backward = nl_mv_aggregate_backward."""

    @staticmethod
    def forward(ctx, xyz, renderer):
        xyz = xyz.contiguous()
        ctx.r, ctx.gen = renderer, renderer.state_gen
        ctx.save_for_backward(xyz)
        mv, _, _, valid = renderer.mv_aggregate(xyz, torch.zeros(3), want_raw=False)
        ctx.mark_non_differentiable(valid)
        return mv, valid

    @staticmethod
    def backward(ctx, g_mv, _g_valid):
        _same_state(ctx)
        xyz, = ctx.saved_tensors
        return ctx.r.mv_aggregate_backward(xyz, g_mv.contiguous()), None


class UnetFn(torch.autograd.Function):
    """The ray U-Net (ray_unet.py:55-69) with frozen weights on the HIP library: x (R*S, W) sample-major -> geo (R*S, W); backward = nl_ray_unet_backward."""

    @staticmethod
    def forward(ctx, x, renderer):
        x = x.contiguous()
        ctx.r, ctx.gen = renderer, renderer.state_gen
        ctx.save_for_backward(x)
        return renderer.ray_unet(x)

    @staticmethod
    def backward(ctx, g_geo):
        _same_state(ctx)
        x, = ctx.saved_tensors
        return ctx.r.ray_unet_backward(x, g_geo.contiguous()), None


class BlendFn(torch.autograd.Function):
    """Per-sample colours (model.py:528-538) with frozen weights / maps: (xyz (N,3), feature_agg (N,W), query camera centre (3,)) -> rgb_s (N,3);
    forward = nl_blend, backward = nl_blend_backward."""

    @staticmethod
    def forward(ctx, xyz, fa, qc, renderer):
        xyz, fa = xyz.contiguous(), fa.contiguous()
        ctx.r, ctx.gen = renderer, renderer.state_gen
        ctx.save_for_backward(xyz, fa, qc)
        return renderer.blend(xyz, qc, fa)

    @staticmethod
    def backward(ctx, g_rgb_s):
        _same_state(ctx)
        xyz, fa, qc = ctx.saved_tensors
        gx, gfa, gq = ctx.r.blend_backward(xyz, qc, fa, g_rgb_s.contiguous(), want_g_query_center=ctx.needs_input_grad[2])
        return gx, gfa, (None if gq is None else gq.to(qc.dtype)), None


def rays_from_pose(uv: Tensor, K: Tensor, pose: Tensor):
    """conditional_nerf/utils.py:56-70 + model.py:687-700 for the selected pixels only (integer-truncated pixel coordinates):
    unit directions rotated by the camera-to-world pose, origin = its translation.  Differentiable w.r.t. `pose`."""
    x, y = uv[:, 0].long().to(pose.dtype), uv[:, 1].long().to(pose.dtype)
    cam = torch.stack([(x - K[0, 2]) / K[0, 0], (y - K[1, 2]) / K[1, 1], torch.ones_like(x)], -1)
    d = (cam[:, None, :] * pose[:3, :3]).sum(-1)
    d = d / torch.norm(d, dim=-1, keepdim=True)
    return pose[:3, 3].expand(d.shape), d


def _lrelu(x):
    return F.leaky_relu(x, 0.01)


def _lin(p, name, x, bias=True):
    return F.linear(x, p[f"{name}.weight"], p[f"{name}.bias"] if bias else None)


def _mlp3(p, pre, x):
    return _lin(p, f"{pre}.4", F.elu(_lin(p, f"{pre}.2", F.elu(_lin(p, f"{pre}.0", x)))))


def _posenc(x: Tensor, n_freqs: int = 10) -> Tensor:
    """conditional_nerf/utils.py:5-35."""
    out = [x]
    for i in range(n_freqs):
        out += [torch.sin(x * 2.0 ** i), torch.cos(x * 2.0 ** i)]
    return torch.cat(out, -1)


def _mv_aggregate(p, fr, xyz: Tensor):
    """multiview_aggregator.py:156-222 (+ ibrnet.py:169-231, depth_fusion.py:60-147, visibility_decoder.py:64-148):
    per-view bilinear taps, NeuRay visibility, visibility-weighted mean / variance, out_fc.
    -> (G (N, W), rgb_feat (N, V, 3 + C), vis (N, V, 1), mask1 (N, V))"""
    Ks, poses, images = fr["topk_Ks"], fr["topk_poses"], fr["topk_images"]
    featmaps = fr["feat_fine_src"].permute(0, 3, 1, 2)
    V, (H, W) = images.shape[0], images.shape[-2:]
    dev, dt = xyz.device, xyz.dtype
    w2c = torch.inverse(poses)
    xyz_h = torch.cat([xyz, torch.ones_like(xyz[:, :1])], -1)
    # Projector (ibrnet.py:169-231): K4 . w2c, pixel clamp +-1e6, in front, closed in-image interval; both maps are addressed with
    # coordinates normalised by the full-resolution (W - 1, H - 1), zeros padding, align_corners = True
    K4 = torch.eye(4, device=dev, dtype=dt).repeat(V, 1, 1)
    K4[:, :3, :3] = Ks
    proj = torch.einsum("vij,nj->vni", K4.bmm(w2c), xyz_h)
    pix = torch.clamp(proj[..., :2] / torch.clamp(proj[..., 2:3], min=1e-8), min=-1e6, max=1e6)
    grid = (2 * pix / torch.tensor([W - 1.0, H - 1.0], device=dev, dtype=dt) - 1.0).unsqueeze(2)
    rgb = F.grid_sample(images, grid, align_corners=True).squeeze(-1).permute(2, 0, 1)
    feat = F.grid_sample(featmaps, grid, align_corners=True).squeeze(-1).permute(2, 0, 1)
    inb = (pix[..., 0] <= W - 1.0) & (pix[..., 0] >= 0) & (pix[..., 1] <= H - 1.0) & (pix[..., 1] >= 0)
    mask1 = (inb & (proj[..., 2] > 0)).permute(1, 0)
    rgb_feat = torch.cat([rgb, feat], -1)
    # NeuRay-convention projection for the visibility features (depth_fusion.py:78-147): |z| < 1e-4 -> 1e-3 (invalid), half-open image test
    cam = torch.einsum("vij,nj->vni", Ks.bmm(w2c[:, :3]), xyz_h)
    depth = cam[..., 2:]
    bad = depth.abs() < 1e-4
    depth = torch.where(bad, torch.full_like(depth, 1e-3), depth)
    pv = cam[..., :2] / depth
    outside = (pv[..., 0] < -0.5) | (pv[..., 0] >= W - 0.5) | (pv[..., 1] < -0.5) | (pv[..., 1] >= H - 0.5)
    valid = ((~bad[..., 0]) & (~outside)).to(dt).unsqueeze(-1)
    vf = fr["vis_featmaps"]
    g2 = torch.stack([pv[..., 0] / (W - 1) * 2 - 1, pv[..., 1] / (H - 1) * 2 - 1], -1).unsqueeze(1)
    rf = F.grid_sample(vf, g2, mode="bilinear", padding_mode="border", align_corners=(vf.shape[-2] == H and vf.shape[-1] == W))
    rf = rf.squeeze(2).permute(0, 2, 1) * valid
    # mixture-of-logistics decoders (visibility_decoder.py:64-138)
    pre = "multiview_aggregator.dist_decoder"
    mean = F.softplus(_mlp3(p, f"{pre}.mean_decoder", rf))
    var = F.softplus(_mlp3(p, f"{pre}.var_decoder", rf)) + 0.05
    aw = torch.sigmoid(_mlp3(p, f"{pre}.aw_decoder", rf))
    vis0 = torch.sigmoid(_mlp3(p, f"{pre}.vis_decoder", rf))
    near, far = fr["near"], fr["far"]
    ni, fi = -1.0 / near, -1.0 / far
    ref_d = (-1.0 / (mean[..., 0] * (fi - ni) + ni)).clamp(near, far)
    ddiff = (depth[..., 0] - ref_d).abs() / (far - near)
    dn = (-1.0 / torch.clamp(depth, min=1e-5) - ni) / (fi - ni)
    cdf = (0.5 + 0.5 * torch.tanh((dn - mean) * var)) * vis0
    vis = (torch.sum((1 - cdf) * torch.cat([aw, 1 - aw], -1), -1, keepdim=True) * valid).permute(1, 0, 2)     # (N, V, 1)
    ddiff = ddiff.unsqueeze(-1).permute(1, 0, 2)
    wgt = vis / (torch.sum(vis, dim=1, keepdim=True) + 1e-8)

    def mean_var(x):   # ibrnet.py:8-12
        m = torch.sum(x * wgt, dim=1, keepdim=True)
        return m, torch.sum(wgt * (x - m) ** 2, dim=1, keepdim=True)

    m1, v1 = mean_var(rgb_feat)
    m2, v2 = mean_var(ddiff)
    g = torch.cat([torch.cat([m1, v1, m2, v2], -1).squeeze(1), wgt.mean(dim=1)], -1)
    G = F.elu(_lin(p, "multiview_aggregator.out_fc.2", F.elu(_lin(p, "multiview_aggregator.out_fc.0", g))))
    return G, rgb_feat, vis, mask1


def _point_branch(p, fr, xyz: Tensor, dirs: Optional[Tensor], G: Tensor, idx: Tensor, full: bool = False):
    """conditional_nerf/model.py:344-436 + ibrnet.py:89-119: K nearest neural points per sample (indices given), relative-position /
    ray-difference encoding, 3-layer MLP, 4-head attention with the multi-view feature as query, distance x confidence x learnt weights.
    dirs None: the nearest neighbour's viewing direction (model.py:391-392, the descriptor queries).  Fewer support points than K: the
    missing neighbours are zero rows at squared distance 0 (knn_utils.py:48-53, 211-220) — they take part in the attention and get
    weight 0 through their zero confidence.  full: also return the per-neighbour features (N, K, W) and weights (N, K)."""
    sp = fr["support"]
    K, M = idx.shape[1], sp["xyz"].shape[0]
    keep = None if M >= K else (torch.arange(K, device=idx.device) < M)

    def take(t):
        g = t[idx]
        return g if keep is None else g * keep.view(1, K, 1).to(g.dtype)
    nb_xyz, nb_feat, nb_conf, nb_dir = take(sp["xyz"]), take(sp["feature"]), take(sp["confidence"]), take(sp["direction"])
    if dirs is None:
        dirs = nb_dir[:, 0, :3]
    off = xyz[:, None, :] - nb_xyz
    # = the KNN op's squared distances; its backward (knn.cu:449-490 / knn_cpu.cpp:68-117) is 2 (p1 - p2) grad: the HIP kernel on the GPU,
    # plain autograd of the same expression elsewhere (CPU tests, fp64 checks)
    if keep is None and _hip_ok(xyz, sp["xyz"]):
        d2 = KnnDist2Fn.apply(xyz, sp["xyz"], idx)
    else:
        d2 = (off * off).sum(-1)
    dist = d2.sqrt() if keep is None else torch.where(keep.view(1, K), d2.clamp_min(1e-30).sqrt(), torch.zeros_like(d2))
    rd = dirs[:, None, :] - nb_dir[..., :3]
    rd = rd / (torch.norm(rd, dim=-1, keepdim=True) + 1e-8)
    rd = torch.cat([rd, torch.sum(dirs[:, None, :] * nb_dir[..., :3], dim=-1, keepdim=True)], -1)
    a = _lrelu(_lin(p, "ray_diff_fc.2", _lrelu(_lin(p, "ray_diff_fc.0", rd))))
    x = torch.cat([nb_feat, _posenc(off / (fr["far"] - fr["near"])), a], -1)
    for i in (0, 2, 4):
        x = _lrelu(_lin(p, f"base_mlp.{i}", x))
    pre, H4, dk = "base_mlp_attn", 4, 32
    N = x.shape[0]
    q = G.unsqueeze(1).expand(-1, K, -1)
    qq = _lin(p, f"{pre}.w_qs", q, False).view(N, K, H4, dk).transpose(1, 2)
    kk = _lin(p, f"{pre}.w_ks", x, False).view(N, K, H4, dk).transpose(1, 2)
    vv = _lin(p, f"{pre}.w_vs", x, False).view(N, K, H4, dk).transpose(1, 2)
    att = F.softmax(torch.matmul(qq / dk ** 0.5, kk.transpose(2, 3)), dim=-1)
    o = torch.matmul(att, vv).transpose(1, 2).reshape(N, K, -1)
    o = _lin(p, f"{pre}.fc", o, False) + q
    feat = F.layer_norm(o, (o.shape[-1],), p[f"{pre}.layer_norm.weight"], p[f"{pre}.layer_norm.bias"], eps=1e-6)
    lg = _lin(p, "base_mlp_agg_weight.2", _lrelu(_lin(p, "base_mlp_agg_weight.0", feat))).squeeze(-1)
    w = 1.0 / torch.clamp(dist, min=1e-8) * F.softmax(lg, dim=1) * nb_conf.squeeze(-1)
    w = w / torch.clamp(w.sum(dim=1, keepdim=True), min=1e-8)
    agg = (feat * w.unsqueeze(-1)).sum(dim=1)
    return (agg, feat, w) if full else agg


def query_diff(p: Dict[str, Tensor], fr: Dict, xyz: Tensor, direction: Optional[Tensor], idx: Tensor) -> Dict[str, Tensor]:
    """ConditionalNeRF.query (conditional_nerf/model.py:344-436) with autograd: the descriptor queries' core.  `fr['feat_fine_src']`
    holds the feature maps of the queried level (the reference passes `feat_coarse_src` for query_coarse, model.py:296-302) and
    `fr['support']` that level's support table; idx (N, K) int64 = the exact KNN of xyz in it (no gradient: indices).  The matcher's
    training signal reaches base_mlp / the attention / the aggregator / the feature maps through 'feature_agg'
    (nerf_pose_estimator.py:316-320, 445-448, 465-468)."""
    G, mvf, mvv, _ = _mv_aggregate(p, fr, xyz)
    agg, feat, w = _point_branch(p, fr, xyz, None if direction is None else direction[:, :3], G, idx, full=True)
    return {"feature_agg": agg, "feature": feat, "weights": w, "multiview_feature": mvf, "multiview_visibility": mvv}


def _conv3(t: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """Conv1d(k = 3, stride 1, padding 1) as ONE matrix product over the three taps: t (R, Ci, L), w (Co, Ci, 3) -> (R, Co, L).  (The
    framework's convolution kernels are slow for these short rays: 6 ms of a 512-ray gradient step; a GEMM is 10x faster and its autograd
    is two more GEMMs.)"""
    R, Ci, L = t.shape
    cols = F.pad(t, (1, 1)).unfold(2, 3, 1)                                    # (R, Ci, L, 3)
    y = cols.permute(0, 2, 1, 3).reshape(R * L, Ci * 3) @ w.reshape(w.shape[0], -1).t()
    return y.view(R, L, -1).permute(0, 2, 1) + b[None, :, None]


def _convT3(t: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """ConvTranspose1d(k = 3, stride 2, padding 1, output_padding 1) by its two output phases: y[2m] = W[:, :, 1]^T x[m],
    y[2m + 1] = W[:, :, 2]^T x[m] + W[:, :, 0]^T x[m + 1];  t (R, Ci, L), w (Ci, Co, 3) -> (R, Co, 2L)."""
    R, Ci, L = t.shape
    xt = t.permute(0, 2, 1)
    xn = F.pad(xt, (0, 0, 0, 1))[:, 1:]
    even = xt.reshape(R * L, Ci) @ w[:, :, 1]
    odd = torch.cat([xt, xn], -1).reshape(R * L, 2 * Ci) @ torch.cat([w[:, :, 2], w[:, :, 0]], 0)
    return torch.stack([even, odd], 1).view(R, L, 2, -1).reshape(R, 2 * L, -1).permute(0, 2, 1) + b[None, :, None]


def _ray_unet(p, x: Tensor) -> Tensor:
    """conditional_nerf/ray_unet.py:5-69 — x (R, W, S) -> (R, W, S)."""
    def block(name, t, transposed=False):
        w, b = p[f"ray_unet.{name}.0.weight"], p[f"ray_unet.{name}.0.bias"]
        t = _convT3(t, w, b) if transposed else _conv3(t, w, b)
        g, be = p[f"ray_unet.{name}.1.weight"], p[f"ray_unet.{name}.1.bias"]
        return F.elu(F.layer_norm(t, tuple(g.shape), g, be, eps=1e-5))

    c1 = F.max_pool1d(block("conv1", x), 2)
    c2 = F.max_pool1d(block("conv2", c1), 2)
    c3 = F.max_pool1d(block("conv3", c2), 2)
    x0 = block("trans_conv3", c3, True)
    x1 = block("trans_conv2", torch.cat([c2, x0], 1), True)
    x2 = block("trans_conv1", torch.cat([c1, x1], 1), True)
    return block("conv_out", torch.cat([x, x2], 1))


def _view_angles(xyz: Tensor, query_center: Tensor, view_centers: Tensor) -> Tensor:
    """ibrnet.py:144-167 — (N, V, 4): unit difference of the unit rays to the query / support cameras, and their dot product."""
    tq = query_center.view(1, 1, 3) - xyz.unsqueeze(1)
    tq = tq / (torch.norm(tq, dim=-1, keepdim=True) + 1e-6)
    tv = view_centers.unsqueeze(0) - xyz.unsqueeze(1)
    tv = tv / (torch.norm(tv, dim=-1, keepdim=True) + 1e-6)
    diff = tq - tv
    return torch.cat([diff / torch.clamp(torch.norm(diff, dim=-1, keepdim=True), min=1e-6), torch.sum(tq * tv, dim=-1, keepdim=True)], -1)


def render_rays_diff(p: Dict[str, Tensor], fr: Dict, rays_o: Tensor, rays_d: Tensor, z_vals: Tensor, query_pose: Tensor,
                     knn_idx: Callable[[Tensor], Tensor], white_bkgd: bool = False, beta: bool = False, frozen_renderer=None,
                     train_renderer=None, whole_path: bool = True) -> Dict[str, Tensor]:
    """conditional_nerf/model.py:472-600 with autograd.  `z_vals` (R, S) are constants (the hierarchical resampling detaches its
    weights, model.py:495); `knn_idx(xyz) -> (N, 8) int64` is the exact KNN (no gradient: indices); beta: the training-mode
    uncertainty output (model.py:587-592).  Gradients reach whatever requires grad among rays_o / rays_d / query_pose, the
    parameters `p` and the frame tensors.  frozen_renderer: a HipRenderer holding exactly `p` and `fr['support']` — states that both are
    constants of this call, so the neural-point branch may run as PointBranchFn (HIP forward + HIP backward) instead of eager ops.
    train_renderer: a HipRenderer holding the CURRENT VALUES of `p` and of the frame tensors (a training step): the stages whose weight
    gradients the library computes (`*TrainFn`) run as HIP nodes that also return d/d parameters and d/d frame tensors; the others stay eager.
    whole_path (with either renderer; no `beta`): the whole function is ONE node (`RenderFn`: fused forward, `nl_render_rays_backward`) instead of one
    node per stage.
    fr: topk_Ks, topk_poses, topk_images, feat_fine_src, vis_featmaps, near, far (python floats), support {xyz, feature, confidence, direction}."""
    R, S = z_vals.shape
    xyz = (rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., None]).reshape(-1, 3)
    dirs = rays_d[:, None, :].expand(R, S, 3).reshape(-1, 3)
    frozen = frozen_renderer is not None and _hip_ok(xyz, dirs)
    hip_train = not frozen and train_renderer is not None and _hip_ok(xyz, dirs) and fr["support"]["xyz"].shape[0] >= 1 and train_renderer.train_capable()
    r = frozen_renderer if frozen else train_renderer
    beta_ok = not beta or (hip_train and KEEP_BYTES and "beta_mlp.0.weight" in p and
                           r.lib.nl_render_rays_keep_workspace_bytes(ct.byref(r.cfg), r.V, R, 1) <= KEEP_BYTES)   # (the uncertainty head lives in the keep / kept pair)
    if (frozen or hip_train) and whole_path and beta_ok and S == r.S and not z_vals.requires_grad and fr["support"]["xyz"].shape[0] >= 1 \
            and all(n in p for n in HEAD_PARAMS):
        # ONE autograd node for the whole path: the fused inference kernels forward, nl_render_rays_backward backward
        extra = () if frozen else (fr["feat_fine_src"], fr["vis_featmaps"], fr["support"]["feature"]) + tuple(p[n] for n in RENDER_PARAMS)
        if beta:
            extra = extra + (p["beta_mlp.0.weight"], p["beta_mlp.0.bias"])
        res = RenderFn.apply(rays_o, rays_d, query_pose[:3, 3], z_vals, r, white_bkgd, *extra)
        rgb, depth, unc, feat, wts, valid = res[:6]
        out = {"rgb": rgb, "feat": feat, "depth": depth, "weights": wts, "mask": valid, "depth_uncertainty": unc}
        if beta:
            out["beta"] = res[6]
        return out
    if frozen:
        # frozen weights + frozen per-frame tables (pose refinement): aggregation, neural-point branch and blend are three autograd nodes whose
        # forward AND backward run in the HIP library; nothing of them is kept on the tape but their inputs
        G, valid_s = MvAggFn.apply(xyz, r)
        agg = PointBranchFn.apply(xyz, dirs.contiguous(), G, r, 8)
    elif hip_train:
        # a training step: the same three nodes, whose backward also returns the gradients of their parameters and of the per-frame tensors
        G, valid_s = MvAggTrainFn.apply(xyz, fr["feat_fine_src"], fr["vis_featmaps"], r, *[p[n] for n in MV_PARAMS])
        agg = PointBranchTrainFn.apply(xyz, dirs.contiguous(), G, fr["support"]["feature"], r, 8, *[p[n] for n in POINT_PARAMS])
    else:
        G, mvf, mvv, mask1 = _mv_aggregate(p, fr, xyz)
        with torch.no_grad():
            idx = knn_idx(xyz.detach()).long()
        agg = _point_branch(p, fr, xyz, dirs, G, idx)
    W = agg.shape[1]
    if frozen and S == r.S:
        geo = UnetFn.apply(agg, r)
    elif hip_train and S == r.S:
        geo = UnetTrainFn.apply(agg, r, *[p[n] for n in UNET_PARAMS])
    else:
        geo = _ray_unet(p, agg.view(R, S, W).permute(0, 2, 1)).permute(0, 2, 1).reshape(R * S, W)
    sigma = F.softplus(_lin(p, "sigma_mlp.0", geo)).view(R, S)
    if frozen:
        rgb_s = BlendFn.apply(xyz, agg, query_pose[:3, 3], r).view(R, S, 3)
    elif hip_train:
        # the feature columns of rgb_blending_mlp.0 act on bilinear taps of the feature maps = taps of the projected maps (linearity): the
        # projection is one small product per frame here, its graph carries those columns' and the maps' share of the gradient
        Cf = fr["feat_fine_src"].shape[-1]
        pmaps = F.linear(fr["feat_fine_src"], p["rgb_blending_mlp.0.weight"][:, W + 3:W + 3 + Cf])
        rgb_s = BlendTrainFn.apply(xyz, agg, query_pose[:3, 3], pmaps, fr["vis_featmaps"], r, *[p[n] for n in BLEND_PARAMS]).view(R, S, 3)
    else:
        V = mvf.shape[1]
        ang = _view_angles(xyz, query_pose[:3, 3], fr["topk_poses"][:, :3, 3])
        # rgb_blending_mlp.0 on cat[feature_agg (repeated over the views), multi-view feature, visibility, view angles] (model.py:532-535), evaluated
        # by linearity as four partial products: the (N, V, W + C + 8) concatenation (1.2 GB per 512-ray batch) is never materialised
        w0, b0 = p["rgb_blending_mlp.0.weight"], p["rgb_blending_mlp.0.bias"]
        Fd = mvf.shape[-1]
        xb = (F.linear(agg, w0[:, :W]) + b0).unsqueeze(1) + F.linear(mvf, w0[:, W:W + Fd]) + mvv * w0[:, W + Fd] + F.linear(ang, w0[:, W + Fd + 1:])
        xb = _lrelu(_lin(p, "rgb_blending_mlp.2", _lrelu(xb)))
        bw = F.softmax(_lin(p, "rgb_blending_mlp.4", xb).masked_fill(mvv == 0, -1e9), dim=1)
        rgb_s = torch.sum(mvf[:, :, :3] * bw, dim=1).view(R, S, 3)
    # front-to-back compositing (model.py:544-560, 597): its backward is the HIP kernel nl_composite_backward on the GPU
    ft = _lin(p, "feat_mlp.2", _lrelu(_lin(p, "feat_mlp.0", agg))).view(R, S, -1)
    if _hip_ok(sigma, rgb_s, ft, z_vals) and not z_vals.requires_grad and S <= 256:
        rgb, depth, unc, feat, wts = CompositeFn.apply(sigma, rgb_s, ft, z_vals, white_bkgd)
    else:
        rgb, depth, unc, feat, wts = composite_eager(sigma, rgb_s, ft, z_vals, white_bkgd)
    valid = (valid_s.view(R, S) > 0).float().sum(1) > 8 if (frozen or hip_train) else (mask1.view(R, S, -1).sum(2) > 1).float().sum(1) > 8
    out = {"rgb": rgb, "feat": feat, "depth": depth, "weights": wts, "mask": valid, "depth_uncertainty": unc}
    if beta:
        out["beta"] = (wts * F.softplus(_lin(p, "beta_mlp.0", geo)).view(R, S)).sum(1) + 0.1   # beta_min, model.py:98
    return out


# ----------------------------------------------------------------------------- hierarchical branch with its graph (a20)
def coarse_weights_diff(p: Dict[str, Tensor], fr: Dict, pix: Tensor, K: Tensor, pose: Tensor, zc: Tensor) -> Tensor:
    """multiview_aggregator.py:95-154 (+ depth_fusion.py:9-58, visibility_decoder.py:6-51,150-181): hit-probability weights (R, dn) of the
    coarse depths `zc` along the query rays through pixels `pix`, predicted from the support views' visibility features."""
    Ks, poses = fr["topk_Ks"], fr["topk_poses"]
    V = Ks.shape[0]
    H, W = fr["topk_images"].shape[-2:]
    near, far = fr["near"], fr["far"]
    ni, fi = -1.0 / near, -1.0 / far
    dev, dt = zc.device, zc.dtype
    rn, dn = zc.shape
    # query rays: centre + un-normalised K^-1 [u, v, 1] directions in the world frame (depth_fusion.py:9-45)
    q_w2c = torch.inverse(pose)[:3]
    rot = q_w2c[:, :3].t()
    cen = -rot @ q_w2c[:, 3]
    cam = torch.inverse(K) @ torch.cat([pix, torch.ones(rn, 1, device=dev, dtype=dt)], 1).t()      # (3, rn)
    dirs = (rot @ cam).t()
    pts = (cen[None, None, :] + dirs[:, None, :] * zc[..., None]).reshape(-1, 3)
    # interval lengths in normalised inverse depth, last = 1e6 (depth_fusion.py:47-58)
    di = (-1.0 / zc - ni) / (fi - ni)
    dist = torch.cat([di[:, 1:] - di[:, :-1], torch.full((rn, 1), 1e6, device=dev, dtype=dt)], -1)
    # projection into the support views + visibility features (as in _mv_aggregate)
    w2c = torch.inverse(poses)
    xyz_h = torch.cat([pts, torch.ones_like(pts[:, :1])], -1)
    camv = torch.einsum("vij,nj->vni", Ks.bmm(w2c[:, :3]), xyz_h)
    depth = camv[..., 2:]
    bad = depth.abs() < 1e-4
    depth = torch.where(bad, torch.full_like(depth, 1e-3), depth)
    pv = camv[..., :2] / depth
    outside = (pv[..., 0] < -0.5) | (pv[..., 0] >= W - 0.5) | (pv[..., 1] < -0.5) | (pv[..., 1] >= H - 0.5)
    mask = ((~bad[..., 0]) & (~outside)).to(dt).unsqueeze(-1)
    vf = fr["vis_featmaps"]
    g2 = torch.stack([pv[..., 0] / (W - 1) * 2 - 1, pv[..., 1] / (H - 1) * 2 - 1], -1).unsqueeze(1)
    rf = F.grid_sample(vf, g2, mode="bilinear", padding_mode="border", align_corners=(vf.shape[-2] == H and vf.shape[-1] == W))
    rf = rf.squeeze(2).permute(0, 2, 1) * mask
    pre = "multiview_aggregator.dist_decoder"
    mean = F.softplus(_mlp3(p, f"{pre}.mean_decoder", rf)).view(V, rn, dn, -1)
    var = (F.softplus(_mlp3(p, f"{pre}.var_decoder", rf)) + 0.05).view(V, rn, dn, -1)
    aw = torch.sigmoid(_mlp3(p, f"{pre}.aw_decoder", rf)).view(V, rn, dn, -1)
    vis0 = torch.sigmoid(_mlp3(p, f"{pre}.vis_decoder", rf)).view(V, rn, dn, -1)
    # probability of a hit inside [d - half interval, d + half interval] (visibility_decoder.py:6-51, 150-181; is_ref = True)
    d = ((-1.0 / torch.clamp(depth.view(V, rn, dn), min=1e-5) - ni) / (fi - ni))
    half = dist[None] / 2
    ext = torch.cat([half[..., 0:1], half], -1)
    lo, hi = (d - ext[..., :-1]).unsqueeze(-1), (d + ext[..., 1:]).unsqueeze(-1)
    mix = torch.cat([aw, 1 - aw], -1)
    c0 = (0.5 + 0.5 * torch.tanh((lo - mean) * var)) * vis0
    c1 = (0.5 + 0.5 * torch.tanh((hi - mean) * var)) * vis0
    visib = torch.sum((1 - c0) * mix, -1)
    hit = torch.sum((c1 - c0) * mix, -1)
    alpha = torch.log(hit / (visib - hit + 1e-5) + 1e-5)
    ground = -15.0
    m = mask.view(V, rn, dn)
    a = alpha * m + (1 - m) * ground
    vs = visib * m
    a = (a * vs).sum(0) / torch.clip(vs.sum(0), min=1e-8)
    none = (m.sum(0) == 0).to(dt)
    a = torch.sigmoid(a * (1 - none) + none * ground)
    T = torch.cumprod(torch.cat([torch.ones_like(a[:, :1]), 1 - a], -1)[:, :-1], -1)
    return a * T


def sample_pdf_diff(bins: Tensor, weights: Tensor, u: Tensor, eps: float = 1e-5) -> Tensor:
    """conditional_nerf/utils.py:73-112: inverse-CDF samples of the piecewise-constant pdf `weights` over `bins` at the uniform draws u."""
    n_bins = weights.shape[1]
    w = weights + eps
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    inds = torch.searchsorted(cdf, u.contiguous(), right=True)
    lo, hi = torch.clamp_min(inds - 1, 0), torch.clamp_max(inds, n_bins)
    c_lo, c_hi = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
    b_lo, b_hi = torch.gather(bins, 1, lo), torch.gather(bins, 1, hi)
    den = c_hi - c_lo
    den = torch.where(den < eps, torch.ones_like(den), den)
    return b_lo + (u - c_lo) / den * (b_hi - b_lo)


# ----------------------------------------------------------------------------- training step (model.py:641-685)
def backproject_support_diff(imgs: Tensor, feats: Tensor, depths: Tensor, Ks: Tensor, c2ws: Tensor, stride: int):
    """conditional_nerf/model.py:203-265 in autograd ops: every valid-depth pixel of every (nearest-resized) support view becomes a
    neural point.  -> (rgb + feature (M, 3 + C), xyz world, xyz in view 0's camera, unit direction + depth), rows in the reference's
    order (view, then nonzero()'s row-major pixels).  The gradient reaches `feats` (the 2D backbone's maps)."""
    desc, world, ref, dirs = [], [], [], []
    w2c0 = torch.inverse(c2ws[0])
    for img, feat, depth, K, c2w in zip(imgs, feats, depths, Ks, c2ws):
        H, W = int(img.shape[-2] / stride), int(img.shape[-1] / stride)
        K = K.clone()
        K[:2] = K[:2] / stride
        depth = F.interpolate(depth[None, None], size=(H, W)).squeeze()
        img = F.interpolate(img[None], size=(H, W)).squeeze().permute(1, 2, 0)
        v, u = torch.nonzero(depth > 0, as_tuple=True)
        z = depth[v, u]
        cam = torch.matmul(torch.inverse(K), torch.stack([u, v, torch.ones_like(u)], 0).to(z.dtype)) * z
        world.append((torch.matmul(c2w[:3, :3], cam) + c2w[:3, 3:]).t())
        ref.append(torch.matmul(torch.matmul(w2c0, c2w), torch.cat([cam, torch.ones_like(cam[:1])]))[:3].t())
        ray = torch.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], torch.ones_like(z)], -1)   # utils.py:56-70 at (u, v)
        ray = (ray[:, None, :] * c2w[:3, :3]).sum(-1)
        dirs.append(torch.cat([ray / torch.norm(ray, dim=-1, keepdim=True), z.view(-1, 1)], 1))
        desc.append(torch.cat([img[v, u], feat[v, u]], 1))
    return torch.cat(desc), torch.cat(world), torch.cat(ref), torch.cat(dirs)


def support_tables_diff(p: Dict[str, Tensor], fr: Dict, depths: Tensor, stride: int) -> Dict[str, Tensor]:
    """The fine-level support table with its graph (model.py:144-201, 137-142): features gathered from the feature maps, confidence =
    confidence_mlp(multi-view aggregate at the points).  fr as for render_rays_diff, without 'support'."""
    desc, xyz, ref, dirs = backproject_support_diff(fr["topk_images"], fr["feat_fine_src"], depths, fr["topk_Ks"], fr["topk_poses"], stride)
    G = _mv_aggregate(p, fr, xyz)[0]
    conf = torch.sigmoid(_lin(p, "confidence_mlp.2", _lrelu(_lin(p, "confidence_mlp.0", G))))
    return {"xyz": xyz, "xyz_ndc": ref, "feature": desc, "confidence": conf, "direction": dirs}


def coarse_support_diff(p: Dict[str, Tensor], imgs: Tensor, feats_coarse: Tensor, depths: Tensor, Ks: Tensor, c2ws: Tensor, stride: int) -> Dict[str, Tensor]:
    """The coarse-level support table with its graph (model.py:144-201): features gathered from the coarse feature maps, confidence 1,
    keypoint_score = keypoint_head(feature) (what `sample_points_3d` draws from)."""
    desc, xyz, ref, dirs = backproject_support_diff(imgs, feats_coarse, depths, Ks, c2ws, stride)
    return {"xyz": xyz, "xyz_ndc": ref, "feature": desc, "confidence": torch.ones_like(xyz[:, :1]), "direction": dirs,
            "keypoint_score": torch.sigmoid(_lin(p, "keypoint_head.0", desc[:, 3:]))}


def to_inverse_normalized_depth(depth: Tensor, near, far) -> Tensor:
    """conditional_nerf/losses.py:15-21."""
    ni, fi = -1 / near, -1 / far
    return torch.clamp((-1 / torch.clamp(depth, min=1e-5) - ni) / (fi - ni), min=0, max=1.0)


def rendering_loss(pred: Dict[str, Tensor], tgt: Dict, use_depth: bool = False, coef: float = 1.0) -> Tensor:
    """conditional_nerf/losses.py:23-93 (RenderingLoss, NeRF-W eq. 13): beta-weighted colour loss + log-beta, optional inverse-depth
    terms, 0.1 x feature MSE — over the rays of the mask."""
    mask = tgt["mask"].bool() if "mask" in tgt else torch.ones_like(tgt["rgb"][:, 0]).bool()
    rgb, depth, rgb_t = pred["rgb"][mask], pred["depth"][mask], tgt["rgb"][mask]
    if "beta" in pred:
        b = pred["beta"][mask]
        loss = coef * (((rgb - rgb_t) ** 2 / (2 * b.unsqueeze(1) ** 2)).mean() + 3 + torch.log(b).mean())
    else:
        loss = coef * ((rgb - rgb_t) ** 2).mean()
    if use_depth and "depth" in tgt:
        td = tgt["depth"][mask]
        dm = td > 0
        near, far = tgt["depth_range"]
        tdn = to_inverse_normalized_depth(td, near, far)
        loss = loss + coef * (((to_inverse_normalized_depth(depth, near, far) - tdn) ** 2 * dm).sum() / (1e-8 + dm.sum()))
        if "depth_coarse" in pred:
            dc = to_inverse_normalized_depth(pred["depth_coarse"][mask], near, far)
            loss = loss + coef * (((dc - tdn) ** 2 * dm).sum() / (1e-8 + dm.sum()))
    if "feat" in pred and "feat" in tgt:
        loss = loss + coef * 0.1 * ((pred["feat"][mask] - tgt["feat"][mask]) ** 2).mean()
    return loss


def masked_psnr(x: Tensor, y: Tensor, mask: Tensor) -> Tensor:
    """conditional_nerf/utils.py:115-128 (img2mse with mask, mse2psnr)."""
    mse = torch.sum((x - y) * (x - y) * mask.unsqueeze(-1)) / (torch.sum(mask) * x.shape[-1] + 1e-8)
    return -10.0 * torch.log(mse) / torch.log(torch.tensor([10.0], device=x.device))
