"""Thin host wrapper over the C-ABI: torch tensors -> raw device pointers, stream handling, workspaces.

PyTorch is plumbing here (device memory + the current HIP stream); every computation on the ray path
happens inside libnerfloc_render.so.  Mirrors the pieces of state the reference keeps on the module:
packed weights <- state_dict, frame <- (`data`, `support_neural_points['fine']`, `vis_featmaps`).
"""
from __future__ import annotations

import ctypes as ct
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib as L


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dev_f32(t, device) -> torch.Tensor:
    t = torch.as_tensor(t)
    return t.to(device=device, dtype=torch.float32).contiguous()


class GraphedKeep:
    """The keep / kept pair of ONE batch shape against ONE frame and weight set as two HIP graphs (torch.cuda.CUDAGraph around the library calls): a pose
    refinement loop (pose_optimizer.py:131-168: ~50 steps of 512 rays against one frame) otherwise re-issues ~130 launches per step, which is what
    its wall time follows once the host's cores are busy.  Inputs are copied into static buffers, the graph is replayed, results are cloned out;
    frozen weights only (a training step meets a new frame — new camera kernel arguments — every time)."""

    def __init__(self, r: "HipRenderer", R: int, white: bool):
        dev, S, C = r.device, r.S, r.C
        self.r, self.R, self.white, self.gen = r, int(R), bool(white), r.state_gen
        e = lambda *shp: torch.zeros(*shp, device=dev)
        self.o, self.d, self.z, self.q = e(R, 3), e(R, 3), e(R, S), e(R, 3)
        self.out = {"rgb": e(R, 3), "depth": e(R), "weights": e(R, S), "mask": torch.zeros(R, dtype=torch.uint8, device=dev), "depth_uncertainty": e(R), "feat": e(R, C)}
        self.cot = {"g_rgb": e(R, 3), "g_depth": e(R), "g_depth_uncertainty": e(R), "g_feat": e(R, C), "g_weights": e(R, S)}
        self.go, self.gd, self.gq = e(R, 3), e(R, 3), e(R, 3)
        need = r.lib.nl_render_rays_keep_workspace_bytes(ct.byref(r.cfg), r.V, R, 0)
        self.ws = torch.empty(need, dtype=torch.uint8, device=dev)
        self.ro = L.NlRenderOut()
        for k, t in self.out.items():
            setattr(self.ro, k, t.data_ptr())
        self.c = L.NlRenderCotangents()
        self.c.g_rgb, self.c.g_depth, self.c.g_depth_uncertainty, self.c.g_feat, self.c.g_weights = [self.cot[k].data_ptr() for k in
                                                                                                      ("g_rgb", "g_depth", "g_depth_uncertainty", "g_feat", "g_weights")]
        self.fwd = self.bwd = None
        self.stream = torch.cuda.Stream(device=dev)
        self._pool = {"busy": False}   # the static buffers serve ONE forward / backward pair at a time
        self.fwd_count = 0             # generation of the kept activations: a node checks in backward that no later forward replaced them

    def acquire(self):
        """A lease on the static buffers (released when dropped), or None while another forward's backward is still pending."""
        if self._pool["busy"]:
            return None
        self._pool["busy"] = True
        return _Lease(self._pool)

    def _fwd_call(self):
        r = self.r
        L.check(r.lib.nl_render_rays_forward_keep(ct.byref(r.cfg), r.packed.data_ptr(), r._frame, None, self.q.data_ptr(), self.o.data_ptr(), self.d.data_ptr(),
                                                  self.z.data_ptr(), self.R, 1 if self.white else 0, ct.byref(self.ro), None, 0, self.ws.data_ptr(), self.ws.numel(),
                                                  r._stream()), "nl_render_rays_forward_keep")

    def _bwd_call(self):
        r = self.r
        L.check(r.lib.nl_render_rays_backward_kept(ct.byref(r.cfg), r.packed.data_ptr(), r._frame, None, self.q.data_ptr(), self.d.data_ptr(), self.R,
                                                   1 if self.white else 0, ct.byref(self.c), None, self.go.data_ptr(), self.gd.data_ptr(), self.gq.data_ptr(), None,
                                                   self.ws.data_ptr(), self.ws.numel(), r._stream()), "nl_render_rays_backward_kept")

    def _run(self, which):
        g = getattr(self, which)
        if g is None:   # first use: capture on a side stream (the library only enqueues kernels / async copies on the stream it is given)
            call = self._fwd_call if which == "fwd" else self._bwd_call
            cur = torch.cuda.current_stream(self.r.device)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                if which == "fwd":
                    call()                              # warm (per-frame tables that are built on first use).  NOT for the way back: it consumes the
                g = torch.cuda.CUDAGraph()              # kept activations (buffers are reused), so it must run exactly once per forward — the replay
                with torch.cuda.graph(g, stream=self.stream):
                    call()
            cur.wait_stream(self.stream)
            setattr(self, which, g)
        g.replay()

    def forward(self, o, d, z, q):
        self.o.copy_(o); self.d.copy_(d); self.z.copy_(z); self.q.copy_(q.reshape(-1, 3).expand(self.R, 3))
        self._run("fwd")
        self.fwd_count += 1
        out = {k: v.clone() for k, v in self.out.items()}
        out["mask"] = out["mask"].view(torch.bool)
        return out

    def backward(self, g_rgb=None, g_depth=None, g_depth_uncertainty=None, g_feat=None, g_weights=None):
        for k, v in (("g_rgb", g_rgb), ("g_depth", g_depth), ("g_depth_uncertainty", g_depth_uncertainty), ("g_feat", g_feat), ("g_weights", g_weights)):
            if v is None:
                self.cot[k].zero_()
            else:
                self.cot[k].copy_(v)
        self._run("bwd")
        return self.go.clone(), self.gd.clone(), self.gq.sum(0)


class GraphedRender:
    """The fused inference path (`nl_render_rays_ex`) of ONE batch shape against ONE frame / weight set / precision as a HIP graph (round 5).
    The call is an 18-kernel dependency chain on two streams; for a batch of a few hundred rays (BASELINE config 1; a 512-ray shard of config 2 on 8 GPUs; a
    pose-scoring loop) each launch's ramp and tail (~10 us) is a tenth of the step.  Inputs are copied into static buffers (the query centre travels as per-ray rows
    in device memory, so a new pose needs no new capture), the graph is replayed, the outputs are cloned out.  Results are bit-identical to the eager call
    (tests/test_gpu_configs.py).  The loop it replaces on the reference's side: one `render_rays` per chunk / pose (model.py:615-639)."""

    def __init__(self, r: "HipRenderer", R: int, white: bool, want_feat: bool, have_z: bool):
        dev, S, C = r.device, r.S, r.C
        self.r, self.R, self.white, self.gen, self.precision = r, int(R), bool(white), r.state_gen, r.precision
        e = lambda *shp: torch.zeros(*shp, device=dev)
        self.o, self.d, self.q = e(R, 3), e(R, 3), e(R, 3)
        self.z = e(R, S) if have_z else None
        self.out = {"rgb": e(R, 3), "depth": e(R), "weights": e(R, S), "mask": torch.zeros(R, dtype=torch.uint8, device=dev), "depth_uncertainty": e(R)}
        if want_feat:
            self.out["feat"] = e(R, C)
        self.ro = L.NlRenderOut()
        for k, t in self.out.items():
            setattr(self.ro, k, t.data_ptr())
        self.opts = L.NlRenderOpts()
        self.opts.ray_centers = self.q.data_ptr()
        need = r.lib.nl_render_rays_workspace_bytes(ct.byref(r.cfg), r.V, R)
        self.ws = torch.empty(need, dtype=torch.uint8, device=dev)
        self.graph = None
        self.stream = torch.cuda.Stream(device=dev)

    def _call(self):
        r = self.r
        L.check(r.lib.nl_render_rays_ex(ct.byref(r.cfg), r.packed.data_ptr(), r._frame, None, self.o.data_ptr(), self.d.data_ptr(), _ptr(self.z), self.R,
                                        1 if self.white else 0, ct.byref(self.ro), self.ws.data_ptr(), self.ws.numel(), r._stream(), ct.byref(self.opts)), "nl_render_rays")

    def run(self, o, d, q_rows, z, clone: bool = True):
        self.o.copy_(o, non_blocking=True); self.d.copy_(d, non_blocking=True); self.q.copy_(q_rows, non_blocking=True)
        if self.z is not None:
            self.z.copy_(z, non_blocking=True)
        if self.graph is None:   # first use: warm run (per-frame tables are built on first use), then the capture, on a side stream
            cur = torch.cuda.current_stream(self.r.device)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                self._call()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.stream):
                    self._call()
            cur.wait_stream(self.stream)
            self.graph = g
        self.graph.replay()
        out = {k: (v.clone() if clone else v) for k, v in self.out.items()}
        out["mask"] = out["mask"].view(torch.bool)
        return out


class _Lease:
    """Marks a renderer's pooled keep-workspace as free again when the state that borrowed it is dropped (after the backward call, or never used)."""

    def __init__(self, pool):
        self._pool = pool

    def __del__(self):
        self._pool["busy"] = False


class TrainGrads:
    """Gradient buffers of one training step (nl_train_grads): `.weights[name]` fp32 tensors shaped like the state_dict entries,
    `.support_feature` (M, C+3) or None; the library ADDS into them, so one object serves all backward calls (and chunks) of a step."""

    def __init__(self, renderer: "HipRenderer", names: Sequence[str], support_feature: bool, feat_maps: bool = False, vis_featmaps: bool = False,
                 blend_feat_maps: bool = False):
        all_names = L.weight_names()
        shapes = renderer.weight_shapes()
        unknown = [n for n in names if n not in shapes]
        if unknown:
            raise KeyError(f"not a weight of the ray path: {unknown}")
        dev = renderer.device
        # one zero-filled buffer, one fill kernel (84 separate torch.zeros calls were 0.4 ms of a 14 ms step); every tensor starts on a 64-byte boundary
        offs, tot = {}, 0
        for n in names:
            offs[n] = tot
            tot += (int(np.prod(shapes[n])) + 15) // 16 * 16
        flat = torch.zeros(max(tot, 1), device=dev)
        self.weights = {n: flat[offs[n]:offs[n] + int(np.prod(shapes[n]))].view(shapes[n]) for n in names}
        self._arr = (ct.c_void_p * len(all_names))()
        for i, n in enumerate(all_names):
            self._arr[i] = self.weights[n].data_ptr() if n in self.weights else None
        self.support_feature = torch.zeros(renderer.M, renderer.C + 3, device=dev) if support_feature else None
        shp = renderer._map_shapes if (feat_maps or vis_featmaps or blend_feat_maps) else (None, None, None)
        self.feat_maps = torch.zeros(shp[0], device=dev) if feat_maps else None
        self._vis_hwc = torch.zeros(shp[1], device=dev) if vis_featmaps else None
        self.blend_feat_maps = torch.zeros(shp[2], device=dev) if blend_feat_maps else None
        nb = renderer.lib.nl_train_scratch_bytes(ct.byref(renderer.cfg))
        self._scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        self.c = L.NlTrainGrads()
        self.c.weights = ct.cast(self._arr, ct.POINTER(ct.c_void_p))
        self.c.support_feature = _ptr(self.support_feature)
        self.c.feat_maps, self.c.vis_featmaps, self.c.blend_feat_maps = _ptr(self.feat_maps), _ptr(self._vis_hwc), _ptr(self.blend_feat_maps)
        self.c.scratch = self._scratch.data_ptr()
        self.c.scratch_bytes = nb

    @property
    def vis_featmaps(self):
        """(V, 32, vh, vw) like data's DepthFusionNet maps (the library accumulates channels-last)."""
        return None if self._vis_hwc is None else self._vis_hwc.permute(0, 3, 1, 2)


class HipRenderer:
    """One renderer per (device, W, C, S, precision).  Not thread-safe (like the reference module)."""
    supports_out_buffers = True   # render_rays(out_buffers=...): sharding.py renders into one buffer per rank and gathers that buffer

    def __init__(self, W: int, C: int, S: int, precision: str = "bf16x3", device: str = "cuda:0", workspace_bytes: Optional[int] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("HipRenderer needs a HIP device; there is no CPU fallback (the oracle is test-only)")
        self.lib = L.load()
        self.device = torch.device(device)
        self.cfg = L.NlConfig(W, C, S, L.PRECISIONS[precision])
        self.W, self.C, self.S = W, C, S
        self.precision = precision
        nbytes = self.lib.nl_packed_weights_bytes(ct.byref(self.cfg))
        if nbytes == 0:
            raise ValueError(f"unsupported renderer config W={W} C={C} S={S}")
        self.packed = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._weights_loaded = False
        self._frame = ct.c_void_p(None)
        self._frame_keep = None
        self._ws = None
        self._ws_request = workspace_bytes
        self.V = 0
        self._zc_rows = {}

    # ------------------------------------------------------------------ weights
    def load_weights(self, state_dict: Dict[str, torch.Tensor]) -> None:
        names = L.weight_names()
        keep = []
        arr = (ct.c_void_p * len(names))()
        for i, n in enumerate(names):
            if n not in state_dict:
                raise KeyError(f"state_dict is missing {n}")
            t = _dev_f32(state_dict[n].detach() if isinstance(state_dict[n], torch.Tensor) else state_dict[n], self.device)
            keep.append(t)
            arr[i] = t.data_ptr()
        self._weight_shapes = {n: tuple(t.shape) for n, t in zip(names, keep)}
        self._weight_tensors = dict(zip(names, keep))   # fp32 device copies as packed (a few MB): the training nodes read some of them back
        self.state_gen = getattr(self, "state_gen", 0) + 1   # autograd nodes check that weights / frame did not change between their forward and backward
        st = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.nl_pack_weights(ct.byref(self.cfg), arr, len(names), self.packed.data_ptr(), self.packed.numel(), st), "nl_pack_weights")
        torch.cuda.current_stream(self.device).synchronize()  # sources may be freed after this
        self._weights_loaded = True

    def weight_shapes(self) -> Dict[str, tuple]:
        """Shapes of the state_dict tensors the ray path consumes (as last loaded)."""
        if not self._weights_loaded:
            raise RuntimeError("load_weights first")
        return dict(self._weight_shapes)

    def weight_tensor(self, name: str) -> torch.Tensor:
        """The fp32 device copy of a state_dict tensor as it was last packed."""
        if not self._weights_loaded:
            raise RuntimeError("load_weights first")
        return self._weight_tensors[name]

    def train_capable(self) -> bool:
        """Whether the library's TRAINING nodes (`nl_*_backward_train`) take this configuration.  Their weight-gradient products read 16-byte rows:
        nl_config.C must be a multiple of 4 (include/nerfloc_render.h, "training: weight gradients"; wgrad.hip returns NL_ERR_UNSUPPORTED otherwise) and
        the scratch query must accept the configuration.  Probed BEFORE a forward pass chooses its autograd nodes: a refusal discovered inside
        `loss.backward()` has no eager graph left to fall back to."""
        ok = self.__dict__.get("_train_capable")
        if ok is None:
            ok = self.C % 4 == 0 and self.W <= 256 and int(self.lib.nl_train_scratch_bytes(ct.byref(self.cfg))) > 0
            self._train_capable = ok
        return ok

    def set_precision(self, precision: str) -> None:
        """All three weight layouts are packed at once, so switching is free."""
        self.precision = precision
        self.cfg.precision = L.PRECISIONS[precision]

    # ------------------------------------------------------------------ frame
    def set_frame(self, images, featmaps_hwc, vis_featmaps, Ks, poses, near: float, far: float, support: Dict[str, torch.Tensor]) -> None:
        """images (V,3,H,W); featmaps_hwc (V,h,w,C) = data['feat_fine_src']; vis_featmaps (V,32,h,w);
        Ks (V,3,3); poses (V,4,4) c2w; support = support_neural_points['fine'] (xyz, feature, confidence, direction)."""
        self.clear_frame()
        dev = self.device
        images = _dev_f32(images, dev)
        feat = _dev_f32(featmaps_hwc, dev)
        visf = _dev_f32(vis_featmaps, dev)
        V, _, H, Wimg = images.shape
        h, w = feat.shape[1], feat.shape[2]
        if V > L.MAX_VIEWS:
            raise ValueError(f"at most {L.MAX_VIEWS} support views")
        # the tiny per-view matrices are formed on the host exactly like the reference forms them
        Kc = torch.as_tensor(Ks).detach().float().cpu()
        Pc = torch.as_tensor(poses).detach().float().cpu()
        K4 = torch.eye(4).expand(V, 4, 4).clone()
        K4[:, :3, :3] = Kc
        proj_ibr = K4.bmm(torch.inverse(Pc))[:, :3].contiguous()       # ibrnet.py:183
        proj_neuray = (Kc @ Pc.inverse()[:, :3]).contiguous()          # depth_fusion.py:90, multiview_aggregator.py:184
        cams = Pc[:, :3, 3].contiguous()
        sp = {k: _dev_f32(support[k], dev) for k in ("xyz", "feature", "confidence", "direction")}
        M = sp["xyz"].shape[0]
        d = L.NlFrameDesc()
        d.V, d.H, d.Wimg, d.h, d.w = V, H, Wimg, h, w
        d.vis_h, d.vis_w = visf.shape[2], visf.shape[3]
        d.near_, d.far_ = float(near), float(far)
        d.images, d.featmaps, d.vis_featmaps = images.data_ptr(), feat.data_ptr(), visf.data_ptr()
        d.proj_ibr, d.proj_neuray, d.cam_centers = proj_ibr.data_ptr(), proj_neuray.data_ptr(), cams.data_ptr()
        d.M = M
        if M > 0:
            d.sp_xyz, d.sp_feature = sp["xyz"].data_ptr(), sp["feature"].data_ptr()
            d.sp_confidence, d.sp_direction = sp["confidence"].data_ptr(), sp["direction"].data_ptr()
        nbytes = self.lib.nl_frame_bytes(ct.byref(self.cfg), ct.byref(d))
        if nbytes == 0:
            raise ValueError("nl_frame_bytes rejected the frame description")
        mem = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        fr = ct.c_void_p(None)
        L.check(self.lib.nl_frame_create(ct.byref(self.cfg), ct.byref(d), mem.data_ptr(), nbytes, st, ct.byref(fr)), "nl_frame_create")
        self._frame = fr
        self._frame_keep = (images, feat, visf, sp, mem, proj_ibr, proj_neuray, cams)
        self.V, self.near, self.far, self.M = V, float(near), float(far), M
        self._map_shapes = ((V, h, w, feat.shape[3]), (V, visf.shape[2], visf.shape[3], 32), (V, h, w, 32))
        self.state_gen = getattr(self, "state_gen", 0) + 1

    def diagnostics(self) -> Dict[str, float]:
        """nl_frame_diagnostics: {'table_absmax': max |T| of the per-frame table, 'logit_absmax': the largest |attention logit| the fused neural-point kernel has
        scored against this frame so far}.  Synchronises the current stream (a device-to-host copy of two floats)."""
        self._ready()
        buf = (ct.c_float * L.DIAG_COUNT)()
        L.check(self.lib.nl_frame_diagnostics(self._frame, buf, L.DIAG_COUNT, self._stream()), "nl_frame_diagnostics")
        return {"table_absmax": float(buf[0]), "logit_absmax": float(buf[1]), "point_kernel_GHz": float(buf[2]),
                "guard_precision": L.PRECISION_NAMES.get(int(buf[3])), "guard_escalations": int(buf[4])}

    def clear_frame(self) -> None:
        if self._frame:
            self.lib.nl_frame_destroy(self._frame)
        self._frame = ct.c_void_p(None)
        self._frame_keep = None
        self._zc_rows = {}

    def __del__(self):
        try:
            self.clear_frame()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _workspace(self, nbytes: int) -> torch.Tensor:
        if getattr(self, "guard_bytes", 0):
            return self._guarded(nbytes)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    # test facility: with `guard_bytes` > 0 every workspace handed to the library is EXACTLY the size its *_workspace_bytes query asked for, followed
    # by a canary region; `check_guards()` verifies that no call wrote past its workspace
    def set_guard(self, guard_bytes: int, gap_bytes: int = 0) -> None:
        """guard_bytes: canary behind every workspace; gap_bytes: also BETWEEN the buffers the library carves from it (nl_debug_bump_gap: process-global,
        switch it off again with set_guard(0))."""
        self.guard_bytes = int(guard_bytes)
        L.check(self.lib.nl_debug_bump_gap(int(gap_bytes)), "nl_debug_bump_gap")

    def _guarded(self, nbytes: int) -> torch.Tensor:
        buf = torch.full((int(nbytes) + int(self.guard_bytes),), 0xA5, dtype=torch.uint8, device=self.device)
        self.__dict__.setdefault("_guards", []).append((buf, int(nbytes)))
        return buf[:nbytes]

    def check_guards(self) -> int:
        torch.cuda.synchronize(self.device)
        n = 0
        for buf, nb in self.__dict__.get("_guards", []):
            if not bool((buf[nb:] == 0xA5).all()):
                raise AssertionError(f"a library call wrote past its {nb}-byte workspace")
            n += 1
        bad, checked = ct.c_int(0), ct.c_int(0)
        scratch = torch.zeros(1, dtype=torch.int32, device=self.device)
        L.check(self.lib.nl_debug_check_gaps(0xA5, scratch.data_ptr(), ct.byref(bad), ct.byref(checked), self._stream()), "nl_debug_check_gaps")
        if bad.value:
            raise AssertionError(f"a kernel wrote outside its buffer inside a workspace ({bad.value} of {checked.value} gap regions touched)")
        self.gaps_checked = getattr(self, "gaps_checked", 0) + checked.value
        self._guards = []
        return n

    def _centres(self, query_center, R: int):
        """(host pointer tensor or None, device (R,3) rows or None): a query centre that lives on the device (a pose being optimised) is handed over as per-ray
        rows — no device-to-host copy, i.e. no synchronisation point in the middle of a training / refinement loop."""
        qc = torch.as_tensor(query_center).detach()
        if qc.is_cuda:
            rows = qc.float().reshape(-1, 3)
            rows = rows.expand(R, 3) if rows.shape[0] == 1 else rows
            return None, rows.contiguous()
        return qc.float().reshape(3).contiguous(), None

    def _ready(self):
        if not self._weights_loaded:
            raise RuntimeError("load_weights() first")
        if not self._frame:
            raise RuntimeError("set_frame() first")

    # ------------------------------------------------------------------ fused path
    GRAPH_MAX_RAYS = 1024   # batches up to this size replay as a HIP graph when the caller asks for it (render_rays(graph=True)); larger ones are GPU-bound

    def _graphed_render(self, R: int, white: bool, want_feat: bool, have_z: bool):
        """The GraphedRender of this batch shape for the CURRENT frame / weights / precision: None on the first request of a shape (the caller renders that batch
        eagerly: it may be the only one), the graph from the second request on; dropped when the frame or the weights change; at most four shapes are kept."""
        reg = self.__dict__.setdefault("_rgraphs", {})
        if reg.get("gen") != (self.state_gen, self.precision):
            reg.clear()
            reg["gen"] = (self.state_gen, self.precision)
        key = (int(R), bool(white), bool(want_feat), bool(have_z))
        ent = reg.get(key)
        if ent is None:
            reg[key] = "seen"
            return None
        if ent == "seen":
            live = [k for k, v in reg.items() if isinstance(v, GraphedRender)]
            for k in live[:-3]:
                reg[k] = "seen"
            ent = reg[key] = GraphedRender(self, R, white, want_feat, have_z)
        return ent

    def render_rays(self, rays_o, rays_d, query_center, z_vals=None, white_bkgd: bool = False,
                    intermediates: bool = False, want_feat: bool = True, early_term_eps: float = 0.0,
                    side_stream: bool = True, want_knn: bool = False, graph: bool = False, precision_guard: bool = False,
                    out_buffers: Optional[Dict[str, torch.Tensor]] = None, want_weights: bool = True) -> Dict[str, torch.Tensor]:
        """want_weights=False: the per-sample compositing weights (R, S) are not returned (nl_render_out.weights = null; every other output is unchanged).
        out_buffers: preallocated, contiguous destination tensors for the per-ray outputs (rgb (R,3), depth (R), weights (R,S), mask (R) uint8, depth_uncertainty (R),
        feat (R,C); fp32 but for the mask) — the kernels write straight into them (sharding.py hands in views of ONE buffer per rank, so the all-gather needs no pack step).
        precision_guard=True (nl_render_opts.flags = NL_RENDER_PRECISION_GUARD, ABI 7): the LIBRARY checks the frame's conditioning indicator after the batch
        (max |attention logit|; one 4-byte copy + a stream synchronisation) and renders the batch again in the next more exact mode (f16mx -> bf16x3 -> fp32) while
        it lies beyond the validated range of the mode the outputs were produced in; the frame then stays in that mode for later guarded calls
        (`diagnostics()['guard_precision']`).  Off by default: the synchronisation keeps the host from running ahead of the device.
        side_stream=False (nl_render_opts.flags = NL_RENDER_NO_SIDE_STREAM): every kernel on the current stream (bit-identical results;
        for profiling kernels one at a time).
        early_term_eps > 0: early-termination compositing (nl_render_opts): colours / features of the samples behind the point where a
        ray's transmittance falls below eps are not evaluated (rgb / feat move by < eps * max|value|; everything else is unchanged).
        query_center: (3,) for the whole batch, or (R, 3) per ray — rays of several query frames (poses) against this support frame in
        one launch (nl_render_opts.ray_centers).
        graph=True: a batch of <= GRAPH_MAX_RAYS rays whose shape was seen before (same frame, weights, precision) is replayed as a HIP graph
        (GraphedRender: bit-identical outputs, one graph launch instead of an 18-kernel chain)."""
        self._ready()
        dev = self.device
        o, d = _dev_f32(rays_o, dev), _dev_f32(rays_d, dev)
        R, S, W = o.shape[0], self.S, self.W
        z = None if z_vals is None else _dev_f32(z_vals, dev)
        qc_t = torch.as_tensor(query_center).detach().float()
        per_ray = qc_t.dim() == 2
        if per_ray and tuple(qc_t.shape) != (R, 3):
            raise ValueError(f"per-ray query centres must have shape ({R}, 3), got {tuple(qc_t.shape)}")
        if graph and 0 < R <= self.GRAPH_MAX_RAYS and not intermediates and not want_knn and early_term_eps == 0.0 and side_stream and not precision_guard and out_buffers is None and want_weights \
                and not getattr(self, "guard_bytes", 0) and self._ws_request is None:
            g = self._graphed_render(R, white_bkgd, want_feat, z is not None)
            if g is not None:
                rows = qc_t.to(dev) if per_ray else qc_t.reshape(1, 3).to(dev).expand(R, 3)
                return g.run(o, d, rows, z)
        if not per_ray and qc_t.is_cuda:   # one centre that lives on the device: R identical rows instead of a device-to-host copy (= a synchronisation
            qc_t, per_ray = qc_t.reshape(1, 3).expand(R, 3), True   # point per call: 21 of them in a render_image loop)
        qc = qc_t.to(dev).contiguous() if per_ray else qc_t.cpu().contiguous()
        if out_buffers is not None:
            want = {"rgb": (R, 3), "depth": (R,), "weights": (R, S), "mask": (R,), "depth_uncertainty": (R,)}
            if want_feat:
                want["feat"] = (R, self.C)
            out = {}
            for k, shp in want.items():
                t = out_buffers[k]
                if tuple(t.shape) != shp or not t.is_contiguous() or t.device != dev or t.dtype != (torch.uint8 if k == "mask" else torch.float32):
                    raise ValueError(f"out_buffers['{k}']: expected a contiguous {'uint8' if k == 'mask' else 'float32'} tensor of shape {shp} on {dev}")
                out[k] = t
        else:
            out = {
                "rgb": torch.empty(R, 3, device=dev), "depth": torch.empty(R, device=dev), "weights": torch.empty(R, S, device=dev),
                "mask": torch.empty(R, dtype=torch.uint8, device=dev), "depth_uncertainty": torch.empty(R, device=dev),
            }
            if want_feat:
                out["feat"] = torch.empty(R, self.C, device=dev)
            if not want_weights:
                del out["weights"]
        if want_knn and not intermediates:   # the neighbours alone (the gradient path hands them to nl_render_rays_backward)
            out.update({"knn_idx": torch.empty(R * S, 8, dtype=torch.int32, device=dev), "knn_d2": torch.empty(R * S, 8, device=dev)})
        if intermediates:
            N = R * S
            out.update({"sigma": torch.empty(N, device=dev), "feature_agg": torch.empty(N, W, device=dev),
                        "mv_feature_agg": torch.empty(N, W, device=dev), "geo": torch.empty(N, W, device=dev),
                        "knn_idx": torch.empty(N, 8, dtype=torch.int32, device=dev), "knn_d2": torch.empty(N, 8, device=dev)})
        ro = L.NlRenderOut()
        for k, t in out.items():
            setattr(ro, k, t.data_ptr())
        need = self._ws_request or self.lib.nl_render_rays_workspace_bytes(ct.byref(self.cfg), self.V, R)
        ws = self._workspace(need)
        opts = L.NlRenderOpts()
        opts.early_term_eps = float(early_term_eps)
        if per_ray:
            opts.ray_centers = qc.data_ptr()
        if not side_stream:
            opts.flags |= L.RENDER_NO_SIDE_STREAM
        if precision_guard:
            opts.flags |= L.RENDER_PRECISION_GUARD
        L.check(self.lib.nl_render_rays_ex(ct.byref(self.cfg), self.packed.data_ptr(), self._frame, None if per_ray else qc.data_ptr(), o.data_ptr(),
                                           d.data_ptr(), _ptr(z), R, int(bool(white_bkgd)), ct.byref(ro), ws.data_ptr(), ws.numel(), self._stream(),
                                           ct.byref(opts) if (early_term_eps > 0 or per_ray or not side_stream or precision_guard) else None), "nl_render_rays")
        out["mask"] = out["mask"].view(torch.bool)   # 0 / 1 bytes reinterpreted: no conversion kernel
        if intermediates:
            out["sigma"] = out["sigma"].view(R, S)
        return out

    # ------------------------------------------------------------------ stages (used by tests and staged callers)
    def knn(self, xyz, K: int = 8):
        self._ready()
        x = _dev_f32(xyz, self.device)
        N = x.shape[0]
        idx = torch.empty(N, K, dtype=torch.int32, device=self.device)
        d2 = torch.empty(N, K, device=self.device)
        L.check(self.lib.nl_knn(self._frame, x.data_ptr(), N, K, idx.data_ptr(), d2.data_ptr(), self._stream()), "nl_knn")
        return d2, idx

    def sample_points(self, rays_o, rays_d, z_vals=None):
        o, d = _dev_f32(rays_o, self.device), _dev_f32(rays_d, self.device)
        R = o.shape[0]
        z = None if z_vals is None else _dev_f32(z_vals, self.device)
        zo = torch.empty(R, self.S, device=self.device)
        xyz = torch.empty(R * self.S, 3, device=self.device)
        L.check(self.lib.nl_sample_points(o.data_ptr(), d.data_ptr(), R, self.S, self.near, self.far, _ptr(z), zo.data_ptr(), xyz.data_ptr(), self._stream()), "nl_sample_points")
        return zo, xyz

    def mv_aggregate(self, xyz, query_center, want_raw: bool = True):
        """nl_mv_aggregate.  want_raw=False skips the raw per-view tensors (rgb_feat, vis_ang come back as None) — the form the
        fused render path uses, which selects the eight-samples-per-wave gather kernel."""
        self._ready()
        x = _dev_f32(xyz, self.device)
        N, V = x.shape[0], self.V
        qc = torch.as_tensor(query_center).detach().float().cpu().contiguous()
        mv = torch.empty(N, self.W, device=self.device)
        valid = torch.empty(N, dtype=torch.int32, device=self.device)
        if not want_raw:
            ws = self._workspace(self.lib.nl_mv_aggregate_workspace_bytes(ct.byref(self.cfg), V, N))
            L.check(self.lib.nl_mv_aggregate(ct.byref(self.cfg), self.packed.data_ptr(), self._frame, qc.data_ptr(), x.data_ptr(), N, mv.data_ptr(),
                                             None, None, valid.data_ptr(), None, None, ws.data_ptr(), ws.numel(), self._stream()), "nl_mv_aggregate")
            return mv, None, None, valid
        rgb_feat = torch.empty(N * V, 196, device=self.device)
        vis_ang = torch.empty(N * V, 8, device=self.device)
        ws = self._workspace(self.lib.nl_mv_aggregate_workspace_bytes(ct.byref(self.cfg), V, N))
        L.check(self.lib.nl_mv_aggregate(ct.byref(self.cfg), self.packed.data_ptr(), self._frame, qc.data_ptr(), x.data_ptr(), N, mv.data_ptr(),
                                         rgb_feat.data_ptr(), vis_ang.data_ptr(), valid.data_ptr(), None, None, ws.data_ptr(), ws.numel(), self._stream()), "nl_mv_aggregate")
        return mv, rgb_feat.view(N, V, 196), vis_ang.view(N, V, 8), valid

    def point_mlp(self, xyz, direction, mv_feat, K: int = 8):
        self._ready()
        x = _dev_f32(xyz, self.device)
        N = x.shape[0]
        dr = None if direction is None else _dev_f32(direction, self.device)
        g = _dev_f32(mv_feat, self.device)
        fa = torch.empty(N, self.W, device=self.device)
        idx = torch.empty(N, K, dtype=torch.int32, device=self.device)
        d2 = torch.empty(N, K, device=self.device)
        ws = self._workspace(self.lib.nl_point_mlp_workspace_bytes(ct.byref(self.cfg), N))
        L.check(self.lib.nl_point_mlp(ct.byref(self.cfg), self.packed.data_ptr(), self._frame, x.data_ptr(), _ptr(dr), 0 if dr is None else dr.shape[1],
                                      g.data_ptr(), N, K, fa.data_ptr(), idx.data_ptr(), d2.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()), "nl_point_mlp")
        return fa, d2, idx

    def render_rays_backward(self, rays_o, rays_d, z_vals, query_center, g_rgb=None, g_depth=None, g_depth_uncertainty=None, g_feat=None, g_weights=None,
                             white_bkgd: bool = False, want_g_query_center: bool = False, train: "TrainGrads" = None, workspace_rays: Optional[int] = None,
                             knn=None):
        """The whole ray path backwards in one library call (nl_render_rays_backward): cotangents of render_rays' per-ray outputs ->
        (g_rays_o (R,3), g_rays_d (R,3), g_query_center (3,) or None); train: also ADD every parameter / map / table gradient into that TrainGrads.
        z_vals (R,S): the sample depths of the forward call."""
        self._ready()
        dev = self.device
        o, d, z = _dev_f32(rays_o, dev), _dev_f32(rays_d, dev), _dev_f32(z_vals, dev)
        R = o.shape[0]
        if z.shape != (R, self.S):
            raise ValueError(f"z_vals must be ({R}, {self.S})")
        qc, qrows = self._centres(query_center, R)
        cots = [None if t is None else _dev_f32(t, dev) for t in (g_rgb, g_depth, g_depth_uncertainty, g_feat, g_weights)]
        c = L.NlRenderCotangents()
        c.g_rgb, c.g_depth, c.g_depth_uncertainty, c.g_feat, c.g_weights = [_ptr(t) for t in cots]
        if knn is not None:   # (d2, idx) of the forward call: saves the second neighbour search
            kd2, kidx = knn[0].contiguous(), knn[1].to(torch.int32).contiguous()
            c.knn_d2, c.knn_idx = kd2.data_ptr(), kidx.data_ptr()
        go, gd = torch.empty(R, 3, device=dev), torch.empty(R, 3, device=dev)
        gq = torch.empty(R, 3, device=dev) if want_g_query_center else None
        if workspace_rays is None:
            workspace_rays = getattr(self, "backward_rays_per_chunk", None)   # None: the library's default (~64 k samples per chunk, ~90 KB each at W = 256)
        ws = self._workspace(self.lib.nl_render_rays_backward_workspace_bytes(ct.byref(self.cfg), self.V, R if workspace_rays is None else int(workspace_rays),
                                                                              0 if train is None else 1))
        L.check(self.lib.nl_render_rays_backward(ct.byref(self.cfg), self.packed.data_ptr(), self._frame, _ptr(qc), _ptr(qrows), o.data_ptr(), d.data_ptr(), z.data_ptr(), R,
                                                 1 if white_bkgd else 0, ct.byref(c), go.data_ptr(), gd.data_ptr(), _ptr(gq),
                                                 None if train is None else ct.byref(train.c), ws.data_ptr(), ws.numel(), self._stream()),
                "nl_render_rays_backward")
        return go, gd, (None if gq is None else gq.sum(0))

    def render_rays_keep(self, rays_o, rays_d, z_vals, query_center, white_bkgd: bool = False, train: bool = False, max_bytes: Optional[int] = None,
                         beta_head=None, beta_min: float = 0.1):
        """Forward of the gradient path that KEEPS its staged activations (nl_render_rays_forward_keep): -> (outputs dict, state) where `state` goes to
        `render_rays_backward_kept` — or None when the batch does not fit `max_bytes` of workspace as one chunk (then: render_rays + render_rays_backward).
        beta_head = (beta_mlp.0.weight, beta_mlp.0.bias): also the training-mode uncertainty output 'beta' (model.py:587-592)."""
        self._ready()
        dev = self.device
        o, d, z = _dev_f32(rays_o, dev), _dev_f32(rays_d, dev), _dev_f32(z_vals, dev)
        R, S = o.shape[0], self.S
        need = self.lib.nl_render_rays_keep_workspace_bytes(ct.byref(self.cfg), self.V, R, 1 if train else 0)
        if need == 0 or (max_bytes is not None and need > max_bytes):
            return None
        qc, qrows = self._centres(query_center, R)
        out = {"rgb": torch.empty(R, 3, device=dev), "depth": torch.empty(R, device=dev), "weights": torch.empty(R, S, device=dev),
               "mask": torch.empty(R, dtype=torch.uint8, device=dev), "depth_uncertainty": torch.empty(R, device=dev), "feat": torch.empty(R, self.C, device=dev)}
        ro = L.NlRenderOut()
        for k, t in out.items():
            setattr(ro, k, t.data_ptr())
        # owned by the returned state, not the shared workspace: it must survive until the backward call.  One buffer per renderer is pooled (a training /
        # refinement loop would otherwise allocate and free gigabytes every step: the caching allocator falls back to hipMalloc when it cannot serve
        # that from its cache, milliseconds each time); a second forward before the first one's backward gets a buffer of its own
        lease = None
        if getattr(self, "guard_bytes", 0):
            ws = self._guarded(need)
        else:
            pool = self.__dict__.setdefault("_keep_pool", {"buf": None, "busy": False})
            if not pool["busy"]:
                if pool["buf"] is None or pool["buf"].numel() < need:
                    pool["buf"] = None
                    pool["buf"] = torch.empty(need, dtype=torch.uint8, device=dev)
                pool["busy"] = True
                ws, lease = pool["buf"][:need], _Lease(pool)
            else:
                ws = torch.empty(need, dtype=torch.uint8, device=dev)
        bh, bkeep = None, None
        if beta_head is not None:
            bw, bb = _dev_f32(beta_head[0], dev).reshape(-1), _dev_f32(beta_head[1], dev).reshape(-1)
            out["beta"] = torch.empty(R, device=dev)
            bh = L.NlBetaHead()
            bh.weight, bh.bias, bh.beta_min, bh.beta = bw.data_ptr(), bb.data_ptr(), float(beta_min), out["beta"].data_ptr()
            bkeep = (bw, bb, float(beta_min))
        L.check(self.lib.nl_render_rays_forward_keep(ct.byref(self.cfg), self.packed.data_ptr(), self._frame, _ptr(qc), _ptr(qrows), o.data_ptr(), d.data_ptr(), z.data_ptr(), R,
                                                     1 if white_bkgd else 0, ct.byref(ro), None if bh is None else ct.byref(bh), 1 if train else 0, ws.data_ptr(),
                                                     ws.numel(), self._stream()), "nl_render_rays_forward_keep")
        out["mask"] = out["mask"].view(torch.bool)
        return out, (ws, (qc, qrows), d, R, bool(white_bkgd), bool(train), bkeep, lease)

    def render_rays_backward_kept(self, state, g_rgb=None, g_depth=None, g_depth_uncertainty=None, g_feat=None, g_weights=None, want_g_query_center: bool = False,
                                  train: "TrainGrads" = None, g_beta=None, want_beta_grads: bool = False):
        """nl_render_rays_backward_kept: the way back from the activations `render_rays_keep` left in `state` -> (g_rays_o, g_rays_d, g_query_center or None)
        [+ (g_beta_weight (1,W), g_beta_bias (1,)) when want_beta_grads]."""
        ws, (qc, qrows), d, R, white, was_train, bkeep, _lease = state   # (the lease returns the pooled buffer when the state is dropped)
        if (train is not None) != was_train:
            raise ValueError("the state was made for " + ("a training" if was_train else "a frozen-weights") + " backward pass")
        dev = self.device
        cots = [None if t is None else _dev_f32(t, dev) for t in (g_rgb, g_depth, g_depth_uncertainty, g_feat, g_weights)]
        c = L.NlRenderCotangents()
        c.g_rgb, c.g_depth, c.g_depth_uncertainty, c.g_feat, c.g_weights = [_ptr(t) for t in cots]
        go, gd = torch.empty(R, 3, device=dev), torch.empty(R, 3, device=dev)
        gq = torch.empty(R, 3, device=dev) if want_g_query_center else None
        bh, gbw, gbb = None, None, None
        if bkeep is not None and g_beta is not None:
            gbt = _dev_f32(g_beta, dev)
            bh = L.NlBetaHead()
            bh.weight, bh.bias, bh.beta_min, bh.g_beta = bkeep[0].data_ptr(), bkeep[1].data_ptr(), bkeep[2], gbt.data_ptr()
            if want_beta_grads and train is not None:
                gbw, gbb = torch.zeros(1, self.W, device=dev), torch.zeros(1, device=dev)
                bh.g_weight, bh.g_bias = gbw.data_ptr(), gbb.data_ptr()
        L.check(self.lib.nl_render_rays_backward_kept(ct.byref(self.cfg), self.packed.data_ptr(), self._frame, _ptr(qc), _ptr(qrows), d.data_ptr(), R, 1 if white else 0,
                                                      ct.byref(c), None if bh is None else ct.byref(bh), go.data_ptr(), gd.data_ptr(), _ptr(gq),
                                                      None if train is None else ct.byref(train.c), ws.data_ptr(), ws.numel(), self._stream()),
                "nl_render_rays_backward_kept")
        res = (go, gd, (None if gq is None else gq.sum(0)))
        return res + ((gbw, gbb),) if want_beta_grads else res

    def graphed_keep(self, R: int, white_bkgd: bool = False, max_bytes: Optional[int] = None):
        """The GraphedKeep of this batch shape for the CURRENT frame / weights: None on the first request of a shape (the caller runs that step
        ungraphed: it may be the only one), the graphs from the second request on; dropped when the frame or the weights change."""
        self._ready()
        if getattr(self, "guard_bytes", 0):
            return None
        reg = self.__dict__.setdefault("_graphs", {})
        if reg.get("gen") != self.state_gen:
            reg.clear()
            reg["gen"] = self.state_gen
        key = (int(R), bool(white_bkgd))
        ent = reg.get(key)
        if ent is None:
            reg[key] = "seen"
            return None
        if ent == "seen":
            if max_bytes is not None and self.lib.nl_render_rays_keep_workspace_bytes(ct.byref(self.cfg), self.V, int(R), 0) > max_bytes:
                return None
            # (every captured shape owns its static workspace — gigabytes: at most two live, the older one makes room)
            live = [k for k, v in reg.items() if isinstance(v, GraphedKeep)]
            for k in live[:-1]:
                reg[k] = "seen"
            ent = reg[key] = GraphedKeep(self, R, white_bkgd)
        return ent

    def ray_unet_backward(self, x, g_geo, workspace_rays: Optional[int] = None, train: "TrainGrads" = None):
        """Input gradient of `ray_unet` (nl_ray_unet_backward): x, g_geo (R*S, W) -> g_x (R*S, W).  train: also ADD the gradients of the 28 U-Net tensors
        into that TrainGrads (nl_ray_unet_backward_train)."""
        if not self._weights_loaded:
            raise RuntimeError("load_weights() first")
        xin, g = _dev_f32(x, self.device), _dev_f32(g_geo, self.device)
        R = xin.shape[0] // self.S
        gx = torch.empty_like(xin)
        wsb = self.lib.nl_ray_unet_backward_workspace_bytes if train is None else self.lib.nl_ray_unet_backward_train_workspace_bytes
        ws = self._workspace(wsb(ct.byref(self.cfg), R))
        if workspace_rays is not None:
            ws = ws[: wsb(ct.byref(self.cfg), int(workspace_rays))]
        if train is not None:
            L.check(self.lib.nl_ray_unet_backward_train(ct.byref(self.cfg), self.packed.data_ptr(), xin.data_ptr(), R, g.data_ptr(), gx.data_ptr(), ct.byref(train.c),
                                                        ws.data_ptr(), ws.numel(), self._stream()), "nl_ray_unet_backward_train")
            return gx
        L.check(self.lib.nl_ray_unet_backward(ct.byref(self.cfg), self.packed.data_ptr(), xin.data_ptr(), R, g.data_ptr(), gx.data_ptr(), ws.data_ptr(), ws.numel(),
                                              self._stream()), "nl_ray_unet_backward")
        return gx

    def mv_aggregate_backward(self, xyz, g_mv_feat, workspace_samples: Optional[int] = None, train: "TrainGrads" = None):
        """Input gradient of `mv_aggregate`'s feature rows (nl_mv_aggregate_backward): -> g_xyz (N,3).  train: also ADD the gradients of out_fc, the
        decoders, the feature maps and the DepthFusionNet maps into that TrainGrads (nl_mv_aggregate_backward_train)."""
        self._ready()
        x, g = _dev_f32(xyz, self.device), _dev_f32(g_mv_feat, self.device)
        N = x.shape[0]
        gx = torch.empty(N, 3, device=self.device)
        wsb = self.lib.nl_mv_aggregate_backward_workspace_bytes if train is None else self.lib.nl_mv_aggregate_backward_train_workspace_bytes
        ws = self._workspace(wsb(ct.byref(self.cfg), self.V, N))
        if workspace_samples is not None:
            ws = ws[: wsb(ct.byref(self.cfg), self.V, int(workspace_samples))]
        if train is not None:
            L.check(self.lib.nl_mv_aggregate_backward_train(ct.byref(self.cfg), self.packed.data_ptr(), self._frame, x.data_ptr(), N, g.data_ptr(), gx.data_ptr(),
                                                            ct.byref(train.c), ws.data_ptr(), ws.numel(), self._stream()), "nl_mv_aggregate_backward_train")
            return gx
        L.check(self.lib.nl_mv_aggregate_backward(ct.byref(self.cfg), self.packed.data_ptr(), self._frame, x.data_ptr(), N, g.data_ptr(), gx.data_ptr(),
                                                  ws.data_ptr(), ws.numel(), self._stream()), "nl_mv_aggregate_backward")
        return gx

    def blend(self, xyz, query_center, feature_agg):
        """Row a15 as a stage (nl_blend): per-sample colours (N,3) from the sample positions and feature_agg."""
        self._ready()
        x, fa = _dev_f32(xyz, self.device), _dev_f32(feature_agg, self.device)
        N = x.shape[0]
        qc = torch.as_tensor(query_center).detach().float().cpu().contiguous()
        out = torch.empty(N, 3, device=self.device)
        ws = self._workspace(self.lib.nl_blend_workspace_bytes(ct.byref(self.cfg), self.V, N))
        L.check(self.lib.nl_blend(ct.byref(self.cfg), self.packed.data_ptr(), self._frame, qc.data_ptr(), x.data_ptr(), fa.data_ptr(), N, out.data_ptr(),
                                  ws.data_ptr(), ws.numel(), self._stream()), "nl_blend")
        return out

    def blend_backward(self, xyz, query_center, feature_agg, g_rgb_s, want_g_query_center: bool = True, workspace_samples: Optional[int] = None,
                       train: "TrainGrads" = None):
        """nl_blend_backward: -> (g_xyz (N,3), g_feature_agg (N,W), g_query_center (3,) or None).  train: also ADD the gradients of rgb_blending_mlp, the
        decoders, the DepthFusionNet maps and the blend-projected feature maps into that TrainGrads (nl_blend_backward_train)."""
        self._ready()
        x, fa, g = _dev_f32(xyz, self.device), _dev_f32(feature_agg, self.device), _dev_f32(g_rgb_s, self.device)
        N = x.shape[0]
        qc = torch.as_tensor(query_center).detach().float().cpu().contiguous()
        gx = torch.empty(N, 3, device=self.device)
        gfa = torch.empty(N, self.W, device=self.device)
        gq = torch.empty(N, 3, device=self.device) if want_g_query_center else None
        wsb = self.lib.nl_blend_workspace_bytes if train is None else self.lib.nl_blend_backward_train_workspace_bytes
        ws = self._workspace(wsb(ct.byref(self.cfg), self.V, N))
        if workspace_samples is not None:
            ws = ws[: wsb(ct.byref(self.cfg), self.V, int(workspace_samples))]
        if train is not None:
            L.check(self.lib.nl_blend_backward_train(ct.byref(self.cfg), self.packed.data_ptr(), self._frame, qc.data_ptr(), x.data_ptr(), fa.data_ptr(), N,
                                                     g.data_ptr(), gx.data_ptr(), gfa.data_ptr(), _ptr(gq), ct.byref(train.c), ws.data_ptr(), ws.numel(),
                                                     self._stream()), "nl_blend_backward_train")
            return gx, gfa, (None if gq is None else gq.sum(0))
        L.check(self.lib.nl_blend_backward(ct.byref(self.cfg), self.packed.data_ptr(), self._frame, qc.data_ptr(), x.data_ptr(), fa.data_ptr(), N, g.data_ptr(),
                                           gx.data_ptr(), gfa.data_ptr(), _ptr(gq), ws.data_ptr(), ws.numel(), self._stream()), "nl_blend_backward")
        return gx, gfa, (None if gq is None else gq.sum(0))

    # ------------------------------------------------------------------ training: weight gradients
    def train_grads(self, names: Sequence[str], support_feature: bool = False, feat_maps: bool = False, vis_featmaps: bool = False,
                    blend_feat_maps: bool = False) -> "TrainGrads":
        """Zero-filled gradient buffers for the listed state_dict tensors (and the per-frame tensors asked for): hand the object to the
        `*_backward(..., train=...)` calls of one step, then read `.weights[name]` / `.support_feature` / `.feat_maps` (V,h,w,C) /
        `.vis_featmaps` (V,32,vh,vw) / `.blend_feat_maps` (V,h,w,32)."""
        if support_feature or feat_maps or vis_featmaps or blend_feat_maps:
            self._ready()
        elif not self._weights_loaded:
            raise RuntimeError("load_weights() first")
        return TrainGrads(self, names, support_feature, feat_maps, vis_featmaps, blend_feat_maps)

    def point_mlp_backward(self, xyz, direction, mv_feat, g_feature_agg, K: int = 8, knn=None, workspace_samples: Optional[int] = None, train: "TrainGrads" = None):
        """Input gradient of `point_mlp` (nl_point_mlp_backward): -> (g_xyz (N,3), g_direction (N,3) or None, g_mv_feat (N,W)).
        knn = (d2, idx) as `point_mlp` returned them for the same points: saves the second neighbour search.
        train: a TrainGrads — the call also ADDS the gradients of the stage's weights / the support features into it (nl_point_mlp_backward_train)."""
        self._ready()
        dev = self.device
        x, g, gy = _dev_f32(xyz, dev), _dev_f32(mv_feat, dev), _dev_f32(g_feature_agg, dev)
        N = x.shape[0]
        dr = None if direction is None else _dev_f32(direction, dev)
        gx = torch.empty(N, 3, device=dev)
        gd = None if dr is None else torch.empty(N, 3, device=dev)
        gg = torch.empty(N, self.W, device=dev)
        wsb = self.lib.nl_point_mlp_backward_workspace_bytes if train is None else self.lib.nl_point_mlp_backward_train_workspace_bytes
        ws = self._workspace(wsb(ct.byref(self.cfg), N))
        if workspace_samples is not None:   # (tests: a workspace for fewer samples than N makes the entry point walk N in chunks)
            ws = ws[: wsb(ct.byref(self.cfg), int(workspace_samples))]
        d2, idx = (None, None) if knn is None else (knn[0].contiguous(), knn[1].to(torch.int32).contiguous())
        if train is not None:
            L.check(self.lib.nl_point_mlp_backward_train(ct.byref(self.cfg), self.packed.data_ptr(), self._frame, x.data_ptr(), _ptr(dr), 0 if dr is None else dr.shape[1],
                                                         g.data_ptr(), N, K, _ptr(idx), _ptr(d2), gy.data_ptr(), gx.data_ptr(), _ptr(gd), gg.data_ptr(),
                                                         ct.byref(train.c), ws.data_ptr(), ws.numel(), self._stream()), "nl_point_mlp_backward_train")
            return gx, gd, gg
        L.check(self.lib.nl_point_mlp_backward(ct.byref(self.cfg), self.packed.data_ptr(), self._frame, x.data_ptr(), _ptr(dr), 0 if dr is None else dr.shape[1],
                                               g.data_ptr(), N, K, _ptr(idx), _ptr(d2), gy.data_ptr(), gx.data_ptr(), _ptr(gd), gg.data_ptr(), ws.data_ptr(),
                                               ws.numel(), self._stream()), "nl_point_mlp_backward")
        return gx, gd, gg

    def hierarchical_depths(self, pixel_coordinates, K, pose, z_base, u, n_coarse: int = 64, near=None, far=None, lindisp: bool = False):
        """a20: coarse NeuRay weights along the pixel rays -> inverse-CDF samples (uniforms `u` (R,Ni)) merged with
        z_base (R,Sb) and sorted.  Returns (z_vals (R,Sb+Ni), depth_coarse (R,), weights_coarse (R,n_coarse)).
        near / far: the RAYS' depth range (model.py:489 samples the coarse depths from rays['depth_range']) as python floats or as
        tensors on any device — tensors are used where they live, without a device-to-host copy; default: the frame's range.
        lindisp: the coarse row is sampled like ConditionalNeRF.sample_depths (model.py:451-458), linearly in depth or in disparity."""
        self._ready()
        dev = self.device
        pix = _dev_f32(pixel_coordinates, dev)
        zb = _dev_f32(z_base, dev)
        uu = _dev_f32(u, dev)
        R, Sb, Ni = pix.shape[0], zb.shape[1], uu.shape[1]
        Kc = torch.as_tensor(K).detach().float().cpu()
        Pc = torch.as_tensor(pose).detach().float().cpu()
        cam = torch.cat([Pc[None].inverse()[0, :3].reshape(-1), torch.inverse(Kc).reshape(-1)]).contiguous()  # depth_fusion.py:19-26
        near = self.near if near is None else near
        far = self.far if far is None else far

        def row(zn, zf, t_lin):   # the reference's own expression (model.py:451-458), evaluated where the operands live
            return zn * (1 - t_lin) + zf * t_lin if not lindisp else 1 / (1 / zn * (1 - t_lin) + 1 / zf * t_lin)
        if isinstance(near, torch.Tensor) or isinstance(far, torch.Tensor):
            zn = torch.as_tensor(near, dtype=torch.float32).detach().to(dev)
            zf = torch.as_tensor(far, dtype=torch.float32).detach().to(dev)
            t_dev = self._zc_rows.get(("t", n_coarse))
            if t_dev is None:   # linspace evaluated on the host (the goldens' arithmetic), uploaded once
                t_dev = self._zc_rows[("t", n_coarse)] = torch.linspace(0, 1, n_coarse).to(dev)
            zrow = row(zn, zf, t_dev)
        else:   # python floats: the row is formed once on the host like the reference forms it and kept on the device (a few entries, reset per frame)
            key = (n_coarse, float(near), float(far), bool(lindisp))
            zrow = self._zc_rows.get(key)
            if zrow is None:
                if len(self._zc_rows) >= 8:
                    self._zc_rows.clear()
                zrow = row(torch.tensor(float(near)), torch.tensor(float(far)), torch.linspace(0, 1, n_coarse)).to(dev)
                self._zc_rows[key] = zrow
        zc = zrow.expand(R, n_coarse).contiguous()
        wc = torch.empty(R, n_coarse, device=dev)
        dc = torch.empty(R, device=dev)
        zo = torch.empty(R, Sb + Ni, device=dev)
        ws = self._workspace(self.lib.nl_coarse_weights_workspace_bytes(self.V, R, n_coarse))
        L.check(self.lib.nl_coarse_weights(ct.byref(self.cfg), self.packed.data_ptr(), self._frame, cam.data_ptr(), pix.data_ptr(), zc.data_ptr(),
                                           R, n_coarse, wc.data_ptr(), dc.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()), "nl_coarse_weights")
        L.check(self.lib.nl_sample_pdf(zc.data_ptr(), wc.data_ptr(), n_coarse, uu.data_ptr(), Ni, zb.data_ptr(), Sb, R, zo.data_ptr(), self._stream()),
                "nl_sample_pdf")
        return zo, dc, wc

    def ray_unet(self, x):
        """x (R*S, W) sample-major -> geo (R*S, W)."""
        if not self._weights_loaded:
            raise RuntimeError("load_weights() first")
        xin = _dev_f32(x, self.device)
        R = xin.shape[0] // self.S
        geo = torch.empty_like(xin)
        ws = self._workspace(self.lib.nl_ray_unet_workspace_bytes(ct.byref(self.cfg), R))
        L.check(self.lib.nl_ray_unet(ct.byref(self.cfg), self.packed.data_ptr(), xin.data_ptr(), R, geo.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()), "nl_ray_unet")
        return geo


def render_rays_multi(jobs):
    """Several frames — each a HipRenderer holding its own support set — in ONE library call (nl_render_rays_multi): job =
    (renderer, rays_o, rays_d, query_center (3,) or (R, 3)[, kwargs: z_vals, white_bkgd, want_feat, early_term_eps]).  The renderers must be of one
    configuration and hold the same weights (the first one's packed blob serves all).  Outputs are bit-identical to `renderer.render_rays(...)` per job;
    the launch chains of the jobs run on library-owned streams forked from / joined into the current stream."""
    jobs = list(jobs)
    if not jobs:
        return []
    r0 = jobs[0][0]
    dev, lib = r0.device, r0.lib
    arr = (L.NlRenderJob * len(jobs))()
    keep, outs = [], []
    white = None
    for i, job in enumerate(jobs):
        r, o, d, qc = job[:4]
        kw = dict(job[4]) if len(job) > 4 else {}
        if (r.W, r.C, r.S, r.precision) != (r0.W, r0.C, r0.S, r0.precision):
            raise ValueError("render_rays_multi: the renderers must share one configuration")
        r._ready()
        w = bool(kw.get("white_bkgd", False))
        if white is None:
            white = w
        elif white != w:
            raise ValueError("render_rays_multi: one white_bkgd for all jobs")
        o, d = _dev_f32(o, dev), _dev_f32(d, dev)
        R = o.shape[0]
        z = kw.get("z_vals")
        z = None if z is None else _dev_f32(z, dev)
        qc_t = torch.as_tensor(qc).detach().float()
        per_ray = qc_t.dim() == 2 or qc_t.is_cuda
        if per_ray:
            qc_t = qc_t.reshape(-1, 3).expand(R, 3).to(dev).contiguous()
        else:
            qc_t = qc_t.cpu().contiguous()
        out = {"rgb": torch.empty(R, 3, device=dev), "depth": torch.empty(R, device=dev), "weights": torch.empty(R, r.S, device=dev),
               "mask": torch.empty(R, dtype=torch.uint8, device=dev), "depth_uncertainty": torch.empty(R, device=dev)}
        if kw.get("want_feat", True):
            out["feat"] = torch.empty(R, r.C, device=dev)
        ro = L.NlRenderOut()
        for k, t in out.items():
            setattr(ro, k, t.data_ptr())
        ws = r._workspace(r._ws_request or lib.nl_render_rays_workspace_bytes(ct.byref(r.cfg), r.V, R))
        opts = L.NlRenderOpts()
        opts.early_term_eps = float(kw.get("early_term_eps", 0.0))
        if per_ray:
            opts.ray_centers = qc_t.data_ptr()
        a = arr[i]
        a.frame, a.query_center = r._frame, (None if per_ray else qc_t.data_ptr())
        a.rays_o, a.rays_d, a.z_vals, a.R = o.data_ptr(), d.data_ptr(), _ptr(z), R
        a.out, a.ws, a.ws_bytes, a.opts = ct.pointer(ro), ws.data_ptr(), ws.numel(), ct.pointer(opts)
        keep.append((o, d, z, qc_t, ro, opts, ws))
        outs.append(out)
    L.check(lib.nl_render_rays_multi(ct.byref(r0.cfg), r0.packed.data_ptr(), arr, len(jobs), int(bool(white)), r0._stream()), "nl_render_rays_multi")
    cur = torch.cuda.current_stream(dev)
    for tensors in keep:   # (the library's streams read these: the caching allocator must not recycle them before the current stream passed the join)
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(cur)
    for out in outs:
        out["mask"] = out["mask"].view(torch.bool)
    return outs


def render_rays_concurrent(jobs, streams=None):
    """Render several (renderer, rays_o, rays_d, query_center[, kwargs]) jobs — typically query frames with DIFFERENT support sets, one HipRenderer
    (frame tables, workspace, side stream) each — on separate HIP streams, so that the launch chains of small per-frame batches fill the chip together
    (8 frames x 512 rays at config 2: 12.9 ms one after the other, 10.7 ms concurrently; tools/multi_frame_bench.py).  Results are
    bit-identical to rendering the jobs one by one.  Returns the list of output dicts; the current stream waits for all of them."""
    jobs = list(jobs)
    if not jobs:
        return []
    dev = jobs[0][0].device
    cur = torch.cuda.current_stream(dev)
    if streams is None:
        streams = [torch.cuda.Stream(dev) for _ in jobs]
    outs = []
    for job, s in zip(jobs, streams):
        r, o, d, qc = job[:4]
        kw = job[4] if len(job) > 4 else {}
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(r.render_rays(o, d, qc, **kw))
    for s in streams[:len(jobs)]:
        cur.wait_stream(s)
    for out in outs:   # the caching allocator must not hand these buffers out again before the side streams are done with them
        for t in out.values():
            t.record_stream(cur)
    return outs
