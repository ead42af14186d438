#!/usr/bin/env python3
"""bench.py — rendered rays/s of the HIP conditional-NeRF renderer on synthetic BASELINE.json configs.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2] [--precision bf16x3]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one `render_rays` pass (rows a2-a18 of SURVEY.md §8; for a hierarchical config also row a20: the coarse NeuRay
pass + inverse-CDF resampling) over one batch of R rays whose inputs are already resident in HBM; per-frame setup (weight
packing, KNN grid, map repack) is outside the timed region, like the reference's cached `support_neural_points` /
`vis_featmaps`.  With N>1 the per-ray outputs are joined by ONE RCCL all-gather inside the timed step, and
  --scaling strong (the default for every config since round 4: BASELINE's "rays/sec at 1/2/4/8 GPUs" asks how much faster ONE batch gets):
                   ONE R-ray batch, rank r renders shard_range(R, r, N); value = R / step;
  --scaling weak   every rank renders its own R-ray batch of the same frame; value = N * R / step (true by construction up to the all-gather).
With --scaling auto (default) an N>1 run times BOTH: `value` / `ms_per_step` / `scaling` are the STRONG numbers, `weak_scaling` carries the weak
ones in the same line (`scaling_modes: "both"`), and `allgather` the measured latency of the step's one collective on its own.
Rank 0 prints one JSON line.

`roofline`: bound = MFMA; achieved = ALGORITHMIC flops of one step (SURVEY.md §8(d) formula: GEMM/conv MACs x2,
q-projection counted once) / mean step duration from HIP events on the launch stream; peak = 2.5 PFLOP/s dense bf16
(MI355X_MICROARCH.md).  `cpu_baseline`: the CPU oracle (oracle/, a port of the reference's PyTorch path, "kind":"port")
timed on rank 0's host cores over a bounded ray sample of the same workload.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0


def algorithmic_mac_per_sample(W: int, V: int, C: int = 192, K: int = 8) -> float:
    """forward GEMM/conv multiply-adds per sample (reference formulation, q-proj once)."""
    F = C + 3
    point = K * (496 + (F + 90) * W + 2 * W * W + 2 * 128 * W + 128 * W + W * W + W) + 128 * W
    attn = 16384  # 8 query rows x 8 keys x 128 dims x {qk, av} as the reference evaluates it
    mv = (2 * F + 3) * 64 + 64 * W + V * 8384
    unet = 192 * W + 12288 + 12288 + 6144 + 12288 + 6144 + 3 * (W + 32) * W
    heads = W + W * W + C * W + V * ((W + F + 5) * 32 + 528)
    return float(point + attn + mv + unet + heads)


def executed_mfma_equiv_mac_per_sample(W: int, V: int, S: int, precision: str, C: int = 192, K: int = 8):
    """MFMA-equivalent multiply-adds the kernels EXECUTE per sample in a precision mode (one bf16 / fp16 `32x32x16` MAC = 1; a three-term split product = 3; an
    f16mx product = 1.5: fp16 hi.hi + two MX-FP6 cross terms at four times the rate — round 5; MX-FP8 at twice the rate = 2.0 before): (whole step, fused neural-point kernel).  What the kernels multiply differs from
    the algorithmic count of SURVEY 8(d): the 195 feature columns of base_mlp.0 and rgb_blending_mlp.0 come from per-frame tables, the attention's q-projection and
    `fc` run once per sample, out_fc.2 is recomputed in both chain kernels, feat_mlp.2 runs per ray.  algorithmic / executed-equivalent = the fraction
    of the bf16 MFMA peak `roofline.frac` would show with the matrix pipe 100 % busy: the CEILING of the parity mode."""
    per = {"bf16": 1.0, "bf16x3": 3.0, "f16mx": 3.0, "fp32": 16.0}[precision]     # every GEMM but the fused kernel's wide layers
    wide = {"f16mx": 1.5}.get(precision, per)                                      # layers 2, 3 and the k / v projections of point_fused2_kernel
    l1 = {"f16mx": 10.0 / 6.0}.get(precision, per)                                 # layer 1 of point_fused2_kernel in f16mx: 6 f16 + 4 fp6 matrix instructions for its 6 k-steps (slab 1 is half empty)
    point = K * (96 * W * l1 + (2 * W * W + 256 * W) * wide)                        # layer 1 (K = 96: posenc + ray_diff_fc) + layers 2, 3 + k / v
    conv_out = 3 * (W + 32) * W
    unet = 192 * W + 12288 + 12288 + 6144 + 12288 + 6144   # the other six convolutions per sample (pooled levels, transposed = 1.5 taps per output)
    feat0 = W * W                                           # feat_mlp.0
    other = (384 * 64 + 2 * 64 * W + W * 128 + 128 * W + W * 32 + V * (4 * 2 * 32 * 32 + 6 * 32) + unet + C * W / S)
    # round 6: conv_out multiplies in the f16mx arithmetic too (tgemm_mx_kernel: W = 256, S = 128), and so does feat_mlp.0, fused with the compositing of its rows
    # (feat_comp_mx_kernel: W = 256, rays of 32 .. 256 samples in whole 32-row tiles)
    conv_out_per = 1.5 if (precision == "f16mx" and W == 256 and S == 128) else per
    feat0_per = 1.5 if (precision == "f16mx" and W == 256 and S % 32 == 0 and S <= 256 and ((8 % (S // 32)) == 0 or (6 % (S // 32)) == 0)) else per
    return point + other * per + conv_out * conv_out_per + feat0 * feat0_per, point


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2")
    ap.add_argument("--precision", default="f16mx", choices=["fp32", "bf16x3", "bf16", "f16mx"],
                    help="f16mx (default since round 4): the parity mode with 1.5 (round 5: MX-FP6 cross terms; round 4: 2.0, MX-FP8) instead of 3 MFMA-equivalents per product in the fused neural-point kernel")
    ap.add_argument("--rays", type=int, default=0, help="override rays per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-thread-sweep", action="store_true", help="cpu_baseline: also time 64 rays at 8 and 64 threads (off by default: the driver's run should spend "
                                                                    "its wall time on the GPU part, VERDICT r4 item 8)")
    ap.add_argument("--no-gradient-step", action="store_true", help="skip the (untimed for the headline) PoseOptimizer-step measurement")
    ap.add_argument("--force-gather", action="store_true", help="run the N>1 collective path on one GPU (single-rank RCCL group): a functional check")
    ap.add_argument("--also", default="bf16x3,bf16,fp32", help="comma list of extra precisions timed after the headline (''=none)")
    ap.add_argument("--early-term-eps", type=float, default=-1.0,
                    help="early-termination compositing threshold (nl_render_opts); default: 1e-5 for c5 (BASELINE names it there), 0 = off otherwise")
    ap.add_argument("--graph", action="store_true", help="replay batches of <= 1024 rays as a HIP graph (A/B against the eager launch chain; measured: no gain)")
    ap.add_argument("--no-side-stream", action="store_true", help="NL_RENDER_NO_SIDE_STREAM: every kernel on one stream (profiling kernels one at a time)")
    ap.add_argument("--sustained-seconds", type=float, default=10.0,
                    help="after the headline: this many seconds of back-to-back steps of the same workload -> `sustained` (0 = skip; N = 1 only)")
    ap.add_argument("--no-parity-gate", action="store_true", help="report `parity` but do not fail the run (rc 3) when it is above 1e-4")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="N>1: weak = R rays per rank, strong = one R-ray batch sharded over the ranks (auto: BOTH are timed, value = strong)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (the launcher's own
        # form of this command line is what the driver uses; both end in the same code below)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    # NERFLOC_BENCH_ONE_GPU=1: functional check of the N>1 control flow on a box with one GPU — every rank uses device 0 and the
    # collective runs over gloo (RCCL refuses two ranks on one device).  Never used for a reported number.
    one_gpu = os.environ.get("NERFLOC_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    gather = world > 1 or args.force_gather   # --force-gather: exercise the collective path on one GPU (single-rank RCCL group)
    if gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        elif one_gpu:
            dist.init_process_group("gloo")
        else:
            if torch.cuda.device_count() < world:
                raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} HIP devices are visible (NERFLOC_BENCH_ONE_GPU=1 runs the "
                                 "control flow on one device over gloo: a functional check, never a reported number)")
            dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")

    from nerf_loc_amd.renderer import HipRenderer
    from nerf_loc_amd.sharding import ShardedRenderLoop, gather_ray_outputs_async, shard_range
    from nerf_loc_amd.synth import make_u

    cfg = CONFIGS[args.config]
    R = args.rays or cfg.R            # weak: rays per rank; strong: rays of the one batch
    S = cfg.S_total
    both = args.scaling == "auto" and world > 1          # N > 1 by default: the strong pass is the headline, the weak pass rides in the same line
    scaling = args.scaling if args.scaling != "auto" else "strong"
    frame = make_frame(cfg)

    def make_batch(mode):
        """Host-side rays of this rank for one scaling mode -> (rays, uniforms, rays on this rank, rays per rank or None)."""
        if mode == "strong" and world > 1:
            rr = make_rays(cfg, frame, R=R, seed_offset=1000)   # the SAME batch on every rank ...
            lo, hi = shard_range(R, rank, world)                 # ... of which this rank renders a contiguous range
            cnt = [shard_range(R, r, world)[1] - shard_range(R, r, world)[0] for r in range(world)]
            rr = {k: (v[lo:hi] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == R else v) for k, v in rr.items()}
            return rr, (make_u(cfg, R)[lo:hi] if cfg.N_importance > 0 else None), hi - lo, cnt
        rr = make_rays(cfg, frame, R=R, seed_offset=1000 + rank)  # every rank its own batch of pixels
        return rr, (make_u(cfg, R) if cfg.N_importance > 0 else None), R, None
    rays, u_all, R_local, counts = make_batch(scaling)
    weights = make_weights(cfg)

    rnd = HipRenderer(cfg.W, cfg.C, S, args.precision, device=f"cuda:{local_rank}")
    rnd.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
    rnd.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"],
                  cfg.near, cfg.far, frame["support_fine"])
    Sb = cfg.S                        # base samples (model.py:483-484); hierarchical configs add N_importance resampled depths
    t_lin = torch.linspace(0, 1, Sb)
    z_row = torch.tensor(cfg.near, dtype=torch.float32) * (1 - t_lin) + torch.tensor(cfg.far, dtype=torch.float32) * t_lin  # model.py:451-458
    qc = frame["pose"][:3, 3]
    hier = cfg.N_importance > 0
    et_eps = args.early_term_eps if args.early_term_eps >= 0 else (1e-5 if args.config == "c5" else 0.0)

    def to_device(rr, uu, n_local, cnt):
        b = {"o": torch.from_numpy(rr["rays_o"]).to(dev), "d": torch.from_numpy(rr["rays_d"]).to(dev), "z": z_row.expand(n_local, Sb).contiguous().to(dev),
             "counts": cnt, "R_local": n_local}
        if hier:   # model.py:487-496: coarse NeuRay weights along the pixel rays -> sample_pdf with FIXED uniforms -> sort(cat)
            b.update({"pix": torch.from_numpy(rr["pixel_coordinates"]).to(dev), "u": torch.from_numpy(uu).to(dev),
                      "Kq": torch.from_numpy(rr["K"]), "pose_q": torch.from_numpy(rr["pose"])})
        return b
    B = to_device(rays, u_all, R_local, counts)   # the batch `step()` renders (swapped for the weak pass of an N > 1 run)

    # N > 1: every step ends with ONE all-gather of the packed per-ray outputs — the product's own sharded step (nerf_loc_amd/sharding.py: ShardedRenderLoop,
    # the pipelined form of render_rays_sharded): the gather is started asynchronously (RCCL's stream) and collected one step later, so the xGMI transfer of
    # batch i overlaps the kernels of batch i + 1; drain() collects the last one inside the timed region, so K timed steps = K renders + K completed gathers.
    loop = [None]
    last_out = [None]

    def render_local(out_buffers=None):
        zz = B["z"]
        if hier:
            zz, depth_coarse, _ = rnd.hierarchical_depths(B["pix"], B["Kq"], B["pose_q"], B["z"], B["u"], near=cfg.near, far=cfg.far)
        # --graph: batches of <= 1024 rays replay as a HIP graph from the second step on (HipRenderer.GRAPH_MAX_RAYS).  Opt-in: measured, the replay is no faster
        # than the eager launch chain (config 1: 0.397 ms eager, 0.423 ms with the graph's static-buffer copies; a 512-ray shard of config 2: 1.400 / 1.412 ms)
        out = rnd.render_rays(B["o"], B["d"], qc, z_vals=zz, white_bkgd=cfg.white_bkgd, early_term_eps=et_eps, side_stream=not args.no_side_stream,
                              graph=args.graph, out_buffers=out_buffers)
        if hier:
            out["depth_coarse"] = depth_coarse
        return out

    def step():
        if gather:
            if loop[0] is None or loop[0].counts is not B["counts"]:
                loop[0] = ShardedRenderLoop(dist, B["counts"])
            # (plain configs: the kernels write into ONE buffer per rank and that buffer is gathered — no pack step; a hierarchical config carries `depth_coarse`
            # from outside the render call and takes the generic packed-dict path)
            out = loop[0].step(render_local) if (hier or args.graph) else loop[0].step_packed(rnd, B["R_local"], render_local)
            last_out[0] = loop[0].last_local
            return out
        out = render_local()
        last_out[0] = out
        return out

    def drain():
        return loop[0].drain() if loop[0] is not None else None

    def timed(steps, warmup):
        for _ in range(warmup):
            step()
        drain()
        # the interpreter's cyclic collector stays out of the timed region, as in timeit: a generation-2 pass is 40-50 ms of host time with the GPU idle
        # (one landed in the gradient bench's five timed steps whenever its allocation counter happened to cross the threshold there)
        gc.collect()
        gc.disable()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            step()
        drain()
        ev1.record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        wall = time.perf_counter() - t0
        gc.enable()
        dev_ms = ev0.elapsed_time(ev1)
        if world > 1:
            t = torch.tensor([wall, dev_ms], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall, dev_ms = float(t[0]), float(t[1])
        return wall, dev_ms

    wall, dev_ms = timed(args.steps, args.warmup)
    ms_per_step = wall * 1e3 / args.steps
    # the outputs of the LAST timed step of this rank (the batch is the same every step): what `parity` below compares with the CPU oracle
    headline_out = {k: v.clone() for k, v in last_out[0].items()}

    # dominant kernel (fused neural-point kernel): HIP events around each of its launches, on the
    # stream it is launched on, over a second pass of the same steps (the events themselves are outside `value`)
    import ctypes as ct
    from nerf_loc_amd import _lib
    lib = _lib.load()
    lib.nl_profile_begin()
    for _ in range(args.steps):
        step()
    drain()
    fused_ms, launches = ct.c_float(0), ct.c_int(0)
    _lib.check(lib.nl_profile_end(ct.byref(fused_ms), ct.byref(launches)), "nl_profile_end")
    total_rays = R if (scaling == "strong" and world > 1) else world * R     # rays all ranks rendered per step
    value = total_rays * args.steps / wall
    # algorithmic flops of what THIS rank computes per step (+ the coarse pass of a hierarchical config: 64 x V x 8384 MAC per ray)
    flops_step = 2.0 * algorithmic_mac_per_sample(cfg.W, cfg.V, cfg.C) * R_local * S + (2.0 * 64 * cfg.V * 8384 * R_local if hier else 0.0)
    ach = flops_step / (dev_ms * 1e-3 / args.steps) / 1e12
    result = {
        "metric": baseline_metric(), "value": value, "unit": "rays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": {"bf16x3": "bf16x3 (3-term split-bf16 MFMA, fp32 accumulate; meets 1e-4)",
                                                            "f16mx": "f16mx (neural-point kernel, conv_out and feat_mlp.0: fp16 hi.hi + two MX-FP6 cross terms, 1.5 MFMA-equivalents per product; every other GEMM "
                                                                     "3-term split-bf16 or split-fp16; fp32 accumulate; meets 1e-4)",
                                                            "bf16": "bf16", "fp32": "f32"}[args.precision],
        "data": "synthetic",
        "config": {"workload": f"{cfg.name}: {total_rays} rays x {S} samples" + (f" (64 coarse + {cfg.S} + {cfg.N_importance} resampled)" if hier else "")
                               + f", W={cfg.W}, V={cfg.V} views {cfg.H}x{cfg.Wimg}, M={frame['support_fine']['xyz'].shape[0]} neural points",
                   "rays_per_gpu": R_local, "precision": args.precision,
                   "early_term_eps": et_eps,   # random-init weights give a thin medium: nothing terminates early, the option only costs its two tiny kernels
                   "parallelism": f"ray-shard x{dist.get_world_size() if dist is not None else 1} ({scaling})"
                                  + (f" + {dist.get_backend()} all-gather" if gather else ""),
                   "collective_world_size": dist.get_world_size() if dist is not None else 1,
                   "collective_backend": (("rccl" if dist.get_backend() == "nccl" else dist.get_backend()) if dist is not None else None)},
        "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                     "traffic": hbm_traffic(args.config, args.precision), "scope": "whole render_rays step (all kernels), algorithmic flops SURVEY §8(d)",
                     "flops_per_step": flops_step, "device_ms_per_step": dev_ms / args.steps},
    }
    ex_all, ex_point = executed_mfma_equiv_mac_per_sample(cfg.W, cfg.V, S, args.precision, cfg.C)
    diag = rnd.diagnostics()
    result["roofline"]["parity_mode_ceiling"] = {
        "whole_step": algorithmic_mac_per_sample(cfg.W, cfg.V, cfg.C) / ex_all,
        "what": "algorithmic MAC / MFMA-equivalent MAC the kernels execute in this precision mode (3 per split-bf16 product, 1.5 per f16mx product): "
                "`frac` with the matrix pipe 100 % busy at the 2.4 GHz the peak assumes"}
    result["roofline"]["clock_GHz_under_load"] = {"point_fused2_kernel": diag["point_kernel_GHz"], "peak_assumes": 2.4,
                                                  "how": "s_memtime cycles / s_memrealtime of workgroup 0 over the last launch (nl_frame_diagnostics)"}
    result["conditioning"] = {"attention_logit_absmax": diag["logit_absmax"], "table_absmax": diag["table_absmax"],
                              "note": "conditioning indicator of DESIGN.md 2.3: the parity modes are validated to 1e-4 of the CPU oracle up to max |attention logit| = 100 (f16mx) / 500 (bf16x3) "
                                      "(tools/scale_sweep.py, profiles/r5_scale_sweep.txt); the drop-in module escalates the precision beyond that"}
    if launches.value > 0:
        K, F, W = 8, cfg.C + 3, cfg.W
        result["roofline"]["parity_mode_ceiling"]["dominant_kernel"] = K * (496 + (F + 90) * W + 2 * W * W + 2 * 128 * W + 16384 / K) / ex_point
        mac_alg = K * (496 + (F + 90) * W + 2 * W * W + 2 * 128 * W) + 16384      # a9 ray_diff_fc, a10 base_mlp, k/v projection, a11 attention (SURVEY §8d terms)
        mac_exec = K * (96 * W + 2 * W * W + 256 * W)                               # what the kernel multiplies (feature columns come from the per-frame table T)
        samples_per_launch = R_local * S * args.steps / launches.value
        sec = fused_ms.value * 1e-3 / launches.value
        alg = 2.0 * mac_alg * samples_per_launch / sec / 1e12
        result["roofline"]["dominant_kernel"] = {
            "name": "point_fused2_kernel", "launches": launches.value, "avg_ms": fused_ms.value / launches.value,
            "share_of_step": fused_ms.value / args.steps / (dev_ms / args.steps),
            "achieved": alg, "frac": alg / PEAK_BF16_TFLOPS, "unit": "TFLOP/s (algorithmic, SURVEY §8d)",
            "executed_mfma_TFLOPs": 2.0 * (K * 96 * W * {"bf16x3": 3.0, "f16mx": 10.0 / 6.0}.get(args.precision, 1.0) + (mac_exec - K * 96 * W) * {"bf16x3": 3.0, "f16mx": 1.5}.get(args.precision, 1.0))
                                    * samples_per_launch / sec / 1e12,   # (f16mx, layer 1: 10 matrix instructions of 8 passes for 6 k-steps)
        }

    if gather:
        # the step's one collective on its own: K blocking all-gathers of the last step's packed per-ray outputs (pack + RCCL + unpack), nothing else running —
        # the latency a step would pay if it could not overlap it with the next batch's kernels (it can: gather_ray_outputs_async)
        out_l = last_out[0]
        for _ in range(2):
            gather_ray_outputs_async(out_l, dist, B["counts"]).result()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            gather_ray_outputs_async(out_l, dist, B["counts"]).result()
        torch.cuda.synchronize()
        ag = torch.tensor([(time.perf_counter() - t0) * 1e3 / args.steps], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(ag, op=dist.ReduceOp.MAX)
        from nerf_loc_amd.sharding import pack_outputs
        nb = pack_outputs(out_l)[0]
        result["allgather"] = {"ms_per_call_blocking": float(ag[0]), "bytes_per_rank": int(nb.numel() * 4), "bytes_total": int(nb.numel() * 4 * world),
                               "note": "pack + all_gather_into_tensor + unpack of one step's per-ray outputs, alone on the device; inside the timed step it runs "
                                       "asynchronously beside the next step's kernels"}
    if both:
        # the weak-scaling pass of the same run (every rank its own R-ray batch): reported beside the strong headline, never as `value`
        rw, uw, nw, cw = make_batch("weak")
        B = to_device(rw, uw, nw, cw)
        wall_w, dev_ms_w = timed(args.steps, args.warmup)
        result["weak_scaling"] = {"value": world * R * args.steps / wall_w, "unit": "rays/s", "ms_per_step": wall_w * 1e3 / args.steps,
                                  "rays_per_gpu": R, "device_ms_per_step": dev_ms_w / args.steps,
                                  "note": "every rank renders its own R-ray batch of the same frame + the all-gather: N x the one-GPU rate up to the collective"}
        result["scaling_modes"] = "both"

    if args.also and world == 1:
        extra = {}
        for p in [x for x in args.also.split(",") if x and x != args.precision]:
            rnd.set_precision(p)
            w2, d2 = timed(max(3, args.steps // 2), 2)
            n2 = max(3, args.steps // 2)
            extra[p] = {"rays_per_s": R * n2 / w2, "ms_per_step": w2 * 1e3 / n2, "roofline_frac": flops_step / (d2 * 1e-3 / n2) / 1e12 / PEAK_BF16_TFLOPS,
                        "note": {"bf16": "single bf16 MFMA per product: throughput mode, does NOT meet 1e-4", "fp32": "f32-input MFMA, generic kernels: strictest parity mode",
                                 "bf16x3": "parity mode (3-term split-bf16 everywhere)", "f16mx": "parity mode, fp16 + MX-FP6 cross terms in the neural-point kernel"}[p]}
        rnd.set_precision(args.precision)
        result["other_precisions"] = extra

    if world == 1 and args.sustained_seconds > 0:
        result["sustained"] = sustained(step, drain, rnd, R, args.sustained_seconds, ms_per_step)

    if rank == 0 and world == 1 and not args.no_gradient_step and not hier and args.precision != "fp32":
        try:   # (an extra next to the headline: whatever happens here must not cost the line)
            result["gradient_step"] = gradient_step(rnd, cfg, frame, rays, weights, torch.device(f"cuda:{local_rank}"))
        except Exception as e:   # noqa: BLE001
            result["gradient_step"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"], result["parity"] = cpu_baseline(cfg, frame, rays, weights, u_all, thread_sweep=args.cpu_thread_sweep, gpu_out=headline_out,
                                                                precision=args.precision)
    # RCCL prints its version banner (NCCL_DEBUG=VERSION) through C stdio, which a pipe flushes only at exit: every rank pushes it out
    # now, then rank 0 prints the JSON line as the last thing on stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if gather:
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)
        par = result.get("parity")
        if par is not None and not par["ok"] and not args.no_parity_gate:
            print(f"bench.py: parity of the timed batch against the CPU oracle is above {par['bar']:g}: {par['max_rel']}", file=sys.stderr)
            raise SystemExit(3)


def gradient_step(rnd, cfg, frame, rays, weights, dev):
    """Not the headline: one PoseOptimizer step (pose_optimizer.py:131-160: 512 rays, feature loss, forward + backward to the 4x4 pose) of the same
    scene through the library's gradient path (diff_render.RenderFn: nl_render_rays_forward_keep / nl_render_rays_backward_kept, replayed as two HIP graphs from
    the second step on), after the timed region of the headline."""
    from nerf_loc_amd import diff_render as dr
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    Rg = min(512, cfg.R)
    p = {k: t(v) for k, v in weights.items()}
    fr = {k: t(frame[k]) for k in ("topk_Ks", "topk_poses", "topk_images", "feat_fine_src", "vis_featmaps")}
    fr.update({"near": float(cfg.near), "far": float(cfg.far), "support": {k: t(v) for k, v in frame["support_fine"].items()}})
    sel = np.random.default_rng(0).choice(len(rays["pixel_coordinates"]), Rg, replace=False)
    uv, K = t(rays["pixel_coordinates"][sel]), t(rays["K"])
    lin = torch.linspace(0, 1, cfg.S, device=dev)
    z = (cfg.near * (1 - lin) + cfg.far * lin).expand(Rg, cfg.S).contiguous()
    pose = t(frame["pose"]).clone().requires_grad_(True)
    tf = torch.randn(Rg, cfg.C, generator=torch.Generator().manual_seed(0)).to(dev)

    def one():
        o, d = dr.rays_from_pose(uv, K, pose)
        out = dr.render_rays_diff(p, fr, o, d, z, pose, lambda q: rnd.knn(q, 8)[1], frozen_renderer=rnd)
        loss = torch.mean(((out["feat"] - tf) * out["mask"].unsqueeze(1)) ** 2)
        return torch.autograd.grad(loss, pose)[0]
    for _ in range(2):
        one()
    gc.collect()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        g = one()
    torch.cuda.synchronize(dev)
    return {"what": f"PoseOptimizer step: {Rg} rays x {cfg.S} samples, forward + backward to the pose, frozen weights", "ms_per_step": (time.perf_counter() - t0) / n * 1e3,
            "grad_finite": bool(torch.isfinite(g).all()), "path": "RenderFn: nl_render_rays_forward_keep + nl_render_rays_backward_kept (two HIP graphs from the second step on)"}


def sustained(step, drain, rnd, R, seconds, burst_ms):
    """>= `seconds` of back-to-back steps of the headline workload (same batch, same mode), in blocks of 64 steps with one synchronisation per block (the host
    runs ahead inside a block like in the timed region).  The headline is a 0.1-s burst; on a part whose dominant kernel is power-limited the
    steady state may sit lower — this says by how much.  Clock: nl_frame_diagnostics' shader clock of the fused neural-point kernel's last launch, read after the
    first and after the last block."""
    block = 64
    for _ in range(3):
        step()
    drain()
    torch.cuda.synchronize()
    ghz_first = ghz_last = None
    blocks = []
    gc.collect()
    gc.disable()
    t_start = time.perf_counter()
    n = 0
    while True:
        t0 = time.perf_counter()
        for _ in range(block):
            step()
        drain()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n += block
        blocks.append((t1 - t0) * 1e3 / block)
        if ghz_first is None:
            ghz_first = rnd.diagnostics()["point_kernel_GHz"]
        if t1 - t_start >= seconds:
            break
    total = time.perf_counter() - t_start
    gc.enable()
    ghz_last = rnd.diagnostics()["point_kernel_GHz"]
    rate = R * n / total
    return {"seconds": total, "steps": n, "rays_per_s": rate, "ms_per_step": total * 1e3 / n, "ms_per_step_first_block": blocks[0], "ms_per_step_last_block": blocks[-1],
            "ms_per_step_worst_block": max(blocks), "clock_GHz_first": ghz_first, "clock_GHz_last": ghz_last, "vs_burst": (burst_ms / (total * 1e3 / n)),
            "note": "back-to-back steps of the headline workload after the headline's timed burst (64-step blocks, one sync + one 12-byte diagnostics read per block, "
                    "both inside the time); `value` stays the burst the contract defines (exactly K steps); vs_burst = sustained rate / burst rate"}


def baseline_metric() -> str:
    """The headline metric exactly as BASELINE.json names it (falls back to the same wording if the file is not there)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as fh:
            return json.load(fh)["metric"]
    except (OSError, KeyError, ValueError):
        return "rendered rays/sec (4096 rays×128 samples, 256-wide MLP) at 1/2/4/8 MI355X"


def sources_sha() -> str:
    """Fingerprint of the kernel sources a measurement belongs to (the GPU box has no .git)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "nerf_loc_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "nerf_loc_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def hbm_traffic(config: str, precision: str):
    """HBM GB per step from rocprofv3 PMC passes of this same command (FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs
    by tools/hbm_traffic.py; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md).  A constant read from a committed
    file, NOT measured in this run: it carries the fingerprint of the kernel sources it was measured on and says whether that is
    the build being benchmarked.  None when no committed measurement matches the workload."""
    for name in ("r6_hbm_traffic.json", "r5_hbm_traffic.json", "r4_hbm_traffic.json", "r3_hbm_traffic.json", "r2_hbm_traffic.json", "r1_hbm_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if config != "c2" or not os.path.exists(path):
            continue
        d = json.load(open(path))
        if d.get("precision", "bf16x3") != precision:
            continue
        sha = d.get("sources_sha")
        return {"GB_per_step": d["fetch_GB_per_step_x2_gfx950_correction"] + d["write_GB_per_step"], "write_GB_per_step": d["write_GB_per_step"],
                "source": f"profiles/{name} (separate rocprofv3 --pmc passes, not this run)", "measured_on_sources": sha,
                "matches_this_build": (sha == sources_sha()) if sha else False}
    return None


def cpu_baseline(cfg, frame, rays, weights, u_all=None, budget_s: float = 12.0, thread_sweep: bool = False, gpu_out=None, precision: str = ""):
    """Time the CPU oracle (port of the reference's PyTorch path) on a bounded ray sample of the same workload.

    Threads: torch intra-op parallelism saturates around 8-64 threads on this path and collapses beyond (measured on the
    256-core GPU box: 75 rays/s at 8..64 threads, 36 at 128, <1 at 256), so min(cores, 32) threads are used and reported.
    Rays are processed in 256-ray chunks (a c2 chunk peaks around 15 GB), like the reference's `render.chunk` loop."""
    from oracle import render_oracle as orc
    cores = os.cpu_count() or 1
    threads = min(cores, 32)
    torch.set_num_threads(threads)
    p = {k: torch.from_numpy(v) for k, v in weights.items()}
    fr = orc.to_torch(frame)

    keep = None

    def run(lo, hi, timers=None, nthreads=threads):
        sub = {k: (torch.from_numpy(v[lo:hi]) if k in ("rays_o", "rays_d", "pixel_coordinates") else (torch.from_numpy(v) if isinstance(v, np.ndarray) else v))
               for k, v in rays.items()}
        uu = None if u_all is None else torch.from_numpy(u_all[lo:hi])   # fixed uniforms for sample_pdf (reference: torch.rand)
        t0 = time.perf_counter()
        with torch.no_grad():
            ref = orc.render_rays(p, fr, sub, cfg.S, cfg.N_importance, u=uu, knn_threads=nthreads, timers=timers)
        dt = time.perf_counter() - t0
        if keep is not None:
            keep.append((lo, hi, {k: ref[k].numpy() for k in PARITY_KEYS + ("mask", "z_vals") if k in ref}))
        return dt

    run(0, 16)                          # warm-up (first-touch, thread pool)
    t = run(0, 64)
    total = int(max(64, min(len(rays["rays_o"]), 64 * budget_s / max(t, 1e-3))))
    total = min(total - total % 64, 1024)
    timers, t_all, done = {}, 0.0, 0
    keep = [] if gpu_out is not None else None   # (run() appends the oracle's outputs of the timed chunks from here on)
    while done < total:
        n = min(256, total - done)
        t_all += run(done, done + n, timers)
        done += n
    # the same path at other thread counts (64 rays each): intra-op parallelism saturates early on this path
    sweep = {}
    for nt in ((8, 64) if thread_sweep else ()):
        if nt <= cores and nt != threads:
            torch.set_num_threads(nt)
            run(0, 16, nthreads=nt)
            sweep[str(nt)] = round(64 / run(0, 64, nthreads=nt), 1)
    kept, keep = keep, None
    torch.set_num_threads(threads)
    sweep[str(threads)] = round(total / t_all, 1)
    parity = parity_block(cfg, frame, rays, kept, gpu_out, precision) if gpu_out is not None else None
    return {"value": total / t_all, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"{total} rays x {cfg.S_total} samples of the same workload in 256-ray chunks, {t_all:.1f} s, {threads} of {cores} host cores",
            "rays_per_s_by_threads": sweep,
            "stage_seconds": {k: round(v, 3) for k, v in timers.items()}}, parity


PARITY_KEYS = ("rgb", "depth", "weights", "depth_uncertainty", "feat")
PARITY_BAR = 1e-4


def parity_block(cfg, frame, rays, kept, gpu_out, precision):
    """Parity of THE RUN THAT WAS TIMED: the rows of the headline batch the CPU oracle rendered for `cpu_baseline` against the same rows of the
    GPU's last timed step.  max_rel = max |gpu - oracle| / max |oracle| per output over the compared rays (the metric of every parity test, north_star's "1e-4
    rel"), l2_rel = ||gpu - oracle|| / ||oracle||; mask_equal = the valid-ray mask bit for bit.  Rays with a sample within 1e-3 pixel of a support view's image
    border are hard-threshold cases of the reference's in-image test (two correct fp32 evaluations may differ: synth.borderline_rays): they are compared and
    counted separately (`borderline`), the gate is on the others."""
    from nerf_loc_amd.synth import borderline_rays
    rows = np.concatenate([np.arange(lo, hi) for lo, hi, _ in kept])
    ref = {k: np.concatenate([o[k] for _, _, o in kept], 0) for k in kept[0][2]}
    got = {k: gpu_out[k][torch.from_numpy(rows).to(gpu_out[k].device)].cpu().numpy() for k in PARITY_KEYS + ("mask",)}
    border = borderline_rays(cfg, frame, rays["rays_o"][rows], rays["rays_d"][rows], ref["z_vals"])
    core = ~border

    def rel(a, b):
        a, b = a.astype(np.float64), b.astype(np.float64)
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)) if a.size else 0.0

    def l2(a, b):
        a, b = a.astype(np.float64), b.astype(np.float64)
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)) if a.size else 0.0
    max_rel = {k: rel(got[k][core], ref[k][core]) for k in PARITY_KEYS}
    l2_rel = {k: l2(got[k][core], ref[k][core]) for k in PARITY_KEYS}
    mask_equal = bool(np.array_equal(got["mask"][core].astype(bool), ref["mask"][core].astype(bool)))
    b_rel = {}
    if border.any():   # (denominator: the whole sample's max |oracle|, so that the two groups are on one scale)
        b_rel = {k: float(np.abs(got[k][border].astype(np.float64) - ref[k][border].astype(np.float64)).max() / max(np.abs(ref[k].astype(np.float64)).max(), 1e-30))
                 for k in PARITY_KEYS}
    ok = mask_equal and max(max_rel.values()) < PARITY_BAR and max(l2_rel.values()) < PARITY_BAR
    return {"rays": int(core.sum()), "of": f"rows {int(rows[0])}..{int(rows[-1])} of the timed batch (the cpu_baseline sample)", "precision": precision, "bar": PARITY_BAR,
            "max_rel": max_rel, "l2_rel": l2_rel, "mask_equal": mask_equal, "ok": bool(ok),
            "borderline": {"rays": int(border.sum()), "max_rel": b_rel,
                           "mask_equal": bool(np.array_equal(got["mask"][border].astype(bool), ref["mask"][border].astype(bool))),
                           "note": "rays with a sample within 1e-3 px of a support view's image border (hard thresholds of the reference's in-image test): reported, not gated"}}


if __name__ == "__main__":
    main()
