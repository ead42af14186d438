"""The sharded render on the GPU box (run with `pytest -m gpu`): two ranks with the real `HipRenderer`, RCCL when two devices are visible,
otherwise both ranks on device 0 over gloo (RCCL refuses two ranks on one device).  The gathered outputs of `render_rays_sharded`
(nerf_loc_amd/sharding.py) must equal the single-rank render of the same batch BIT FOR BIT — on BASELINE config 1 and on two 512-ray
shards of config 2 — and so must the pipelined loop bench.py times and the module-level `render_image_sharded`.
(CPU counterpart: tests/test_sharding_gloo.py.)"""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
KEYS = ("rgb", "depth", "weights", "depth_uncertainty", "feat", "mask")


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _scene(name, R):
    from nerf_loc_amd.synth import CONFIGS, make_frame, make_rays, make_weights
    cfg = CONFIGS[name]
    frame = make_frame(cfg)
    return cfg, frame, make_rays(cfg, frame, R=R, seed_offset=7), make_weights(cfg)


def _renderer(cfg, frame, weights, dev):
    from nerf_loc_amd.renderer import HipRenderer
    r = HipRenderer(cfg.W, cfg.C, cfg.S_total, "f16mx", device=str(dev))
    r.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
    r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far,
                frame["support_fine"])
    return r


def _rank(rank, world, port, two_devices, q):
    try:
        import torch.distributed as dist
        from nerf_loc_amd.sharding import ShardedRenderLoop, render_rays_sharded, shard_counts, shard_range
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dev = torch.device("cuda", rank if two_devices else 0)
        torch.cuda.set_device(dev)
        if two_devices:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        res = {"backend": dist.get_backend()}
        for name, R in (("c1", 256), ("c2", 1024), ("c2", 1023)):     # c1 whole; c2: two 512-ray shards; an uneven cut (512 + 511)
            cfg, frame, rays, weights = _scene(name, R)
            r = _renderer(cfg, frame, weights, dev)
            o, d = torch.from_numpy(rays["rays_o"]).to(dev), torch.from_numpy(rays["rays_d"]).to(dev)
            lin = torch.linspace(0, 1, cfg.S)
            z = (cfg.near * (1 - lin) + cfg.far * lin).expand(R, cfg.S).contiguous().to(dev)
            qc = frame["pose"][:3, 3]
            single = r.render_rays(o, d, qc, z_vals=z, white_bkgd=cfg.white_bkgd)
            got = render_rays_sharded(r, o, d, qc, dist, z_vals=z, white_bkgd=cfg.white_bkgd)
            ok = all(got[k].shape == single[k].shape and torch.equal(got[k], single[k]) for k in KEYS)
            # the pipelined loop (bench.py --gpus N): two steps in flight order, the second batch = the first reversed
            lo, hi = shard_range(R, rank, world)
            cnt = shard_counts(R, world)
            loop = ShardedRenderLoop(dist, None if len(set(cnt)) == 1 else cnt)
            o2, d2 = o.flip(0).contiguous(), d.flip(0).contiguous()
            first = loop.step(lambda: r.render_rays(o[lo:hi], d[lo:hi], qc, z_vals=z[lo:hi], white_bkgd=cfg.white_bkgd))
            a = loop.step(lambda: r.render_rays(o2[lo:hi], d2[lo:hi], qc, z_vals=z[lo:hi], white_bkgd=cfg.white_bkgd))
            b = loop.drain()
            ok_loop = first is None and all(torch.equal(a[k], single[k]) and torch.equal(b[k], single[k].flip(0)) for k in KEYS)
            # ... and its packed form (what bench.py --gpus N times for the plain configs): the kernels write into the buffer that is gathered
            lp = ShardedRenderLoop(dist, None if len(set(cnt)) == 1 else cnt)
            p1 = lp.step_packed(r, hi - lo, lambda ob: r.render_rays(o[lo:hi], d[lo:hi], qc, z_vals=z[lo:hi], white_bkgd=cfg.white_bkgd, out_buffers=ob))
            p2 = lp.step_packed(r, hi - lo, lambda ob: r.render_rays(o2[lo:hi], d2[lo:hi], qc, z_vals=z[lo:hi], white_bkgd=cfg.white_bkgd, out_buffers=ob))
            p3 = lp.drain()
            ok_loop = ok_loop and p1 is None and all(torch.equal(p2[k], single[k]) and torch.equal(p3[k], single[k].flip(0)) for k in KEYS)
            res[f"{name}/{R}"] = (bool(ok), bool(ok_loop), int(hi - lo))
            del r
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, res, None))
    except Exception as e:   # noqa: BLE001
        import traceback
        q.put((rank, None, f"{type(e).__name__}: {e}\n{traceback.format_exc()[-1500:]}"))


def _spawn(target, extra=()):
    import torch.multiprocessing as mp
    two = torch.cuda.device_count() >= 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=target, args=(r, 2, port, two, q) + tuple(extra)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=900) for _ in ps]
    for p in ps:
        p.join(timeout=120)
    return sorted(res, key=lambda t: t[0])


def test_two_rank_sharded_render_is_bit_identical_to_the_single_rank_render():
    res = _spawn(_rank)
    for rank, r, err in res:
        assert err is None, f"rank {rank}: {err}"
    for rank, r, _ in res:
        assert r["backend"] in ("nccl", "gloo")
        for case in ("c1/256", "c2/1024", "c2/1023"):
            ok, ok_loop, n_local = r[case]
            assert ok, (rank, case, "gathered != single-rank")
            assert ok_loop, (rank, case, "pipelined loop != single-rank")
    assert res[0][1]["c2/1024"][2] == 512 and res[1][1]["c2/1024"][2] == 512
    assert res[0][1]["c2/1023"][2] == 512 and res[1][1]["c2/1023"][2] == 511
    print("backend:", res[0][1]["backend"])


def _rank_module(rank, world, port, two_devices, q):
    try:
        import torch.distributed as dist
        from tests.golden_cases import build_setup_case
        from tests.test_dropin_module import _module_and_data
        from tests.util import load_golden
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dev = torch.device("cuda", rank if two_devices else 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl" if two_devices else "gloo", rank=rank, world_size=world, **({"device_id": dev} if two_devices else {}))
        net, data, _ = _module_and_data(build_setup_case("setup"), dev, "f16mx")
        # the per-frame CNN runs on MIOpen, whose algorithm choice may differ between processes in the last bits: the reference's own maps are
        # injected (as in stage A of the end-to-end test) so that both ranks render from identical tables and the comparison can be bitwise
        net.support_neural_points = None
        net.multiview_aggregator.vis_featmaps = torch.from_numpy(load_golden("setup")["vis_featmaps"]).to(dev)
        with torch.no_grad():
            single = net.render_image(data)
            got = net.render_image_sharded(data, dist)
        ok = set(got) == set(single) and all(got[k].shape == single[k].shape and torch.equal(got[k], single[k]) for k in single)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, {"ok": bool(ok), "keys": sorted(single), "hw": tuple(single["rgb"].shape)}, None))
    except Exception as e:   # noqa: BLE001
        import traceback
        q.put((rank, None, f"{type(e).__name__}: {e}\n{traceback.format_exc()[-1500:]}"))


def test_two_rank_render_image_sharded_equals_render_image():
    res = _spawn(_rank_module)
    for rank, r, err in res:
        assert err is None, f"rank {rank}: {err}"
        assert r["ok"], (rank, r)
