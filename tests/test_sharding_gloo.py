"""N>1 path on CPU: contiguous ray-range sharding + the single all-gather, world_size 2 over gloo."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerf_loc_amd.sharding import gather_ray_outputs, gather_ray_outputs_async, pack_outputs, shard_range, unpack_outputs


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 4096, 4097):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pack_roundtrip():
    R = 5
    out = {"rgb": torch.rand(R, 3), "depth": torch.rand(R), "weights": torch.rand(R, 16), "mask": torch.rand(R) > 0.5,
           "depth_uncertainty": torch.rand(R), "feat": torch.rand(R, 192)}
    buf, layout = pack_outputs(out)
    assert buf.shape == (R, 3 + 1 + 1 + 1 + 192 + 16)
    back = unpack_outputs(buf, layout)
    for k in out:
        assert torch.equal(back[k], out[k]), k


def _worker(rank, world, port, uneven, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R = 11 if uneven else 12
    g = torch.Generator().manual_seed(0)
    full = {"rgb": torch.rand(R, 3, generator=g), "depth": torch.rand(R, generator=g), "weights": torch.rand(R, 8, generator=g),
            "mask": torch.rand(R, generator=g) > 0.5, "depth_uncertainty": torch.rand(R, generator=g), "feat": torch.rand(R, 4, generator=g)}
    lo, hi = shard_range(R, rank, world)
    mine = {k: v[lo:hi].clone() for k, v in full.items()}
    counts = [shard_range(R, r, world)[1] - shard_range(R, r, world)[0] for r in range(world)]
    got = gather_ray_outputs(mine, dist, counts if uneven else None)
    ok = all(torch.equal(got[k], full[k]) for k in full)
    # pipelined form used by bench.py: two gathers in flight order, collected one step late
    first = gather_ray_outputs_async(mine, dist, counts if uneven else None)
    mine2 = {k: (~v if v.dtype == torch.bool else v * 2) for k, v in mine.items()}
    second = gather_ray_outputs_async(mine2, dist, counts if uneven else None)
    got1, got2 = first.result(), second.result()
    ok = ok and all(torch.equal(got1[k], full[k]) for k in full)
    ok = ok and all(torch.equal(got2[k], (~full[k] if full[k].dtype == torch.bool else full[k] * 2)) for k in full)
    q.put((rank, ok))
    dist.destroy_process_group()


def _run(uneven, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, uneven, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_two_rank_gather_even():
    _run(False, 29611)


def test_two_rank_gather_uneven():
    _run(True, 29612)


def _strong_worker(rank, world, port, q):
    """bench.py --scaling strong: every rank builds the SAME batch, renders shard_range(R, rank, world) of it, one gather joins them."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R = 37   # uneven over 2 ranks
    g = torch.Generator().manual_seed(3)
    rays_o, rays_d = torch.rand(R, 3, generator=g), torch.rand(R, 3, generator=g)

    def fake_render(o, d):   # any per-ray function: a ray's result must not depend on the batch it is rendered in
        return {"rgb": o * d, "depth": (o * d).sum(1), "weights": torch.stack([o[:, 0], d[:, 1], o[:, 2] * d[:, 0]], 1),
                "mask": o[:, 0] > d[:, 0], "depth_uncertainty": o.sum(1), "feat": torch.cat([o, d], 1)}
    lo, hi = shard_range(R, rank, world)
    counts = [shard_range(R, r, world)[1] - shard_range(R, r, world)[0] for r in range(world)]
    got = gather_ray_outputs(fake_render(rays_o[lo:hi], rays_d[lo:hi]), dist, counts)
    full = fake_render(rays_o, rays_d)
    q.put((rank, all(torch.equal(got[k], full[k]) for k in full) and sum(counts) == R))
    dist.destroy_process_group()


def test_strong_scaling_split_reassembles_the_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_strong_worker, args=(r, 2, 29561, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


class _FakeRenderer:
    """Stands in for HipRenderer on the CPU: any per-ray function with render_rays' signature and output keys."""

    def render_rays(self, o, d, qc, z_vals=None, white_bkgd=False):
        qc = torch.as_tensor(qc, dtype=o.dtype)
        c = qc if qc.dim() == 2 else qc.expand(o.shape[0], 3)
        return {"rgb": o * d + c, "depth": (o * d).sum(1) + (0 if z_vals is None else z_vals.sum(1)), "weights": torch.stack([o[:, 0], d[:, 1]], 1),
                "mask": o[:, 0] > d[:, 0], "depth_uncertainty": o.sum(1), "feat": torch.cat([o, d], 1), "knn_idx": torch.zeros(o.shape[0], 8)}


class _FakePackedRenderer(_FakeRenderer):
    """... and one that writes into preallocated buffers like HipRenderer.render_rays(out_buffers=...) (the packed gather path)"""
    supports_out_buffers = True
    S, C, device = 2, 6, torch.device("cpu")

    def render_rays(self, o, d, qc, z_vals=None, white_bkgd=False, out_buffers=None, want_feat=True):
        out = _FakeRenderer.render_rays(self, o, d, qc, z_vals, white_bkgd)
        out.pop("knn_idx")
        if out_buffers is None:
            return out
        for k, v in out.items():
            out_buffers[k].copy_(v.to(torch.uint8) if k == "mask" else v)
        return {k: (v.view(torch.bool) if k == "mask" else v) for k, v in out_buffers.items()}


def _product_worker(rank, world, port, q):
    """render_rays_sharded / ShardedRenderLoop / shard_rays (the product-side sharded step) reassemble the single-rank result exactly;
    a rank with ZERO rays (R < world) takes part in the collective like any other."""
    from nerf_loc_amd.sharding import ShardedRenderLoop, render_rays_sharded, shard_counts, shard_rays
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    r = _FakeRenderer()
    for R in (37, 12, 1):
        g = torch.Generator().manual_seed(R)
        o, d, z = torch.rand(R, 3, generator=g), torch.rand(R, 3, generator=g), torch.rand(R, 5, generator=g)
        for qc in (torch.tensor([1., 2., 3.]), torch.rand(R, 3, generator=g)):
            full = r.render_rays(o, d, qc, z_vals=z)
            got = render_rays_sharded(r, o, d, qc, dist, z_vals=z)
            ok = ok and "knn_idx" not in got and all(torch.equal(got[k], full[k]) for k in got) and set(got) == set(full) - {"knn_idx"}
            pend = render_rays_sharded(r, o, d, qc, dist, z_vals=z, async_op=True)
            ok = ok and all(torch.equal(v, full[k]) for k, v in pend.result().items())
        cnt = shard_counts(R, world)
        lo, hi = shard_range(R, rank, world)
        loop = ShardedRenderLoop(dist, None if len(set(cnt)) == 1 else cnt)
        qc = torch.tensor([0.5, 0., 1.])
        first = loop.step(lambda: r.render_rays(o[lo:hi], d[lo:hi], qc))
        second = loop.step(lambda: r.render_rays(d[lo:hi], o[lo:hi], qc))
        last = loop.drain()
        want1, want2 = r.render_rays(o, d, qc), r.render_rays(d, o, qc)
        ok = ok and first is None and all(torch.equal(second[k], want1[k]) and torch.equal(last[k], want2[k]) for k in second) and loop.drain() is None
        # the packed path: the renderer writes into ONE buffer per rank, the buffer is gathered (even, uneven and empty shards)
        rp = _FakePackedRenderer()
        full_p = rp.render_rays(o, d, qc, z_vals=z)
        got_p = render_rays_sharded(rp, o, d, qc, dist, z_vals=z)
        ok = ok and set(got_p) == set(full_p) and all(got_p[k].dtype == full_p[k].dtype and torch.equal(got_p[k], full_p[k]) for k in full_p)
        lp = ShardedRenderLoop(dist, None if len(set(cnt)) == 1 else cnt)
        f1 = lp.step_packed(rp, hi - lo, lambda ob: rp.render_rays(o[lo:hi], d[lo:hi], qc, z_vals=z[lo:hi], out_buffers=ob))
        f2 = lp.step_packed(rp, hi - lo, lambda ob: rp.render_rays(d[lo:hi], o[lo:hi], qc, z_vals=z[lo:hi], out_buffers=ob))
        f3 = lp.drain()
        w1, w2 = rp.render_rays(o, d, qc, z_vals=z), rp.render_rays(d, o, qc, z_vals=z)
        ok = ok and f1 is None and all(torch.equal(f2[k], w1[k]) and torch.equal(f3[k], w2[k]) for k in w1)
        rays = {"rays_o": o, "rays_d": d, "pixel_coordinates": torch.rand(R, 2, generator=g), "K": torch.eye(3), "pose": torch.eye(4), "H": 3, "W": 3}
        sub = shard_rays(rays, rank, world)
        ok = ok and sub["rays_o"].shape[0] == hi - lo and torch.equal(sub["pixel_coordinates"], rays["pixel_coordinates"][lo:hi]) and sub["K"] is rays["K"] \
            and sub["pose"] is rays["pose"]   # (a 3x3 K with R = 3 rays must not be sliced: only the per-ray keys are)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_product_sharded_render_reassembles_single_rank_result():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_product_worker, args=(r, 2, 29563, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res
