"""Ray-range sharding across the GPUs of one node.

Rays are independent (the only coupling — ray U-Net, compositing — is along one ray), so a batch is cut
into contiguous ray ranges, one per rank, per-frame state is replicated, and the per-ray outputs are joined
by ONE all-gather (RCCL over xGMI on the GPU box; `gloo` in the CPU tests).  The payload is tiny
(<= 1.6 kB per ray), so the collective is latency-bound: all outputs are packed into a single fp32 buffer.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch


def shard_range(n_rays: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range [lo, hi) of rank `rank`; the first n_rays % world ranks get one extra ray."""
    base, rem = divmod(n_rays, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_ORDER = ("rgb", "depth", "depth_uncertainty", "mask", "feat", "weights", "depth_coarse")


def pack_outputs(out: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, list]:
    """Concatenate the per-ray outputs into one (R, D) fp32 buffer; returns (buffer, layout)."""
    cols, layout = [], []
    for k in _ORDER:
        if k in out:
            t = out[k]
            t2 = t.reshape(t.shape[0], -1).to(torch.float32)
            layout.append((k, t2.shape[1], t.dtype, tuple(t.shape[1:])))
            cols.append(t2)
    return torch.cat(cols, 1).contiguous(), layout


def unpack_outputs(buf: torch.Tensor, layout: list) -> Dict[str, torch.Tensor]:
    out, c = {}, 0
    for k, n, dt, shp in layout:
        t = buf[:, c:c + n]
        c += n
        t = t.reshape(buf.shape[0], *shp)
        out[k] = (t > 0.5) if dt == torch.bool else t.to(dt)
    return out


class PendingGather:
    """An all-gather in flight (RCCL runs it on its own stream): `result()` makes the current stream wait for it and unpacks."""

    def __init__(self, work, full, buf, layout, counts, even):
        self.work, self.full, self.buf, self.layout, self.counts, self.even = work, full, buf, layout, counts, even

    def result(self) -> Dict[str, torch.Tensor]:
        if self.work is not None:
            self.work.wait()
            self.work = None
        full = self.full
        if not self.even:
            rows = self.buf.shape[0]
            full = torch.cat([full[r * rows: r * rows + c] for r, c in enumerate(self.counts)], 0)
        return unpack_outputs(full, self.layout)


def gather_ray_outputs_async(out: Dict[str, torch.Tensor], dist, counts=None) -> PendingGather:
    """Start the all-gather of the per-ray outputs of every rank (rank order) and return at once, so that the caller can launch the
    next batch's kernels while the collective runs — xGMI transfers and compute overlap on separate streams.  `counts` = rays per
    rank when uneven: shards are zero-padded to the largest one so a single fixed-size collective serves both backends."""
    world = dist.get_world_size()
    buf, layout = pack_outputs(out)
    even = counts is None or len(set(counts)) == 1
    if not even:
        pad = max(counts) - buf.shape[0]
        if pad:
            buf = torch.cat([buf, buf.new_zeros(pad, buf.shape[1])], 0)
    full = torch.empty(world * buf.shape[0], buf.shape[1], dtype=buf.dtype, device=buf.device)
    if buf.is_cuda and dist.get_backend() == "nccl":
        work = dist.all_gather_into_tensor(full, buf, async_op=True)
    else:
        work = dist.all_gather(list(full.chunk(world, 0)), buf, async_op=True)
    return PendingGather(work, full, buf, layout, counts, even)


def gather_ray_outputs(out: Dict[str, torch.Tensor], dist, counts=None) -> Dict[str, torch.Tensor]:
    """All-gather per-ray outputs of every rank, in rank order (blocking form of gather_ray_outputs_async)."""
    return gather_ray_outputs_async(out, dist, counts).result()
