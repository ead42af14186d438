#!/usr/bin/env python3
"""Does the ORDER of the rays in a batch matter?  Renders the same config-2 batch (random pixels, as sample_rays draws them) in three
orders — as drawn, row-major by pixel, Morton order of the pixel — and prints ms/step for each.  (Rays are independent: the results
are the same rows in another order; what changes is which texels / octree leaves / table rows meet in an XCD's L2.)"""
import sys, time
import numpy as np
import torch

sys.path.insert(0, ".")


def morton(px):
    def part(v):
        v = v & 0xffff; v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555
        return v
    return part(px[:, 0].astype(np.uint32)) | (part(px[:, 1].astype(np.uint32)) << 1)


def main():
    from tests.test_gpu_configs import _renderer, _scene, _zbase
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    sc = _scene(name); cfg = sc["cfg"]; r = _renderer(sc, "bf16x3")
    o, d, pix = sc["rays"]["rays_o"], sc["rays"]["rays_d"], sc["rays"]["pixel_coordinates"]
    z = _zbase(cfg, cfg.R); qc = sc["frame"]["pose"][:3, 3]
    orders = {"as drawn (random pixels)": np.arange(cfg.R), "row-major": np.lexsort((pix[:, 0], pix[:, 1])), "morton": np.argsort(morton(pix))}
    for nm, perm in orders.items():
        oo, dd = torch.from_numpy(o[perm]).cuda(), torch.from_numpy(d[perm]).cuda()
        for _ in range(3): r.render_rays(oo, dd, qc, z_vals=z)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): r.render_rays(oo, dd, qc, z_vals=z)
        torch.cuda.synchronize(); print(f"{name} {nm}: {(time.perf_counter() - t0) * 100:.3f} ms/step", flush=True)


if __name__ == "__main__":
    main()
