import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_configs import _scene, _renderer, _zbase
sc = _scene("c2"); cfg = sc["cfg"]; r = _renderer(sc, "bf16x3")
o, d = sc["rays"]["rays_o"], sc["rays"]["rays_d"]; pix = sc["rays"]["pixel_coordinates"]
z = _zbase(cfg, cfg.R); qc = sc["frame"]["pose"][:3, 3]
def morton(px):
    x = (px[:, 0]).astype(np.uint32); y = (px[:, 1]).astype(np.uint32)
    def part(v):
        v = v & 0xffff; v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555; return v
    return part(x) | (part(y) << 1)
orders = {"random": np.arange(cfg.R), "row-major": np.lexsort((pix[:, 0], pix[:, 1])), "morton": np.argsort(morton(pix))}
for name, perm in orders.items():
    oo, dd = torch.from_numpy(o[perm]).cuda(), torch.from_numpy(d[perm]).cuda()
    for _ in range(3): r.render_rays(oo, dd, qc, z_vals=z)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): r.render_rays(oo, dd, qc, z_vals=z)
    torch.cuda.synchronize(); print(name, round((time.perf_counter() - t0) * 100, 3), "ms/step")
