"""Development check of the gradient path's fused keep-forward (pt_forward_keep_fused) against the staged one (pt_forward_staged): the k / v rows, the three
layers' sign bits, the attention output and the gradients, read back from the caller's workspace.  Needs a library built with -DNERFLOC_DEBUG_SWITCHES
(NERFLOC_LIB=nerf_loc_amd/csrc/variants/libnerfloc_dbg.so).  python tools/keep_check.py [case]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loc_amd.renderer import HipRenderer
from tests.golden_cases import build_case

case = sys.argv[1] if len(sys.argv) > 1 else "w256s128"
c = build_case(case)
cfg, frame, rays = c["cfg"], c["frame"], c["rays"]
dev = torch.device("cuda:0")
r = HipRenderer(cfg.W, cfg.C, cfg.S_total, "bf16x3")
r.load_weights({k: torch.from_numpy(v) for k, v in c["weights"].items()})
r.set_frame(frame["topk_images"], frame["feat_fine_src"], frame["vis_featmaps"], frame["topk_Ks"], frame["topk_poses"], cfg.near, cfg.far, frame["support_fine"])
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
R = min(cfg.R, 12)
o, d = t(rays["rays_o"][:R]), t(rays["rays_d"][:R])
lin = torch.linspace(0, 1, cfg.S, device=dev)
z = (cfg.near * (1 - lin) + cfg.far * lin).expand(R, cfg.S)
xyz = (o[:, None, :] + d[:, None, :] * z[..., None]).reshape(-1, 3).contiguous()
dirs = d[:, None, :].expand(R, cfg.S, 3).reshape(-1, 3).contiguous()
g = torch.Generator().manual_seed(3)
G = torch.randn(xyz.shape[0], cfg.W, generator=g).to(dev)
cot = torch.randn(xyz.shape[0], cfg.W, generator=g).to(dev)
N, W, K = xyz.shape[0], cfg.W, 8
NK = N * K
ldx = (cfg.C + 3 + 90 + 31) // 32 * 32

def run(staged, chain=False, att=False):
    for k, on in (("NERFLOC_NO_KEEP_FUSED", staged), ("NERFLOC_NO_BWD_CHAIN", not chain), ("NERFLOC_NO_BWD_ATT", not att)):
        if on: os.environ[k] = "1"
        else: os.environ.pop(k, None)
    r._ws = None
    out = r.point_mlp_backward(xyz, dirs, G, cot)
    torch.cuda.synchronize()
    return [v.cpu().numpy() for v in out], r._ws.cpu().numpy().copy()

(gs, ws_s), (gf, ws_f) = run(True), run(False)
gc, ws_c = run(True, chain=True)   # the same staged forward; the four row products of the way back as one launch (point_bwd.hip)
for n, a, b in zip(("g_xyz", "g_dir", "g_G"), gs, gc):
    print(n, "chained way back vs staged rel", float(np.abs(a - b).max() / np.abs(a).max()))
ga, ws_a = run(True, chain=True, att=True)   # ... with the attention's way back inside the chain kernel's prologue
for n, a, b in zip(("g_xyz", "g_dir", "g_G"), gs, ga):
    print(n, "chained + attention way back vs staged rel", float(np.abs(a - b).max() / np.abs(a).max()))
for n, a, b in zip(("g_xyz", "g_dir", "g_G"), gs, gf):
    print(n, "fused vs staged rel", float(np.abs(a - b).max() / np.abs(a).max()))
# carve_ptb's order (abi.hip): every take is 256-byte aligned
off = 0
def take(nbytes):
    global off
    o_ = off
    off = (off + nbytes + 255) // 256 * 256
    return o_
names = [("idx", NK * 4), ("d2", NK * 4), ("X", NK * ldx * 4), ("H1", NK * W * 4), ("H2", NK * W * 4), ("H3", NK * W * 4), ("KV", NK * 256 * 4), ("Q", N * 128 * 4),
         ("O", N * 128 * 4), ("FCo", N * W * 4), ("wscale", N * 4), ("gpre", N * W * 4), ("gO", N * 128 * 4), ("gQ", N * 128 * 4), ("gKV", NK * 256 * 4),
         ("gA", NK * W * 4), ("gB", NK * W * 4), ("gX", NK * 96 * 4), ("mk0", (NK // 32 + 8) * 1024), ("mk1", (NK // 32 + 8) * 1024), ("mk2", (NK // 32 + 8) * 1024)]
for nm, nb in names:
    o_ = take(nb)
    if nm.startswith("mk"):
        a, b = ws_s[o_:o_ + NK // 32 * 1024].view(np.uint32), ws_f[o_:o_ + NK // 32 * 1024].view(np.uint32)
        x = a ^ b
        bits = int(sum(bin(int(v)).count("1") for v in x[x != 0]))
        print(nm, "differing sign bits", bits, "of", NK * W, "| words differing", int((x != 0).sum()), "first", np.nonzero(x)[0][:6])
    elif nm == "gX":
        a, b = ws_s[o_:o_ + nb].view(np.float32).reshape(NK, 96), ws_c[o_:o_ + nb].view(np.float32).reshape(NK, 96)
        err = np.abs(a - b).max(1) / np.abs(a).max()
        print("gX chained vs staged: max diff / max", float(err.max()), "| rows beyond 1e-4:", int((err > 1e-4).sum()), "of", NK, "| L2-rel", float(np.linalg.norm(a - b) / np.linalg.norm(a)),
              "| first bad rows", np.nonzero(err > 1e-4)[0][:8], "| pad columns max", float(np.abs(b[:, 90:]).max()))
    elif nm == "gQ":
        a, b = ws_s[o_:o_ + nb].view(np.float32), ws_a[o_:o_ + nb].view(np.float32)
        print("gQ attention-in-chain vs staged: max diff / max", float(np.abs(a - b).max() / np.abs(a).max()))
    elif nm in ("KV", "Q", "O", "FCo", "wscale", "idx"):
        dt = np.int32 if nm == "idx" else np.float32
        a, b = ws_s[o_:o_ + nb].view(dt), ws_f[o_:o_ + nb].view(dt)
        if nm == "KV":
            a2, b2 = a.reshape(NK, 256), b.reshape(NK, 256)
            print("K  max diff", float(np.abs(a2[:, :128] - b2[:, :128]).max()), "scale", float(np.abs(a2[:, :128]).max()),
                  "| V max diff", float(np.abs(a2[:, 128:] - b2[:, 128:]).max()), "scale", float(np.abs(a2[:, 128:]).max()))
            np.set_printoptions(precision=4, linewidth=220, suppress=True)
            for rr in (0, 1, 9):
                print("   staged k row", rr, a2[rr, :16]); print("   fused  k row", rr, b2[rr, :16])
                print("   staged v row", rr, a2[rr, 128:144]); print("   fused  v row", rr, b2[rr, 128:144])
            # is the fused row a permutation of the staged one?
            for rr in (0,):
                perm = [int(np.argmin(np.abs(a2[rr, :128] - v))) for v in b2[rr, :32]]
                print("   fused k[0, c] equals staged k[0, perm[c]]:", perm)
                perm = [int(np.argmin(np.abs(a2[:32, 128] - v))) for v in b2[:32, 128]]
                print("   fused v[r, 0] equals staged v[perm[r], 0]:", perm)
            bad = np.nonzero(np.abs(a2 - b2).max(1) > 1e-3 * np.abs(a2).max())[0]
            print("   rows off:", bad.size, bad[:10], "cols off of first:", np.nonzero(np.abs(a2[bad[0]] - b2[bad[0]]) > 1e-3)[0][:12] if bad.size else "")
        else:
            print(nm, "max diff", float(np.abs(a.astype(np.float64) - b).max()), "scale", float(np.abs(a).max()))
