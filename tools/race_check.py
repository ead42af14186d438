import sys, torch, numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_configs import _scene, _renderer, _render, KEYS
sc = _scene("c3")
r = _renderer(sc, "bf16x3")
a = _render(r, sc); torch.cuda.synchronize()
for it in range(3):
    b = _render(r, sc); torch.cuda.synchronize()
    print({k: (float((a[k] - b[k]).abs().max()), int((a[k] != b[k]).reshape(a[k].shape[0], -1).any(1).sum())) for k in KEYS})
    bad = (a["rgb"] != b["rgb"]).any(1).nonzero().flatten()
    print("bad rays", bad[:20].tolist(), "of", a["rgb"].shape[0])
