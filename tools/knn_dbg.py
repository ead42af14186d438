import sys, numpy as np, torch, time
sys.path.insert(0, '/root/repo')
from tests.golden_cases import build_case
from tests.test_gpu_parity import _renderer
from oracle import render_oracle as orc
rng = np.random.default_rng(0)
for (n, m, K) in [(2000, 3000, 8), (2000, 5, 8), (3000, 700, 1)]:
    p = rng.standard_normal((m, 3)).astype(np.float32)
    q = (rng.standard_normal((n, 3)) * 1.5).astype(np.float32)
    case = build_case("tiny_full")
    case["frame"]["support_fine"] = {"xyz": p, "feature": np.zeros((m, 195), np.float32), "confidence": np.ones((m, 1), np.float32), "direction": np.zeros((m, 4), np.float32)}
    r = _renderer(case, "fp32")
    d2, idx = r.knn(q, K)
    torch.cuda.synchronize()
    d2o, idxo = orc.knn_points(torch.from_numpy(q), torch.from_numpy(p), K)
    a, b = d2.cpu().numpy(), d2o.numpy()
    bad = np.where((a != b).any(1))[0]
    print(n, m, K, "mismatch rows", len(bad), "idx mismatch", int((idx.cpu().numpy() != idxo.numpy()).any(1).sum()))
    if len(bad):
        i = bad[0]; print(" q", q[i], "\n got", a[i], idx.cpu().numpy()[i], "\n exp", b[i], idxo.numpy()[i])
