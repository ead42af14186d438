"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))


def rel_err(a, b):
    """max |a-b| / max(|b|)  — the 'max-rel to max' metric of SURVEY.md App. B."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / den)


def l2_rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def oracle_inputs(case):
    """numpy recipe dict -> torch CPU tensors in the layout oracle.render_oracle expects."""
    from oracle.render_oracle import to_torch
    frame = to_torch(case["frame"])
    rays = to_torch(case["rays"])
    params = {k: torch.from_numpy(v) for k, v in case["weights"].items()}
    return params, frame, rays


def knn_bruteforce(points, K: int = 8, chunk: int = 4096):
    """Exact KNN by chunked distance matrices (tests only: the stand-in for `HipRenderer.knn` where no GPU is present).
    Ties: lower index first (knn_cpu.cpp:39-52 sorts (dist, idx) pairs).  -> callable q (N, 3) -> idx (N, K) int64"""
    import torch

    def f(q):
        out = []
        for i in range(0, q.shape[0], chunk):
            d = ((q[i:i + chunk, None, :] - points[None]) ** 2).sum(-1)
            out.append(torch.argsort(d, dim=1, stable=True)[:, :K])
        return torch.cat(out, 0)
    return f
