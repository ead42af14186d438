"""KNN oracle (C restatement + numpy restatement) vs the reference's own knn_cpu.cpp (oracle/_ref) and brute force."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import render_oracle as orc

REF_SO = os.path.join(os.path.dirname(orc.__file__), "_ref", "libref_knn.so")


def _ref_knn(q, p, K):
    """reference op + the ascending sort of its python wrapper (knn_utils.py:60-74)."""
    lib = ctypes.CDLL(REF_SO)
    lib.ref_knn_cpu.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    n, m = q.shape[0], p.shape[0]
    idx = np.zeros((n, K), np.int64)
    d2 = np.zeros((n, K), np.float32)
    lib.ref_knn_cpu(q.ctypes.data, n, p.ctypes.data, m, K, d2.ctypes.data, idx.ctypes.data)
    return d2, idx


def _clouds(seed, n, m, dup=False):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((n, 3)).astype(np.float32)
    p = rng.standard_normal((m, 3)).astype(np.float32)
    if dup and m > 4:
        p[1::3] = p[0:-1:3][: len(p[1::3])]
        p = np.round(p * 4) / 4  # lattice -> many exact ties
        q = np.round(q * 4) / 4
    return q, p


@pytest.mark.parametrize("n,m,K,dup", [(300, 500, 8, False), (200, 300, 8, True), (100, 5, 8, False), (64, 200, 1, False), (50, 40, 8, True)])
def test_c_oracle_equals_numpy_restatement(n, m, K, dup):
    q, p = _clouds(n + m, n, m, dup)
    d_np, i_np = orc.knn_points_np(q, p, K)
    lib = orc._load_knn_lib()
    assert lib, "oracle/libknn_oracle.so missing: run `make -C oracle`"
    d_c, i_c = orc.knn_points(torch.from_numpy(q), torch.from_numpy(p), K, threads=2)
    assert np.array_equal(d_np, d_c.numpy())
    assert np.array_equal(i_np, i_c.numpy())


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (reference absent)")
@pytest.mark.parametrize("n,m,K,dup", [(300, 500, 8, False), (200, 300, 8, True), (100, 5, 8, False), (64, 200, 1, False)])
def test_c_oracle_equals_reference_knn_cpu(n, m, K, dup):
    q, p = _clouds(7 * n + m, n, m, dup)
    d_r, i_r = _ref_knn(q, p, K)
    d_c, i_c = orc.knn_points(torch.from_numpy(q), torch.from_numpy(p), K)
    assert np.array_equal(d_r, d_c.numpy()), "squared distances must be bit-exact incl. tie order"
    assert np.array_equal(i_r, i_c.numpy()), "the heap's (dist, idx) tuple order is reproduced exactly"
