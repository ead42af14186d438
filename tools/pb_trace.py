"""Debug: per-region cycle trace of point_bwd_chain_kernel during one PoseOptimizer-sized gradient step (needs a library built with -DPB_TRACE; NERFLOC_LIB selects it)."""
import ctypes as ct, os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "512", "3", "f16mx"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pose_refine_bench.py")).read().split("res = {}")[0])
for _ in range(3): step(True)
torch.cuda.synchronize()
from nerf_loc_amd import _lib as L
buf = (ct.c_ulonglong * 256)()
assert L.load().nl_debug_pb_trace(buf) == 0
t = np.array(buf[:], dtype=np.int64).reshape(4, 64)
NC = 28
for it in range(4):
    d = np.diff(t[it, :NC + 1])
    print(f"tile {it}: total {t[it, NC] - t[it, 0]} cycles; per region:", d.tolist())
