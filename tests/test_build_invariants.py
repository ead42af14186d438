"""Build-time invariants of the HIP kernels that the parity tests cannot see.

`mv_stats_kernel` and `knn_wave_kernel` use the lanes of a wave as storage: a value is computed by lane v (one view, one list
slot) and later fetched with v_readlane / DPP by code that runs under a different EXEC mask.  A VGPR that the register allocator
spills inside a divergent region is saved for the active lanes only, which is invisible to ordinary SIMT code but not to a
cross-lane read — measured: a build of mv_stats with 9 spilled registers rendered the 10-view golden case 1.6e-3 off while 16
other cases passed.  So these kernels must compile without any VGPR spill, whatever the compiler version."""
import os
import re
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nerf_loc_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src,kernel", [("mvagg.hip", "mv_stats_kernel"), ("knn.hip", "knn_wave_kernel")])
def test_lane_storage_kernels_do_not_spill(tmp_path, src, kernel):
    flags = subprocess.run(["make", "-s", "-C", CSRC, "-pn"], capture_output=True, text=True).stdout
    m = re.search(r"^FLAGS := (.*)$", flags, re.M)
    assert m, "Makefile FLAGS"
    cmd = [HIPCC] + m.group(1).split() + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", str(tmp_path / "x.o")]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    name, seen = None, 0
    for line in out.stderr.splitlines():
        f = re.search(r"Function Name: (\S+)", line)
        if f:
            name = f.group(1)
        s = re.search(r"VGPRs Spill: (\d+)", line)
        if s and name and kernel in name:
            seen += 1
            assert int(s.group(1)) == 0, f"{name} spills {s.group(1)} VGPRs"
    assert seen > 0, f"no {kernel} instantiation found in the remarks"


def test_fp6_packing_conversions_go_through_the_early_clobber_wrappers():
    """hipcc 7.2 lets the 6-register result of __builtin_amdgcn_cvt_scalef32_{pk32_fp6_f16, 2xpk16_fp6_f32} overlap the scale operand
    (v_cvt_scalef32_2xpk16_fp6_f32 v[206:211], v[122:137], v[138:153], v206): the multi-pass instruction then reads a scale it has already
    overwritten and one K slab of one layer of the fused neural-point kernel carried garbage residuals (5e-4 instead of 1.5e-5 on the
    goldens; found with tools/mx6_debug.py).  The kernel sources must issue these conversions only through asm statements whose result
    is an early-clobber operand."""
    for fn in os.listdir(CSRC):
        if not fn.endswith((".hip", ".h")):
            continue
        text = open(os.path.join(CSRC, fn)).read()
        assert "__builtin_amdgcn_cvt_scalef32_pk32_fp6" not in text and "__builtin_amdgcn_cvt_scalef32_2xpk16_fp6" not in text, fn
        for m in re.finditer(r'asm\("v_cvt_scalef32_(?:pk32|2xpk16)_fp6[^"]*"\s*:\s*"([^"]*)"', text):
            assert m.group(1) == "=&v", f"{fn}: fp6 packing conversion without an early-clobber result"
