"""The "fp16 hi.hi + two MX-FP6 cross terms" arithmetic of the fused neural-point kernel on one K = 64 slab, outside the kernel:
tools/ubench/mx6_slab_check.hip applies the kernel's conversions (v_cvt_pk_f16_f32, exact residuals, v_cvt_scalef32_pk32_fp6_f16, v_cvt_scalef32_2xpk16_fp6_f32 with
the block scale 2^(floor(log2 max) - 2)), the host-side weight images in pack_point_mx6_kernel's position order, and the three kinds of matrix instructions, and
compares with the exact product.  Each cross term alone must remove part of the fp16 product's error, both together must bring it to the 1e-5 class."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_one_slab_of_the_mx_fp6_product_matches_the_exact_product(tmp_path):
    exe = str(tmp_path / "slab")
    b = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(ROOT, "tools", "ubench", "mx6_slab_check.hip")], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-2000:]
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    err = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"^(.*?)\s+worst \|error\| / max \|D\| = (\S+)$", out.stdout, re.M)}
    assert set(err) == {"hi.hi alone", "+ w_hi6 x a_lo6", "+ w_lo6 x a_hi6", "+ both"}, out.stdout
    assert err["hi.hi alone"] > 1e-4                      # fp16 alone is not a parity arithmetic
    assert err["+ w_hi6 x a_lo6"] < err["hi.hi alone"] and err["+ w_lo6 x a_hi6"] < err["hi.hi alone"]
    assert err["+ both"] < 3e-5, out.stdout              # measured 1.3e-5 (fp8 cross terms: the same class)
