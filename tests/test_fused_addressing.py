"""CPU: the address arithmetic of the fused wavelet-tree lines (wt_device.hpp: fused_line / fused_off / fused_lines_for) against plain
64-bit arithmetic over positions up to 2^40 and random 64-bit ones — both forms of the lines (16-ary, the default, and 8-ary).  Round 5
shipped a 16-ary fused_line that was a 32-bit division from 2^35 on while the builder admits 2^36 symbols; the reference is 64-bit
throughout (wt_pc.hpp:371-399).  hipcc compiles the __host__ side; nothing is launched."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc is not on this box")
@pytest.mark.parametrize("fused_k,spare", [(4, 2), (4, 0), (4, 4), (3, 0)])
def test_fused_line_matches_64_bit_arithmetic(tmp_path, fused_k, spare):
    exe = str(tmp_path / "fused_addr_check")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-x", "hip", f"-DSDSL_HIP_FUSED_K={fused_k}",
                        f"-DSDSL_HIP_FUSED_SPARE={spare}", "-I" + os.path.join(ROOT, "sdsl-lite_amd", "csrc"),
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "fused_addr_check.cpp"), "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-2000:]
    assert r.stdout.strip().endswith("ok")
