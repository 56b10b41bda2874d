"""CPU (build container: needs the reference's headers): the 128-bit content fingerprint of include/sdsl_hip/adaptors.hpp — deterministic
whatever the number of hashing threads, sensitive to a single flipped bit anywhere (first / middle / last stretch) and to the length."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/include"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "sdsl")), reason="the reference's headers are not on this box")
def test_fingerprint_of_the_replica_registry(tmp_path):
    exe = str(tmp_path / "fingerprint_check")
    lib = os.path.join(ROOT, "sdsl-lite_amd", "lib")
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-msse4.2", "-w", "-pthread", "-I" + REF, "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "cpp", "fingerprint_check.cpp"), "-o", exe, "-L" + lib, "-lsdsl_hip",
                        "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-500:]
