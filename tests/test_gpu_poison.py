"""SDSL_HIP_POISON=<byte> fills every device allocation that is not asked to be zeroed with that byte (common.cpp: DevBuf::alloc) — fresh
device memory is usually zero, which hides reads of memory nobody has written.  The large-batch paths (bucketed rank / select, count
with its record lists, the locate / extract work areas, the footprint changes) must give the same answers on poisoned memory."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_large_batch_paths_on_poisoned_allocations(gpu):
    env = dict(os.environ, SDSL_HIP_POISON="165")
    files = ["tests/test_gpu_bucketed.py", "tests/test_gpu_fm_fast.py", "tests/test_gpu_fm_footprint.py", "tests/test_gpu_wt_sorted.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + files, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
