"""Index builds and group batches in a loop, on the SYSTEM HIP runtime (a C++ client over the C ABI; the Python tests run on the
runtime torch ships).  Round 3's intermittent start-up hang of the adaptor parity client was this: the stream-ordered allocator
(hipMallocAsync / hipFreeAsync) behind the builders' sort helpers and the count path's scratch — under the ROCm 7.2 runtime one
build in a few thousand hung inside a wavelet-tree level sort, died with a GPU memory fault or left wrong tables behind, and two
handles answering on two streams of one device now and then saw each other's scratch.  The allocator is out of the
library; these loops are the regression test (before the fix: one of three runs of phase D failed)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "sdsl-lite_amd", "lib", "group_stress")


def run(rounds, transport, phases, limit):
    assert os.path.exists(EXE), "sdsl-lite_amd/lib/group_stress is built by build()"
    r = subprocess.run([EXE, str(rounds), transport, phases], capture_output=True, text=True, timeout=limit)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return r.stdout


def test_rebuilding_an_index_in_a_loop_neither_hangs_nor_corrupts(gpu):
    out = run(3000, "none", "D", 240)
    assert "D: 0 rounds with different index bytes, 0 with wrong counts, of 3000" in out, out[-1500:]


@pytest.mark.parametrize("transport", ["copy2", "rccl1"])
def test_group_batches_and_rebuilt_replicas_in_a_loop(gpu, transport):
    out = run(600, transport, "AB", 300)
    assert "A (fixed replicas, %s): 0 of 600 rounds with mismatches" % transport in out, out[-1500:]
    assert "B (replicas rebuilt by the group's builder threads): 0 of 600 rounds with mismatches" in out, out[-1500:]
