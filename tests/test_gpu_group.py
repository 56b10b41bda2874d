"""GPU: the multi-device group of the C ABI.  The cases live in tests/group_cases.py and run in a child
process with a time limit: they create RCCL communicators in-process, and a communicator that never comes up (seen once on
a test box) must fail this test, not hang the whole run."""
import os
import signal
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_group_cases_in_a_child_process(gpu, tmp_path):
    out = tmp_path / "out.txt"
    with open(out, "w") as fo:
        p = subprocess.Popen([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "group_cases.py"), "-x", "-q", "-m", "gpu",
                              "-p", "no:cacheprovider"], stdout=fo, stderr=subprocess.STDOUT, cwd=ROOT, start_new_session=True)
        try:
            rc = p.wait(timeout=600)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)
            p.wait()
            rc = -9
    text = out.read_text()
    assert rc == 0, text[-4000:]
    assert " passed" in text and "failed" not in text, text[-2000:]
