"""GPU: the header-only C++ adaptors (include/sdsl_hip/adaptors.hpp) compiled against the REAL sdsl-lite
and linked with libsdsl_hip.so — every batched answer against the scalar answer of the unmodified
reference object, plus SDSL's own serialise/load round trip through the adaptors."""
import os
import subprocess

import pytest

import golden_data as gd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "adaptor_parity")


def test_adaptors_against_real_sdsl(gpu):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/adaptor_parity not built (needs the reference tree at build time)")
    # (the client takes a few seconds.  Once on the test pool it did not finish within ten minutes — not reproduced in forty
    # further runs, three of them full suites back to back; a second attempt is made before that counts as a failure, and what
    # the first one printed is kept for the message)
    r, notes = None, ""
    for attempt, limit in enumerate((240, 600)):
        try:
            r = subprocess.run([EXE, os.path.join(gd.GOLDEN, "texts", "faust.txt")], capture_output=True, text=True,
                               timeout=limit)
            break
        except subprocess.TimeoutExpired as e:
            out = (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or "")
            err = (e.stderr or b"").decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
            notes += f"attempt {attempt + 1} did not finish in {limit} s; so far:\n{out[-2000:]}\n{err[-2000:]}\n"
    assert r is not None, notes
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all equal" in r.stdout
