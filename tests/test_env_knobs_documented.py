"""CPU: every environment variable the library reads (getenv("SDSL_HIP_...") anywhere under sdsl-lite_amd/csrc) is described in
INTEGRATION.md, and so is every name sdsl_hip_set_option accepts (14 behaviour-changing switches were in no document)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sdsl-lite_amd", "csrc")


def sources():
    for dp, _, fs in os.walk(CSRC):
        for f in fs:
            if f.endswith((".hip", ".cpp", ".hpp")):
                yield os.path.join(dp, f)


def test_every_environment_variable_is_documented():
    doc = open(os.path.join(ROOT, "INTEGRATION.md"), encoding="utf-8").read()
    read = set()
    for path in sources():
        read.update(re.findall(r'getenv\("(SDSL_HIP_[A-Z0-9_]+)"\)', open(path, encoding="utf-8").read()))
    assert len(read) > 40
    missing = sorted(v for v in read if v not in doc)
    assert not missing, f"not in INTEGRATION.md: {missing}"


def test_every_option_name_is_documented():
    doc = open(os.path.join(ROOT, "INTEGRATION.md"), encoding="utf-8").read() + open(os.path.join(ROOT, "include", "sdsl_hip.h"), encoding="utf-8").read()
    src = open(os.path.join(CSRC, "common.cpp"), encoding="utf-8").read()
    body = src[src.index("sdsl_hip_status sdsl_hip_set_option"):]
    body = body[:body.index("set_option: unknown option")]
    names = set(re.findall(r'!strcmp\(name, "([a-z0-9_]+)"\)', body))
    assert {"rank_sorted", "select_sorted", "rrr_sorted", "rrr_sparse_limit", "group_timeout_ms"} <= names
    missing = sorted(n for n in names if ('"%s"' % n) not in doc and ("`%s`" % n) not in doc)
    assert not missing, f"options without a description: {missing}"
