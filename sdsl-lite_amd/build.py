"""Builds libsdsl_hip.so (HIP kernels + C ABI) for gfx950, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container and the
resulting .so travels to the GPU box with the repo snapshot.  No torch.utils.cpp_extension:
the library is a plain C-ABI shared object with no torch types in its interface.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libsdsl_hip.so")

ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-gpu-rdc"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libsdsl_hip.so cannot be built (there is no CPU fallback)")


def _sources() -> list[str]:
    out = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".cpp")):
            out.append(os.path.join(CSRC, f))
    return out


def _headers_mtime() -> float:
    m = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for dp, _, fs in os.walk(root):
            for f in fs:
                if f.endswith((".hpp", ".h")):
                    m = max(m, os.path.getmtime(os.path.join(dp, f)))
    return m


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("command failed: " + " ".join(cmd))


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    # SDSL_HIP_FUSED_K=3|4 in the environment builds the other form of the fused wavelet-tree lines (wt_device.hpp); the objects
    # remember the flags they were built with
    extra = ["-DSDSL_HIP_FUSED_K=" + os.environ["SDSL_HIP_FUSED_K"]] if os.environ.get("SDSL_HIP_FUSED_K") else []
    extra += os.environ.get("SDSL_HIP_EXTRA_FLAGS", "").split()
    stamp = os.path.join(OBJDIR, "flags.txt")
    if (open(stamp).read() if os.path.exists(stamp) else "") != " ".join(extra):
        force = True
    hdr_m = _headers_mtime()
    objs, cmds = [], []
    for src in _sources():
        obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_m)):
            continue
        cmd = [hipcc, f"--offload-arch={ARCH}", *CXXFLAGS, *extra, "-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        cmds.append(cmd)
    rebuilt = bool(cmds)
    if cmds:  # translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(cmds), os.cpu_count() or 1)) as ex:
            list(ex.map(_run, cmds))
        with open(stamp, "w") as fh:
            fh.write(" ".join(extra))
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB, "-Wl,-rpath,/opt/rocm/lib",
               "-Wl,--no-undefined"]
        if verbose:
            print(" ".join(cmd))
        _run(cmd)
    # calibration tool (not part of the library)
    tool_src = os.path.join(CSRC, "tools", "gather_probe.hip")
    tool = os.path.join(LIBDIR, "gather_probe")
    if os.path.exists(tool_src) and (force or not os.path.exists(tool)
                                     or os.path.getmtime(tool) < os.path.getmtime(tool_src)):
        _run([hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", tool_src, "-o", tool,
              "-Wl,-rpath,/opt/rocm/lib"])
    # stress client over the C ABI on the SYSTEM HIP runtime (tests/test_gpu_stress_build.py; the Python tests run on the runtime torch ships)
    st_src = os.path.join(HERE, "..", "tools", "cpp", "group_stress.cpp")
    st_bin = os.path.join(LIBDIR, "group_stress")
    if os.path.exists(st_src) and (force or rebuilt or not os.path.exists(st_bin) or os.path.getmtime(st_bin) < os.path.getmtime(st_src)):
        _run(["g++", "-O2", "-std=c++17", st_src, "-I" + os.path.join(HERE, "..", "include"), "-L" + LIBDIR, "-lsdsl_hip",
              "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-o", st_bin])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
