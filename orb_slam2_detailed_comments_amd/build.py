"""Builds liborbx.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the repo snapshot.
-ffp-contract=off: no implicit FMA in host or device code (every fused operation is explicit).
"""
from __future__ import annotations
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "liborbx.so")
SOURCES = ["orbx_kernels.hip", "orbx_api.cpp", "orbx_geometry.cpp", "orbx_policies.cpp"]
HEADERS = ["orbx_device.h", "orbx_internal.h", "orbx_launch.h", "orbx_sincos.h",
           os.path.join("..", "..", "include", "orbx.h"), os.path.join("..", "..", "include", "orbx_pattern_data.h")]


def kernels_hash():
    """identity of the device code AND of what decides its launch geometry (kernel source, device header, the launch plans in
    orbx_api.cpp / orbx_launch.h / orbx_internal.h): profiles/*_traffic.json and *_sq.json carry it, so counters taken on other
    kernels or other launch plans read as 'not measured' in bench.py instead of as stale numbers"""
    import hashlib
    h = hashlib.sha256()
    for f in ("orbx_kernels.hip", "orbx_device.h", "orbx_api.cpp", "orbx_launch.h", "orbx_internal.h"):
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the ORB front-end has no non-HIP build")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, extra=()):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
           "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value",
           *extra, *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True,
                extra=("-Rpass-analysis=kernel-resource-usage",) if "--usage" in sys.argv else ()))
