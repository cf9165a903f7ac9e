"""Typo guard for compat/ (pins nothing): the three drop-in shims -- compat/ORBextractor.h, compat/ORBmatcher.h, compat/
Frame_stereo.inl, 600 lines of C++ that need OpenCV and the ORB-SLAM2 headers and therefore meet no compiler in this image --
are parsed and type-checked with `g++ -std=c++14 -Wall -fsyntax-only` against tests/compat_stubs/, which declares exactly the
cv:: / ORB_SLAM2:: members they touch (tests/compat_stubs/README.md).  It protects the shims from the next edit; the real check
is the maintainer's build inside an ORB-SLAM2 + OpenCV tree."""
import os
import shutil
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUBS = os.path.join(ROOT, "tests", "compat_stubs")


@pytest.mark.parametrize("unit", ["driver_extractor.cpp", "driver_matcher.cpp", "driver_frame.cpp"])
def test_shim_parses_and_type_checks(unit):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not found")
    p = subprocess.run([gxx, "-std=c++14", "-Wall", "-fsyntax-only", "-I" + STUBS, "-I" + os.path.join(ROOT, "compat"),
                        "-I" + os.path.join(ROOT, "include"), os.path.join(STUBS, unit)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "warning" not in p.stderr, p.stderr[-4000:]


def test_matcher_shim_declares_every_reference_signature():
    """the eleven search methods + DescriptorDistance + the three constants of /root/reference/include/ORBmatcher.h:54-225 (names and
    arity; the driver above instantiates each with the reference's argument types)"""
    src = open(os.path.join(ROOT, "compat", "ORBmatcher.h")).read()
    for name, count in (("SearchByProjection", 4), ("SearchByBoW", 2), ("SearchForInitialization", 1), ("SearchForTriangulation", 1),
                        ("SearchBySim3", 1), ("Fuse", 2), ("DescriptorDistance", 1)):
        assert src.count(f" {name}(") + src.count(f"\n    int {name}(") >= count, name
    for const in ("TH_LOW", "TH_HIGH", "HISTO_LENGTH"):
        assert const in src
