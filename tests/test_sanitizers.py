"""Sanitizer evidence for the CPU side (GPU AddressSanitizer is not available on this pool).

* the oracle (oracle/orb_oracle.c, orb_oracle_match.c) is rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer
  (oracle/Makefile target `sanitize`) and the CPU tests that drive it -- known answers, primitives, matcher policies, the
  quadtree model, fp_mode -- run against that build in a child interpreter (the ASan runtime has to be loaded first);
* F6 (reference src/Frame.cc:910,918,934-941 with the fork's padded pyramid): the reference's unchecked vRowIndices[yi]
  IS an out-of-bounds heap access for a level-7 keypoint at the bottom of a 480-row image -- shown as an expected
  AddressSanitizer report of oracle/f6_demo.c built with -DORC_F6_UNCLAMPED, next to a clean run of the clamped restatement
  (the clamp the HIP path shares).
"""
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")
SAN = os.path.join(ORACLE, "_san")
CPU_TESTS = ["test_oracle_kat.py", "test_oracle_primitives.py", "test_policies_cpu.py", "test_quadtree_model.py",
             "test_fp_mode_oracle.py"]


@pytest.fixture(scope="module")
def san_build():
    subprocess.check_call(["make", "-C", ORACLE, "sanitize"], stdout=subprocess.DEVNULL)
    return SAN


def _asan_runtime():
    p = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    return os.path.realpath(p)


def test_cpu_oracle_tests_clean_under_asan_ubsan(san_build):
    env = dict(os.environ)
    env.update(LD_PRELOAD=_asan_runtime(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",   # the interpreter itself is not leak-clean
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", ORB_ORACLE_LIB=os.path.join(san_build, "liborb_oracle_asan.so"))
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu"] +
                       [os.path.join(ROOT, "tests", t) for t in CPU_TESTS], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-4000:]
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert " passed" in out


def test_f6_unchecked_row_index_is_a_heap_overflow_and_the_clamp_is_clean(san_build):
    clean = subprocess.run([os.path.join(san_build, "f6_clamped")], capture_output=True, text=True, timeout=60)
    assert clean.returncode == 0 and "clean run" in clean.stdout and "AddressSanitizer" not in clean.stderr
    bad = subprocess.run([os.path.join(san_build, "f6_unclamped")], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0
    assert "AddressSanitizer: heap-buffer-overflow" in bad.stderr and "orc_stereo_matches" in bad.stderr


def test_host_geometry_unit_clean_under_asan_ubsan(tmp_path):
    """csrc/orbx_geometry.cpp is the HIP-free host translation unit of the product (tables, cells, taps, slot ranges): built
    with g++ -fsanitize=address,undefined together with tests/san_geometry.cpp, swept over BASELINE.json's sizes and 400
    random (size, nfeatures, scale factor, levels) combinations with the kernels' invariants checked."""
    exe = str(tmp_path / "san_geometry")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", os.path.join(ROOT, "tests", "san_geometry.cpp"),
                           os.path.join(ROOT, "orb_slam2_detailed_comments_amd", "csrc", "orbx_geometry.cpp"), "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "0 failures" in p.stdout and "AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr
