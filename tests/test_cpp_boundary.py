"""The C ABI used from plain C++ (include/orbx.hpp + examples/orbx_demo.cpp, g++ only: no OpenCV, no HIP headers).
CPU: the demo compiles with -Wall -Werror and links against liborbx.so, and C sees every declaration of orbx.h.
GPU: the demo runs: extraction of two synthetic frames + ratio-test matching through the C++ wrapper."""
import os
import shutil
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "orb_slam2_detailed_comments_amd", "lib")


def _build(tmp_path):
    exe = str(tmp_path / "orbx_demo")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "orbx_demo.cpp"), "-L" + LIBDIR, "-lorbx", "-Wl,-rpath," + LIBDIR, "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    return exe


def test_cpp_demo_compiles_and_links(built_lib, tmp_path):
    assert shutil.which("g++")
    _build(tmp_path)


def test_header_is_plain_c(built_lib, tmp_path):
    src = tmp_path / "c_only.c"
    src.write_text('#include "orbx.h"\nint main(void) { orbx_params p; orbx_default_params(&p); return p.nlevels == 8 ? 0 : 1; }\n')
    exe = str(tmp_path / "c_only")
    p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"), str(src), "-L" + LIBDIR, "-lorbx",
                        "-Wl,-rpath," + LIBDIR, "-o", exe], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    assert subprocess.run([exe]).returncode == 0          # orbx_default_params needs no device


@pytest.mark.gpu
def test_cpp_demo_runs_on_the_gpu(tmp_path):
    exe = _build(tmp_path)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ratio-test matches" in p.stdout
