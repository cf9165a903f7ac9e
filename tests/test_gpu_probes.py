"""Hardware facts the kernels rely on, re-checked on the GPU box with exact integer data (each probe is a stand-alone HIP program
under tools/, compiled here with hipcc -- the same image as the build container)."""
import os
import shutil
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def test_mfma_fp4_instruction_semantics(tmp_path):
    """k_match_f4 (csrc/orbx_kernels.hip) rests on v_mfma_scale_f32_32x32x64_f8f6f4 with e2m1 operands: nibble values 2.0 x -4.0,
    the block scale taken from byte 0 of the scale registers, lane (l & 31, l >> 5) supplying 32 K elements, the dtype-independent
    32x32 C/D layout, and exact f32 sums of integers below 2^24.  tools/mfma_fp4_probe.hip checks all of that on 4096 outputs."""
    exe = str(tmp_path / "mfma_fp4_probe")
    b = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O2", "-w", os.path.join(ROOT, "tools", "mfma_fp4_probe.hip"), "-o", exe],
                       capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "PROBE OK" in r.stdout, (r.stdout + r.stderr)[-2000:]
