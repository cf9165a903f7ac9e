#!/usr/bin/env python3
"""Static VALU rate-class mix per kernel of liborbx.so (disassembly of the gfx950 code object), for bench.py's roofline.ports.

Issue cost per wave64 VALU instruction on one SIMD-32, from the round-1 issue-rate measurements (profiles/r01_valu_rate*_gfx950.log,
4 waves per SIMD, every CU busy; ns at the clock the chip held, here in cycles of a 2-cycle full-rate slot):
  full rate (2 cycles): VOP1 / VOP2 encodings of integer / bit / 16-bit min-max / f32 add-mul-fma with VGPR or inline operands;
  3 cycles            : shifts, v_fmac_f32;
  half rate (4 cycles): everything VOP3 / VOP3P / DPP / SDWA encoded, every v_cmp, 32-bit integer min / max, v_mul_u32_u24, v_cvt_*,
                        v_rndne, f32 min / max, and a VOP2 instruction with an SGPR source; MFMA and transcendental ops are not VALU
                        issue in this sense and are listed apart;
  quarter rate (8 cycles): the THREE-input 16-bit forms v_min3 / v_max3 / v_med3 _i16 / _u16 (8.3-8.5 cycles measured, round 3,
                        tools/valu_rate6.hip: one of them costs as much as three v_min_i16) and the full 32-bit multiplies.
The mix is STATIC (every instruction of the kernel counted once, no execution weights): an estimate of the class share, not a count.

  tools/valu_mix.py [liborbx.so]      ->  JSON {kernel: {valu, full, mid, half, quarter, cycles_per_valu}}"""
import json, os, re, subprocess, sys, tempfile

HALF_NAMES = ("v_perm_b32", "v_alignbyte", "v_alignbit", "v_bfe_", "v_bfi_", "v_lshl_add", "v_add_lshl", "v_add3", "v_mad_", "v_and_or", "v_or3",
              "v_lshl_or", "v_med3", "v_min3", "v_max3", "v_dot", "v_sad", "v_msad", "v_mbcnt", "v_pk_", "v_bitop3", "v_readlane", "v_writelane",
              "v_mul_u32_u24", "v_mul_i32_i24", "v_mul_lo", "v_mul_hi", "v_cvt", "v_rndne", "v_min_f32", "v_max_f32", "v_min_i32", "v_max_i32",
              "v_min_u32", "v_max_u32", "v_cmp", "v_cndmask_b32_e64", "v_xad", "v_div", "v_ldexp", "v_frexp", "v_trunc", "v_floor", "v_fract")
MID_NAMES = ("v_lshlrev", "v_lshrrev", "v_ashrrev", "v_fmac")
QUARTER_NAMES = ("v_min3_i16", "v_max3_i16", "v_med3_i16", "v_min3_u16", "v_max3_u16", "v_med3_u16", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32")
SKIP = ("v_mfma", "v_accvgpr", "v_nop")


def disassemble(lib):
    b = "/opt/rocm/lib/llvm/bin/"
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "gfx950.co")
        subprocess.check_call([b + "llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(d, "copy.so")])
        subprocess.check_call([b + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"])
        return subprocess.run([b + "llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout


def classify(op, operands):
    if op.startswith(SKIP):
        return None
    if op.startswith(QUARTER_NAMES):
        return "quarter"
    if op.endswith(("_e64", "_sdwa", "_dpp")) or op.startswith(HALF_NAMES):
        return "half"
    if op.startswith(MID_NAMES):
        return "mid"
    srcs = operands.split(",")[1:]
    if any(re.match(r"\s*(s\d+|s\[\d+:\d+\]|vcc|exec|m0)", s) for s in srcs):
        return "half"          # a full-rate VOP2 with a scalar-register source issues at the half rate (valu_rate4)
    return "full"


def mix(lib):
    out, cur = {}, None
    for line in disassemble(lib).splitlines():
        m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
        if m:
            name = m.group(1)
            mm = re.match(r"_Z(\d+)", name)          # Itanium mangling: _Z <length> <name> ...  (template instances share the name)
            cur = name[len(mm.group(0)):len(mm.group(0)) + int(mm.group(1))] if mm else None
            if cur and not cur.startswith("k_"):
                cur = None
            if cur:
                out.setdefault(cur, {"valu": 0, "full": 0, "mid": 0, "half": 0, "quarter": 0})
            continue
        m = re.match(r"^\s+(v_\w+)\s*(.*?)\s*//", line)
        if m and cur:
            c = classify(m.group(1), m.group(2))
            if c:
                out[cur]["valu"] += 1
                out[cur][c] += 1
    for k, v in out.items():
        v["cycles_per_valu"] = round((2 * v["full"] + 3 * v["mid"] + 4 * v["half"] + 8 * v["quarter"]) / max(v["valu"], 1), 3)
    return out


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from orb_slam2_detailed_comments_amd import build
    print(json.dumps(mix(sys.argv[1] if len(sys.argv) > 1 else build.LIB), indent=1))
