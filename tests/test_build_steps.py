"""The device-assembly build step of the recovery kernels (csrc/strip_asm_nops.py): it may only remove the no-op
hipcc puts between two of the kernel's own term statements, and the shipped library must have been built with it."""
import importlib.util
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "jpeg-quantsmooth_amd" / "csrc"


def _tool():
    spec = importlib.util.spec_from_file_location("strip_asm_nops", CSRC / "strip_asm_nops.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


TERM = "\t;;#ASMSTART\n\tv_mul_f32 v140, v140, v141\n\tv_add_f32 v8, v8, v140\n\t;;#ASMEND\n"
PIN = "\t;;#ASMSTART\n\t;;#ASMEND\n"
PARTIAL = "\t;;#ASMSTART\n\tv_cvt_pkrtz_f16_f32 v3, v1, v2\n\t;;#ASMEND\n"
# the small-plane kernel's optional term: scalar compare-and-branch around the VALU body, the statement ends in a label
OPT = "\t;;#ASMSTART\n\ts_cmp_lg_u32 s4, 0\n\ts_cbranch_scc1 1f\n\tv_sub_f32 v1, v2, v3\n\tv_add_f32 v8, v8, v1\n1:\n\t;;#ASMEND\n"
# a partial-register write in the MIDDLE of a statement whose last instruction is on the list
MIDDLE = "\t;;#ASMSTART\n\tv_cvt_pkrtz_f16_f32 v3, v1, v2\n\tv_add_f32 v8, v8, v140\n\t;;#ASMEND\n"


@pytest.mark.parametrize("text,removed,kept", [
    (TERM + "\ts_nop 0\n" + TERM, 1, 0),                                  # term -> term: goes
    (TERM + "\ts_nop 0\n\tv_mov_b32_e32 v1, v2\n" + TERM, 0, 1),          # compiler code behind it: stays
    (PIN + "\ts_nop 0\n" + TERM, 0, 1),                                   # behind an empty register pin: stays
    (PARTIAL + "\ts_nop 0\n" + TERM, 0, 1),                               # last instruction not on the list: stays
    (TERM + "\ts_nop 1\n" + TERM, 0, 0),                                  # a longer wait is not this hazard: untouched
    (TERM + "\ts_nop 0\n" + TERM + "\ts_nop 0\n" + PIN + "\ts_nop 0\n" + TERM, 2, 1),
    (OPT + "\ts_nop 0\n" + TERM, 1, 0),                                   # branchy statement, every instruction on the list: goes
    (MIDDLE + "\ts_nop 0\n" + TERM, 0, 1),                                # any instruction off the list: stays
])
def test_strip_only_between_two_term_statements(tmp_path, text, removed, kept, capsys):
    src, dst = tmp_path / "in.s", tmp_path / "out.s"
    src.write_text(text)
    rc = _tool().main(str(src), str(dst))
    said = capsys.readouterr().out
    assert f"{removed} no-ops between two asm statements removed, {kept} after an asm statement kept" in said
    assert rc == 0
    out = dst.read_text()
    assert out.count("s_nop") == text.count("s_nop") - removed
    assert [l for l in out.split("\n") if "s_nop" not in l] == [l for l in text.split("\n") if "s_nop" not in l]


def test_strip_floor_fails_loudly(tmp_path, capsys):
    """csrc/Makefile passes --min-removed for the shipped translation unit: an assembly printer that no longer marks the
    asm statements (nothing found to strip) must fail the build, not ship slower kernels silently"""
    src, dst = tmp_path / "in.s", tmp_path / "out.s"
    src.write_text(TERM + "\ts_nop 0\n" + TERM)
    assert _tool().main(str(src), str(dst), 1) == 0
    assert _tool().main(str(src), str(dst), 2) == 1
    assert "expected at least 2" in capsys.readouterr().err
    mk = (CSRC / "Makefile").read_text()
    assert "STRIP_FLOOR" in mk and "build_stripped.sh" in mk


def test_shipped_library_was_built_through_the_strip(tmp_path):
    """the term streams of the shipped code object have no `s_nop 0` between two v_add_f32 / v_sub_f32 of consecutive terms"""
    lib = ROOT / "jpeg-quantsmooth_amd" / "libjpegqs_hip.so"
    if not lib.exists():
        pytest.skip("library not built")
    llvm = Path("/opt/rocm/lib/llvm/bin")
    if not (llvm / "clang-offload-bundler").exists():
        pytest.skip("no ROCm LLVM tools")
    fat = tmp_path / "fat.bin"
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", str(lib), str(fat)], check=True)
    data = fat.read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [i for i in range(len(data)) if data.startswith(magic, i)]
    worst = None
    for n, a in enumerate(starts):
        b = starts[n + 1] if n + 1 < len(starts) else len(data)
        (tmp_path / f"b{n}.hipfb").write_bytes(data[a:b])
        co = tmp_path / f"b{n}.co"
        subprocess.run([str(llvm / "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={tmp_path / f'b{n}.hipfb'}", f"--output={co}"], check=True)
        dis = subprocess.run([str(llvm / "llvm-objdump"), "-d", str(co)], check=True, capture_output=True, text=True).stdout
        if "qs_smooth_set_kernel" not in dis:
            continue
        ops, inside = [], False          # the one-block-per-lane kernels only (the small-plane kernel's optional terms keep theirs)
        for l in dis.split("\n"):
            if l.endswith(">:"):
                inside = "qs_smooth_set_kernel" in l or "qs_smooth_plane_kernel" in l
            elif inside and l.startswith("\t") and l.split():
                ops.append(l.split()[0])
        # pattern of an unstripped build: accumulate, no-op, first instruction of the next term
        worst = sum(1 for i in range(1, len(ops) - 1) if ops[i] == "s_nop" and ops[i - 1].startswith("v_add_f32")
                    and (ops[i + 1].startswith("v_sub_f32") or ops[i + 1].startswith("v_mul_f32")))
    assert worst is not None, "no code object with the recovery kernels found"
    assert worst == 0, f"{worst} no-ops between term statements: qs_kernels.o was not built through strip_asm_nops.py"


def test_build_falls_back_to_plain_hipcc_when_the_llvm_tools_are_missing(tmp_path):
    """csrc/build_stripped.sh: a toolchain without lld / clang-offload-bundler next to clang still produces the object
    (plain one-step `hipcc -c`, a warning on stderr) -- and refuses to when QS_REQUIRE_STRIP=1.  LLVMBIN points the
    script at an empty directory; a small stand-in translation unit keeps the test to seconds."""
    import os
    hipcc = Path("/opt/rocm/bin/hipcc")
    if not hipcc.exists():
        pytest.skip("no hipcc")
    src = tmp_path / "tiny.hip"
    src.write_text("#include <hip/hip_runtime.h>\n__global__ void k(float* p) { p[threadIdx.x] += 1.0f; }\n")
    empty = tmp_path / "no_llvm"; empty.mkdir()
    out = tmp_path / "tiny.o"
    env = dict(os.environ, LLVMBIN=str(empty), HIPCC=str(hipcc))
    r = subprocess.run(["bash", str(CSRC / "build_stripped.sh"), str(src), str(out), "0", "--offload-arch=gfx950", "-O2", "-fPIC"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "WARNING: LLVM tool 'clang' not found" in r.stderr and "plain one-step hipcc -c" in r.stderr
    assert out.exists() and out.stat().st_size > 0
    out.unlink()
    r = subprocess.run(["bash", str(CSRC / "build_stripped.sh"), str(src), str(out), "0", "--offload-arch=gfx950", "-O2", "-fPIC"],
                       capture_output=True, text=True, env=dict(env, QS_REQUIRE_STRIP="1"), timeout=600)
    assert r.returncode != 0 and "QS_REQUIRE_STRIP=1: giving up" in r.stderr and not out.exists()
