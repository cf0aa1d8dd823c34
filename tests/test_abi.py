"""CPU tests of the product library's boundary: it loads, exports every symbol
include/jpegqs_hip.h declares, builds bit-exact constants on the host, keeps
the reference's flag values and fails loudly (never falls back) without a GPU."""
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_header_symbols_all_exported(pkg, hip):
    text = (ROOT / "include" / "jpegqs_hip.h").read_text()
    declared = set(re.findall(r"\b(qs_hip_[a-z_0-9]+)\s*\(", text))
    declared.discard("qs_hip_progress_fn")
    assert declared, "no declarations parsed"
    from jpeg_quantsmooth_amd import hipqs
    assert declared == set(hipqs.ABI), declared ^ set(hipqs.ABI)
    for name in declared:
        assert getattr(hip.lib, name) is not None


def test_struct_layouts_match_the_header(tmp_path):
    """the ctypes mirrors of qs_hip_job and qs_hip_plane_ref against what a C compiler makes of include/jpegqs_hip.h:
    size and the offset of every field (a C program that includes the header prints them)"""
    import ctypes as C
    import subprocess
    from jpeg_quantsmooth_amd import hipqs
    fields = {"qs_hip_job": [f[0] for f in hipqs.Job._fields_], "qs_hip_plane_ref": [f[0] for f in hipqs.PlaneRef._fields_]}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "jpegqs_hip.h"', 'int main(void) {']
    for st, fs in fields.items():
        src.append(f'  printf("{st} %zu\\n", sizeof({st}));')
        for f in fs:
            src.append(f'  printf("{st}.{f} %zu\\n", offsetof({st}, {f}));')
    src += ['  return 0;', '}']
    (tmp_path / "layout.c").write_text("\n".join(src))
    subprocess.run(["gcc", "-I", str(ROOT / "include"), "-o", str(tmp_path / "layout"), str(tmp_path / "layout.c")], check=True)
    out = dict(line.split() for line in subprocess.run([str(tmp_path / "layout")], capture_output=True, text=True, check=True).stdout.splitlines())
    for st, cls in (("qs_hip_job", hipqs.Job), ("qs_hip_plane_ref", hipqs.PlaneRef)):
        assert int(out[st]) == C.sizeof(cls), (st, out[st], C.sizeof(cls))
        for f in fields[st]:
            assert int(out[f"{st}.{f}"]) == getattr(cls, f).offset, (st, f)


def test_plane_ref_keeps_its_original_size_and_abi_version():
    """qs_hip_plane_ref is 48 bytes, as in every header since the plane-set calls appeared (a trailing
    d_plane_next field changed the array stride under callers that never zeroed it -- the second planes now travel in
    the parallel array of qs_hip_smooth_planes_next); the library reports the header's ABI version"""
    import ctypes as C
    import re
    from jpeg_quantsmooth_amd import hipqs
    assert C.sizeof(hipqs.PlaneRef) == 48
    assert [f[0] for f in hipqs.PlaneRef._fields_] == ["d_consts", "d_coef", "d_plane", "d_status", "wblk", "hblk", "luma", "band"]
    hdr = (ROOT / "include" / "jpegqs_hip.h").read_text()
    ver = int(re.search(r"#define QS_HIP_ABI_VERSION (\d+)", hdr).group(1))
    hip = hipqs.HipQS.__new__(hipqs.HipQS)
    lib = hipqs.load_library()
    assert lib.qs_hip_abi_version() == ver >= 5
    refs = hipqs.HipQS.plane_refs([(1, 2, 3, 4, 5, 6, 1), (1, 2, 3, 4, 5, 6, 0, 2, 99)])
    assert len(refs) == 2 and refs.next is not None and refs.next[0] is None and refs.next[1] == 99 and refs[1].band == 2
    assert hipqs.HipQS.plane_refs([(1, 2, 3, 4, 5, 6, 1)]).next is None


def test_experiments_translation_unit_still_builds(tmp_path):
    """tools/experiments/ (round-3 kernel variants, built only by tools/build_variants.sh) must keep compiling against the
    product's headers and keep defining every launcher csrc/qs_launch.h declares for qs_kernels.hip -- otherwise a variant
    library links with holes and fails only on the GPU box"""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("no hipcc")
    csrc = ROOT / "jpeg-quantsmooth_amd" / "csrc"
    obj = tmp_path / "exp.o"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
                    "-Wno-unused-function", f"-I{csrc}", "-c", str(ROOT / "tools" / "experiments" / "qs_kernels_r03.hip"), "-o", str(obj)],
                   check=True, capture_output=True, timeout=900)
    have = subprocess.run(["nm", "-C", "--defined-only", str(obj)], capture_output=True, text=True, check=True).stdout
    ship = subprocess.run(["nm", "-C", "--defined-only", str(csrc / "qs_kernels.o")], capture_output=True, text=True, check=True).stdout
    sig = lambda text: {line.split(" T ", 1)[1] for line in text.splitlines() if " T qs_launch_" in line}
    assert sig(ship) and sig(ship) <= sig(have), sig(ship) - sig(have)


def test_flag_values_match_reference_api(pkg):
    F = pkg.FLAGS
    # reference libjpegqs.h:14-32
    assert (F.DIAGONALS, F.JOINT_YUV, F.UPSAMPLE_UV, F.LOW_QUALITY) == (1, 2, 4, 8)
    assert (F.NO_REBALANCE, F.NO_REBALANCE_UV, F.TRANSCODE, F.MASK, F.ITER_MAX) == (16, 32, 64, 0x7F, 100)
    # reference quantsmooth.c:380-393
    assert [pkg.flags_for_quality(q) for q in range(7)] == [9, 11, 15, 0, 1, 3, 7]


def test_consts_tables_bit_exact(hip, oracle, synth):
    quant = synth.quality_table(synth.STD_LUMA, 37)
    quant[9] = 0  # zero multiplier -> treated as 1 (reference quantsmooth.h:2506-2511)
    for flags in (0, 1):
        blob = hip.consts_build(quant, flags)
        ints = blob[:64 * 4 * 9].view(np.int32).reshape(9, 64)
        nat, q, qraw, x1, x2 = ints[0], ints[1], ints[2], ints[3], ints[4]
        rng = blob[64 * 4 * 5:64 * 4 * 6].view(np.float32)
        ts = 272 if flags else 160
        off = 64 * 4 * 9
        assert blob[off:off + 4].view(np.int32)[0] == ts
        tab = blob[off + 64:off + 64 + 64 * 272 * 4].view(np.float32)[:64 * ts].reshape(64, ts)
        want = oracle.tables(flags)
        eff, _, _ = oracle.quant_prep(quant)
        for k in range(64):
            i = int(nat[k])
            assert np.array_equal(tab[k].view(np.uint32), want[i].view(np.uint32)), (flags, k)
            assert q[k] == eff[i] and rng[k] == float(2 * eff[i]) / 4096.0
        assert np.array_equal(qraw, quant.astype(np.int32))
        # reciprocal tables reproduce the exact-division interval
        import ctypes as C
        o = C.c_int(0)
        for k in range(1, 64):
            div = int(q[k])
            for c in (-3000, -div, -1, 0, 1, div // 2, div, 5 * div + 1, 3071):
                a = ((int(x1[k]) * c) >> 16) + c
                a = (((a << int(x2[k])) + 0x4000) >> 15) * div      # x2[k] holds the shift: -a * -(1 << sh)
                assert a == oracle.interval(c, div)[0]


def test_plane_geometry(hip):
    for wblk in (1, 7, 8, 240, 1024, 2048):
        pitch = hip.plane_pitch(wblk)
        assert pitch % 64 == 0 and pitch >= wblk * 8 + 17
        assert hip.plane_bytes(wblk, 3) >= pitch * (3 * 8 + 2)
        assert hip.plane_row_offset(wblk, -1) == 0 and hip.plane_row_offset(wblk, 0) == pitch


def test_early_outs_need_no_gpu(hip, synth):
    """niter <= 0 without upsampling returns 0 and touches nothing
    (reference quantsmooth.h:2455-2458) -- before any device is needed"""
    coef, quant = synth.synth_gray(64, 64, 50)
    res = hip.do_quantsmooth([coef], [quant], 0, 0)
    assert res["ret"] == 0 and np.array_equal(res["coefs"][0], coef) and np.array_equal(res["quants"][0], quant)
    res = hip.do_quantsmooth([coef], [quant], 1, -5)
    assert res["ret"] == 0 and np.array_equal(res["coefs"][0], coef)


def test_no_silent_cpu_fallback(pkg, hip, synth):
    """the C ABI of the hot path (libjpegqs_hip.so) has no CPU route at all: without a device the job layer must
    raise, not compute on the CPU.  (The reference-API library on top of it, libjpegqs.so, runs its own announced
    CPU back end in that case: tests/test_cpu_backend.py::test_cpu_fallback_is_announced_and_can_be_forbidden.)"""
    if hip.device_count() > 0:
        pytest.skip("a GPU is present; covered by the gpu tests")
    coef, quant = synth.synth_gray(64, 64, 50)
    with pytest.raises(pkg.QsHipError) as ei:
        hip.do_quantsmooth([coef], [quant], 0, 1)
    assert ei.value.code == -1  # QS_HIP_ENODEV


def test_bad_arguments_rejected(pkg, hip):
    with pytest.raises(pkg.QsHipError):
        hip.smooth_plane(0, 0, 0, 8, 8, 0)
    with pytest.raises(pkg.QsHipError):
        hip.idct_plane(1, 1, 1, 0, 8, 0, 1, 1, 1)


def test_missing_library_fails_loudly(pkg, tmp_path):
    with pytest.raises(FileNotFoundError):
        pkg.load_library(tmp_path / "libjpegqs_hip.so")


CSRC = ROOT / "jpeg-quantsmooth_amd" / "csrc"
JPEGINC = "/opt/conda/include"


@pytest.mark.parametrize("turbo_number", [3000090, 2001000, 2000090])
def test_turbo_branch_of_the_shim_compiles_and_matches_reference_layout(turbo_number):
    """decode mode after UPSAMPLE_UV patches libjpeg-turbo's PRIVATE master record (reference
    quantsmooth.h:44-60, 2864-2867).  This image has libjpeg 9d headers only, so the branch cannot run
    here; it is at least compiled -- against tests/stubs/turbo/jpeglib.h, which dresses the 9d header up as
    libjpeg-turbo of the given version -- with _Static_asserts on the record's layout, and, where the
    reference is mounted, every field offset is compared with the reference's own declaration."""
    import subprocess
    if not Path(JPEGINC, "jpeglib.h").exists():
        pytest.skip("no libjpeg headers")
    base = ["gcc", "-fsyntax-only", "-Werror=implicit-function-declaration", f"-I{ROOT / 'tests' / 'stubs' / 'turbo'}",
            f"-I{JPEGINC}", f"-DQS_STUB_TURBO_NUMBER={turbo_number}"]
    r = subprocess.run(base + [str(CSRC / "jpegqs_shim.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    if Path("/root/reference/quantsmooth.h").exists():
        r = subprocess.run(base + ["-I/root/reference", f"-I{CSRC}", "-DNO_SIMD", "-w",
                                   str(ROOT / "tests" / "stubs" / "turbo_layout_check.c")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_decode_mode_reports_a_backend_failure(hip):
    """jpegqs_start_decompress() with no usable GPU and the CPU back end forbidden (JPEGQS_BACKEND=hip): the
    reference API has no error return there (reference quantsmooth.h:2880-2895 ignores do_quantsmooth's result), so the
    failure is counted as a libjpeg warning and kept in jpegqs_hip_backend_status(); the demo program exits 3 instead
    of delivering unsmoothed pixels.  Without the variable the same call runs on the CPU back end and succeeds."""
    import os
    import subprocess
    demo = ROOT / "oracle" / "decode_hip"
    if not demo.exists():
        pytest.fail(f"{demo} not built (run __graft_entry__.build())")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1", JPEGQS_BACKEND="hip")     # hide every device
    r = subprocess.run([str(demo), "3", "2", str(ROOT / "tests" / "golden" / "cli" / "gray64.jpg")], capture_output=True, env=env)
    assert r.returncode == 3, (r.returncode, r.stderr.decode())
    assert b"no HIP device" in r.stderr and b"back end failed (status -1" in r.stderr
    assert r.stdout == b""
    env.pop("JPEGQS_BACKEND")
    r = subprocess.run([str(demo), "3", "2", str(ROOT / "tests" / "golden" / "cli" / "gray64.jpg")], capture_output=True, env=env)
    assert r.returncode == 0 and b"using the CPU back end" in r.stderr and len(r.stdout) > 0


def test_band_arithmetic_exported_from_c(hip):
    """the ONE definition of the band split both multi-GPU drivers use (csrc/qs_planes.cpp): bands tile the rows
    without gaps, colour bands keep luma and chroma on the same image rows, halo rows are the plane's first / last
    pixel rows and its two apron rows"""
    for hblk in (1, 7, 64, 135, 1024, 2048):
        for n in (1, 2, 3, 8):
            rows = [hip.band_rows(hblk, n, b) for b in range(n)]
            assert rows[0][0] == 0 and rows[-1][1] == hblk
            assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
            rows2 = [hip.band_rows(hblk, n, b, 2) for b in range(n)]
            assert rows2[0][0] == 0 and rows2[-1][1] == hblk and all(r[0] % 2 == 0 for r in rows2)
    for hby, hbc, vs in ((1024, 512, 2), (135, 68, 2), (135, 135, 1), (67, 34, 2)):
        for n in (1, 2, 5, 8):
            split = [hip.colour_band_rows(hby, hbc, vs, n, b) for b in range(n)]
            assert split[0][0] == 0 and split[0][2] == 0 and split[-1][1] == hby and split[-1][3] == hbc
            for (y0, y1, c0, c1), (y0n, _, c0n, _) in zip(split, split[1:]):
                assert y1 == y0n and c1 == c0n and y0 == min(c0 * vs, hby)
    with pytest.raises(Exception):
        hip.band_rows(10, 0, 0)
    for wblk, hblk in ((1, 1), (240, 17), (1024, 128)):
        st, sb, rt, rb, n = hip.band_halo_rows(wblk, hblk)
        assert n == hip.plane_pitch(wblk)
        assert (st, sb, rt, rb) == tuple(hip.plane_row_offset(wblk, y) for y in (0, hblk * 8 - 1, -1, hblk * 8))


def test_prewarm_is_harmless_without_a_device(pkg, hip, synth):
    """qs_hip_prewarm returns at once, never fails loudly, and the next job-layer call waits for it: with no GPU the
    call then reports QS_HIP_ENODEV as before (no hang, no crash); with one it simply works"""
    import ctypes as C
    assert hip.lib.qs_hip_prewarm(None, 0, 0) == 0
    coef, quant = synth.synth_gray(64, 64, 50)
    job, _ = hip._make_job([coef], [quant])
    assert hip.lib.qs_hip_prewarm(C.byref(job), 0, 3) == 0
    if hip.device_count() > 0:
        assert hip.do_quantsmooth([coef], [quant], 0, 1)["ret"] == 0
    else:
        with pytest.raises(pkg.QsHipError) as ei:
            hip.do_quantsmooth([coef], [quant], 0, 1)
        assert ei.value.code == -1


@pytest.mark.parametrize("progprec", [0, -1, 1, 7, 1000])
def test_progress_plan_equals_the_reference_call_sequence(hip, oracle, reference, synth, progprec):
    """CPU: the call sequence the pipelined GPU routes replay (ProgressPlan, exported as qs_hip_progress_calls: a
    function of the geometry alone) against the calls the oracle port AND the compiled reference actually make
    (quantsmooth.h:2474-2482, 2656-2664) -- gray, 4:2:0 and 4:4:4 with independent components, several niter"""
    cases = []
    c, q = synth.synth_gray(72, 40, 50, seed=2)
    cases.append(([c], [q], {}))
    for hs, vs in ((2, 2), (1, 1), (2, 1)):
        j = synth.synth_ycc(104, 72, hs, vs, quality=50, seed=5)
        cases.append((j["coefs"], j["quants"], dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(104, 72))))
    for coefs, quants, kw in cases:
        for niter in (1, 3, 4):
            want = None
            for impl in (oracle, reference):
                calls = []

                def cb(_u, cur, mx, calls=calls):
                    calls.append((cur, mx))
                    return 0
                impl.do_quantsmooth(coefs, quants, 1, niter, progprec=progprec, progress=cb, **kw)
                assert want is None or calls == want
                want = calls
            assert hip.progress_calls(coefs, quants, niter, progprec, **kw) == want, (len(coefs), niter, progprec)
