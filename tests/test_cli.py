"""The libjpeg-facing drop-in (libjpegqs.so) and the `jpegqs` CLI built on it:
JPEG file in -> JPEG file out, compared byte for byte with what the reference's
own CLI (scalar build) wrote for the same input (tests/golden/cli/*.ref.jpg,
produced by tests/golden/make_golden.py)."""
import ctypes as C
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "jpeg-quantsmooth_amd"
CLI = PKG / "jpegqs"
GOLD = ROOT / "tests" / "golden" / "cli"


def _need_cli():
    if not CLI.exists():
        pytest.fail(f"{CLI} not built (run __graft_entry__.build())")


def test_shim_exports_reference_api(hip):
    """libjpegqs.so exports exactly the three functions of reference libjpegqs.h:47-56"""
    import os
    jpeg = Path("/opt/conda/lib/libjpeg.so.9")
    if jpeg.exists():   # the decode-mode tail needs libjpeg's jinit_* entry points
        C.CDLL(str(jpeg), mode=os.RTLD_GLOBAL | os.RTLD_LAZY)
    lib = C.CDLL(str(PKG / "libjpegqs.so"), mode=os.RTLD_LAZY)
    for name in ("do_quantsmooth", "jpegqs_start_decompress", "jpegqs_finish_decompress"):
        assert getattr(lib, name) is not None


def test_cli_usage_and_exit_code():
    _need_cli()
    r = subprocess.run([str(CLI)], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage:" in r.stderr and "--quality" in r.stderr
    r = subprocess.run([str(CLI), "--bogus", "a", "b"], capture_output=True, text=True)
    assert r.returncode == 1
    r = subprocess.run([str(CLI), "-q", "3", "/nonexistent.jpg", "/tmp/x.jpg"], capture_output=True, text=True)
    assert r.returncode == 1 and "can't open input file" in r.stderr


def test_cli_without_gpu_fails_loudly_and_writes_nothing(hip, tmp_path):
    """JPEGQS_BACKEND=hip (the CPU back end forbidden; tests/test_cpu_backend.py covers the default): a back-end
    failure is an error exit (3) and no output file -- a script must not receive an unsmoothed image with a success status"""
    import os
    _need_cli()
    if hip.device_count() > 0:
        pytest.skip("GPU present")
    out = tmp_path / "o.jpg"
    r = subprocess.run([str(CLI), "-q", "3", "-i", "0", str(GOLD / "gray64.jpg"), str(out)],
                       capture_output=True, text=True, env=dict(os.environ, JPEGQS_BACKEND="hip"))
    assert "no HIP device" in r.stderr and "no output written" in r.stderr
    assert r.returncode == 3 and not out.exists()
    # --niter 0 takes the reference's early-out before any device is needed: plain transcode, exit 0
    r = subprocess.run([str(CLI), "-q", "3", "-n", "0", str(GOLD / "gray64.jpg"), str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0 and out.exists()


@pytest.mark.gpu
@pytest.mark.parametrize("src", ["gray64", "rgb141x93_420", "rgb141x93_444"])
@pytest.mark.parametrize("quality", [2, 3, 4, 5, 6])
def test_cli_matches_reference_cli_bytes(gpu, tmp_path, src, quality):
    _need_cli()
    out = tmp_path / "o.jpg"
    for extra in ([], ["--optimize"]):
        r = subprocess.run([str(CLI), "-q", str(quality), "-n", "3", "-i", "8", *extra,
                            str(GOLD / f"{src}.jpg"), str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "quantsmooth:" in r.stderr       # --info 8 timing line, reference quantsmooth.h:2820-2825
        if not extra:
            assert out.read_bytes() == (GOLD / f"{src}.q{quality}.ref.jpg").read_bytes()


@pytest.mark.gpu
@pytest.mark.parametrize("src,quality", [("cmyk96x64", 3), ("cmyk96x64", 4), ("cmyk96x64", 6),
                                         ("rgb120x88_prog", 3), ("rgb120x88_prog", 6),
                                         ("rgb120x88_422_rst", 2), ("rgb120x88_422_rst", 5), ("rgb120x88_422_rst", 6)])
def test_cli_container_variety_matches_reference_cli_bytes(gpu, tmp_path, src, quality):
    """four components (Adobe CMYK: every component is "luma" for the recovery loop, reference
    quantsmooth.h:2639), progressive scans in, 4:2:2 with restart markers"""
    _need_cli()
    out = tmp_path / "o.jpg"
    r = subprocess.run([str(CLI), "-q", str(quality), "-n", "3", "-i", "0", str(GOLD / f"{src}.jpg"), str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == (GOLD / f"{src}.q{quality}.ref.jpg").read_bytes()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,args", [("f33", ["-f", "33", "-n", "2"]), ("f20_n1", ["--flags", "20", "--niter", "1"]),
                                      ("c0", ["-q", "3", "-n", "3", "-c", "0"]), ("c1", ["-q", "3", "-n", "3", "--copy", "1"]),
                                      ("q5_n0", ["-q", "5", "-n", "0"]), ("q6_n1_o", ["-q", "6", "-n", "1", "-o"])])
def test_cli_option_cases_match_reference_cli_bytes(gpu, tmp_path, tag, args):
    """--flags override, --copy 0/1 (marker copying), niter 0, --optimize: same bytes as the reference CLI
    (the cases are tests/golden/make_golden.py's CLI_OPTION_CASES)"""
    _need_cli()
    out = tmp_path / "o.jpg"
    r = subprocess.run([str(CLI), *args, "-i", "0", str(GOLD / "rgb141x93_420.jpg"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == (GOLD / f"rgb141x93_420.{tag}.ref.jpg").read_bytes()


@pytest.mark.gpu
def test_cli_stdin_stdout_and_inplace(gpu, tmp_path):
    _need_cli()
    data = (GOLD / "gray64.jpg").read_bytes()
    r = subprocess.run([str(CLI), "-q4", "-n3", "-i0", "-", "-"], input=data, capture_output=True)
    assert r.returncode == 0 and r.stdout == (GOLD / "gray64.q4.ref.jpg").read_bytes()
    f = tmp_path / "same.jpg"; f.write_bytes(data)
    r = subprocess.run([str(CLI), "--quality", "4", "--niter", "3", "--info", "0", str(f), str(f)], capture_output=True)
    assert r.returncode == 0 and f.read_bytes() == (GOLD / "gray64.q4.ref.jpg").read_bytes()


@pytest.fixture(scope="module")
def gray8192_jpeg(tmp_path_factory, pkg):
    """BASELINE.md section 3's input for configs[2]: the synthetic 8192 x 8192 luma image of SURVEY.md 8d as a
    real libjpeg-encoded file (quality 50, baseline Huffman)"""
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    path = tmp_path_factory.mktemp("big") / "gray_8192.jpg"
    Image.fromarray(pkg.synth.synth_pixels(8192, 8192), "L").save(path, quality=50)
    return path


@pytest.mark.gpu
@pytest.mark.parametrize("quality", [3, 4])
def test_cli_8192_gray_matches_reference_cli_bytes(gpu, tmp_path, gray8192_jpeg, quality):
    """end to end at the headline size: JPEG in -> entropy decode -> do_quantsmooth on the GPU -> entropy encode,
    byte for byte what the reference's own CLI (scalar build, OpenMP) writes for the same file -- i.e. all
    1,048,576 blocks equal (reference quantsmooth.c:548-596, quantsmooth.h:1517-1549)"""
    _need_cli()
    ref_cli = ROOT / "oracle" / "_ref" / "jpegqs_ref_none"
    if not ref_cli.exists():
        pytest.skip("oracle/_ref/jpegqs_ref_none did not travel with the tree")
    ours, ref = tmp_path / "ours.jpg", tmp_path / "ref.jpg"
    r = subprocess.run([str(CLI), "-q", str(quality), "-i", "8", str(gray8192_jpeg), str(ours)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r2 = subprocess.run([str(ref_cli), "-q", str(quality), "-i", "0", str(gray8192_jpeg), str(ref)], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr
    a, b = ours.read_bytes(), ref.read_bytes()
    assert len(a) > 1 << 20 and a == b, f"outputs differ ({len(a)} vs {len(b)} bytes)"


# ---- decode-mode API (jpegqs_start_decompress / jpegqs_finish_decompress) -------
DECODE = ROOT / "oracle" / "decode_hip"   # oracle/decode_demo.c linked against the product's libjpegqs.so


@pytest.mark.gpu
@pytest.mark.parametrize("src,quality,niter", [("gray64", 4, 2), ("rgb141x93_420", 3, 3),
                                               ("rgb128x96_420", 3, 2), ("rgb128x96_420", 5, 2),
                                               ("rgb141x93_444", 6, 3), ("gray64", 6, 2), ("rgb141x93_444", 5, 2)])
def test_decode_mode_matches_reference_pixels(gpu, src, quality, niter):
    """jpeg_read_scanlines() after jpegqs_start_decompress() delivers the same
    pixels as the reference built into the same demo program (reference
    quantsmooth.h:2861-2905).  --quality 6 on images WITHOUT chroma subsampling
    (4:4:4, gray) takes the JOINT_YUV path and no upsampling: pixels must be equal.
    UPSAMPLE_UV on a subsampled image: see the next test."""
    if not DECODE.exists():
        pytest.fail(f"{DECODE} not built (run __graft_entry__.build())")
    r = subprocess.run([str(DECODE), str(quality), str(niter), str(GOLD / f"{src}.jpg")], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == (GOLD / f"{src}.q{quality}.dec.ref.raw").read_bytes()


@pytest.mark.gpu
@pytest.mark.parametrize("src,quality,niter", [("rgb141x93_420", 6, 3), ("rgb128x96_420", 6, 1)])
def test_decode_mode_after_upsample_behaves_like_the_reference(gpu, src, quality, niter):
    """UPSAMPLE_UV on a 4:2:0 image in decode mode: the reference itself, compiled against the libjpeg 9d of
    this image, dies inside jinit_upsampler with libjpeg's "Fractional sampling not implemented yet" (the
    re-initialisation of reference quantsmooth.h:2861-2876 targets libjpeg-turbo / libjpeg <= 8).  The product runs
    the same libjpeg call sequence after its GPU pass, so it must end the same way: same message, same status."""
    want_rc, want_err = (GOLD / f"{src}.q{quality}.dec.abort.txt").read_text().split("\n", 1)
    r = subprocess.run([str(DECODE), str(quality), str(niter), str(GOLD / f"{src}.jpg")], capture_output=True)
    got_err = "\n".join(l for l in r.stderr.decode().splitlines() if "amdgpu.ids" not in l)
    assert r.returncode == int(want_rc)
    assert got_err.strip() == want_err.strip()


@pytest.mark.gpu
def test_cli_info_output_matches_reference(gpu, tmp_path):
    """default --info 15: component table, quantisation tables, plane sizes and the
    timing line, as the reference prints them (reference quantsmooth.h:2422-2445,
    2569-2572, 2820-2825); only the measured time differs"""
    _need_cli()
    r = subprocess.run([str(CLI), "-q", "3", "-n", "1", str(GOLD / "rgb141x93_420.jpg"), str(tmp_path / "o.jpg")],
                       capture_output=True, text=True)
    assert r.returncode == 0
    want = (GOLD / "rgb141x93_420.q3.info.ref.txt").read_text().splitlines()
    got = [l for l in r.stderr.splitlines() if "amdgpu.ids" not in l]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        if w.startswith("quantsmooth:"):
            assert g.startswith("quantsmooth: ") and g.endswith("ms")
        else:
            assert g == w


# ---- --verbose banner, exit code 2, and the reference's OWN front-ends on the library ---------------------------------------
REFDIR = ROOT / "oracle" / "_ref"


def _built(name):
    p = REFDIR / name
    if not p.exists():
        pytest.skip(f"{p} did not travel with the tree (built by `make -C oracle ref dropin` where /root/reference is mounted)")
    return p


def _no_noise(text):
    """stderr minus what is not the CLI's: the CPU back end's announcement (no-GPU boxes), a ROCm data-file notice"""
    return [l for l in text.splitlines() if "using the CPU back end" not in l and "amdgpu.ids" not in l]


def test_cli_verbose_banner_equals_the_reference_cli(tmp_path):
    """reference quantsmooth.c:405-444: `--verbose n` (n > 0) prints which libjpeg this is, then hands n - 1 to libjpeg's
    trace level; without file arguments it stops there with status 1 and NO usage text.  Line for line what the
    reference's own CLI prints (both link the same libjpeg)."""
    _need_cli()
    ref_cli = _built("jpegqs_ref_none")
    for args in (["-v", "1"], ["--verbose", "3"], ["-v2"]):
        a = subprocess.run([str(CLI), *args], capture_output=True, text=True)
        b = subprocess.run([str(ref_cli), *args], capture_output=True, text=True)
        assert a.returncode == b.returncode == 1
        assert a.stderr == b.stderr and "Compiled with libjpeg" in a.stderr and "Usage" not in a.stderr
    # with files: banner + libjpeg's own trace output at level n - 1, identical too
    for level in ("1", "2"):
        oa, ob = tmp_path / "a.jpg", tmp_path / "b.jpg"
        a = subprocess.run([str(CLI), "-v", level, "-q", "3", "-n", "0", "-i", "0", str(GOLD / "gray64.jpg"), str(oa)], capture_output=True, text=True)
        b = subprocess.run([str(ref_cli), "-v", level, "-q", "3", "-n", "0", "-i", "0", str(GOLD / "gray64.jpg"), str(ob)], capture_output=True, text=True)
        assert a.returncode == b.returncode == 0
        assert _no_noise(a.stderr) == _no_noise(b.stderr)
        assert oa.read_bytes() == ob.read_bytes()
    # -v 0 is silent, and the usage text still comes for a wrong argument count
    r = subprocess.run([str(CLI), "-v", "0"], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage:" in r.stderr and "Compiled with" not in r.stderr


def _truncated(tmp_path, src, frac=0.6):
    data = (GOLD / f"{src}.jpg").read_bytes()
    p = tmp_path / f"{src}.trunc.jpg"
    p.write_bytes(data[: int(len(data) * frac)])
    return p


def _exit2_case(exe, ref_cli, tmp_path, env=None):
    for src in ("gray64", "rgb141x93_420"):
        t = _truncated(tmp_path, src)
        for quality in ("3", "6"):
            oa, ob = tmp_path / "a.jpg", tmp_path / "b.jpg"
            a = subprocess.run([str(exe), "-q", quality, "-i", "0", str(t), str(oa)], capture_output=True, text=True, env=env)
            b = subprocess.run([str(ref_cli), "-q", quality, "-i", "0", str(t), str(ob)], capture_output=True, text=True)
            assert a.returncode == b.returncode == 2, (a.returncode, b.returncode, a.stderr)
            assert "Premature end of JPEG file" in a.stderr
            assert _no_noise(a.stderr) == _no_noise(b.stderr)
            assert oa.read_bytes() == ob.read_bytes(), (src, quality)


def test_cli_exit_code_2_on_libjpeg_warnings(tmp_path):
    """reference quantsmooth.c:626: a file libjpeg warns about (here: cut off at 60 %) is still processed and written,
    with exit status 2 -- same status, same stderr and same output bytes as the reference's CLI.  Runs on whatever back
    end this box has (the GPU box: the GPU; the build container: the CPU back end)."""
    _need_cli()
    _exit2_case(CLI, _built("jpegqs_ref_none"), tmp_path)


@pytest.mark.gpu
def test_gpu_cli_exit_code_2_on_libjpeg_warnings(gpu, tmp_path):
    import os
    _need_cli()
    _exit2_case(CLI, _built("jpegqs_ref_none"), tmp_path, env=dict(os.environ, JPEGQS_BACKEND="hip"))
    _exit2_case(_built("jpegqs_dropin"), _built("jpegqs_ref_none"), tmp_path, env=dict(os.environ, JPEGQS_BACKEND="hip"))


CLI_GOLDEN_CASES = ([(src, ["-q", str(q), "-n", "3"], f"{src}.q{q}.ref.jpg")
                     for src in ("gray64", "rgb141x93_420", "rgb141x93_444") for q in (2, 3, 4, 5, 6)]
                    + [(src, ["-q", str(q), "-n", "3"], f"{src}.q{q}.ref.jpg")
                       for src, q in (("cmyk96x64", 3), ("cmyk96x64", 4), ("cmyk96x64", 6), ("rgb120x88_prog", 3),
                                      ("rgb120x88_prog", 6), ("rgb120x88_422_rst", 2), ("rgb120x88_422_rst", 5), ("rgb120x88_422_rst", 6))]
                    + [("rgb141x93_420", args, f"rgb141x93_420.{tag}.ref.jpg")
                       for tag, args in (("f33", ["-f", "33", "-n", "2"]), ("f20_n1", ["--flags", "20", "--niter", "1"]),
                                         ("c0", ["-q", "3", "-n", "3", "-c", "0"]), ("c1", ["-q", "3", "-n", "3", "--copy", "1"]),
                                         ("q5_n0", ["-q", "5", "-n", "0"]), ("q6_n1_o", ["-q", "6", "-n", "1", "-o"]))])


@pytest.mark.gpu
def test_gpu_reference_cli_on_the_library_writes_the_reference_bytes(gpu, tmp_path):
    """THE drop-in claim, literally: the reference's own UNMODIFIED quantsmooth.c, compiled from where it lies and linked
    against libjpegqs.so + libjpegqs_hip.so (oracle/Makefile `dropin`; JPEGQS_BACKEND=hip: the GPU or nothing), writes
    byte for byte what the same source writes on top of the reference's own implementation -- every CLI golden, --quality
    2..6, every option case"""
    import os
    exe = _built("jpegqs_dropin")
    env = dict(os.environ, JPEGQS_BACKEND="hip")
    out = tmp_path / "o.jpg"
    for src, args, ref in CLI_GOLDEN_CASES:
        r = subprocess.run([str(exe), *args, "-i", "16", str(GOLD / f"{src}.jpg"), str(out)], capture_output=True, text=True, env=env)
        assert r.returncode == 0, (src, args, r.stderr)
        assert "SIMD type: hip/gfx950" in r.stderr and "CPU back end" not in r.stderr
        assert out.read_bytes() == (GOLD / ref).read_bytes(), (src, args)
        out.unlink()


@pytest.mark.gpu
def test_gpu_reference_example_program_on_the_library(gpu, tmp_path):
    """the reference's unmodified example.c (decode mode through jpegqs_start_decompress, the q6 flag set, niter 3 and a
    progress callback that prints percentages -- example.c:96, 137-149) on the library vs on the reference itself: same
    BMP bytes, same stdout (the callback sequence), same exit status"""
    import os
    ours, theirs = _built("example_dropin"), _built("example_ref_none")
    env = dict(os.environ, JPEGQS_BACKEND="hip")
    for src in ("gray64", "rgb141x93_444", "rgb141x93_420", "rgb120x88_422_rst", "cmyk96x64", "rgb128x96_420"):
        a, b = tmp_path / "a.bmp", tmp_path / "b.bmp"
        ra = subprocess.run([str(ours), str(GOLD / f"{src}.jpg"), str(a)], capture_output=True, env=env)
        rb = subprocess.run([str(theirs), str(GOLD / f"{src}.jpg"), str(b)], capture_output=True)
        assert ra.returncode == rb.returncode, (src, ra.stderr)   # (libjpeg 9 cannot decode after UPSAMPLE_UV of 4:2:0: both 1)
        assert ra.stdout == rb.stdout, src
        assert b"CPU back end" not in ra.stderr
        assert a.exists() == b.exists()
        if a.exists():
            assert a.read_bytes() == b.read_bytes(), src
            a.unlink(); b.unlink()
