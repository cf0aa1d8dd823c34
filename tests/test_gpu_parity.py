"""GPU parity tests: the HIP path (through the flat C ABI) must be BIT-EXACT
against the oracle / golden vectors for integer JCOEF output."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

from helpers import GPU_FLAG_MASK_UNSUPPORTED, assert_same_result, golden_names, inject_extreme_blocks, load_golden

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _supported(flags):
    return not (flags & GPU_FLAG_MASK_UNSUPPORTED)


@pytest.mark.parametrize("name", [n for n in golden_names()])
def test_gpu_matches_golden(gpu, pkg, name):
    job, want = load_golden(name)
    if not _supported(job["flags"]):
        with pytest.raises(pkg.QsHipError) as ei:
            gpu.do_quantsmooth(job["coefs"], job["quants"], job["flags"], job["niter"], **job["kw"])
        assert ei.value.code == -4, "unsupported flags must fail loudly with QS_HIP_ENOTSUP"
        return
    got = gpu.do_quantsmooth(job["coefs"], job["quants"], job["flags"], job["niter"], **job["kw"])
    assert_same_result(got, want, name)


@pytest.mark.parametrize("w,h,qual", [(64, 64, 50), (8, 8, 50), (72, 8, 30), (8, 200, 75), (520, 264, 50),
                                      (1000, 40, 15), (24, 88, 95)])
@pytest.mark.parametrize("flags", [0, 1, 16, 1 | 16])
def test_gpu_vs_oracle_gray(gpu, oracle, synth, w, h, qual, flags):
    coef, quant = synth.synth_gray(w, h, qual, seed=w * 31 + h)
    for niter in (1, 3):
        a = gpu.do_quantsmooth([coef], [quant], flags, niter)
        b = oracle.do_quantsmooth([coef], [quant], flags, niter)
        assert_same_result(a, b, f"{w}x{h} q{qual} flags={flags} niter={niter}")


def test_gpu_vs_oracle_colour_independent_components(gpu, oracle, synth):
    """BASELINE.json configs[1] shape at reduced size: 4:2:0 YCbCr, q=3 / q=4,
    components are independent when JOINT_YUV is off; NO_REBALANCE_UV touches chroma only"""
    j = synth.synth_ycc(333, 517, 2, 2, quality=40)
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(333, 517))
    for flags in (0, 1, 32, 1 | 32):
        a = gpu.do_quantsmooth(j["coefs"], j["quants"], flags, 3, **kw)
        b = oracle.do_quantsmooth(j["coefs"], j["quants"], flags, 3, **kw)
        assert_same_result(a, b, f"ycc420 flags={flags}")


@pytest.mark.parametrize("size,samp", [((64, 64), (2, 2)), ((333, 517), (2, 2)), ((321, 100), (2, 2)),
                                       ((129, 65), (1, 1)), ((100, 60), (2, 1)), ((90, 70), (1, 2)),
                                       ((8, 8), (2, 2)), ((200, 64), (4, 1))])
def test_gpu_vs_oracle_colour_all_flags(gpu, oracle, synth, size, samp):
    """every --quality level on YCbCr: JOINT_YUV predictor, luma downsample,
    UPSAMPLE_UV + re-FDCT, LOW_QUALITY; all chroma layouts incl. odd sizes"""
    w, h = size
    j = synth.synth_ycc(w, h, samp[0], samp[1], quality=40, seed=99)
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(w, h))
    for flags in (3, 7, 5, 2, 9, 10, 11, 15, 7 | 16, 7 | 32, 4):
        for niter in (0, 2):
            a = gpu.do_quantsmooth(j["coefs"], j["quants"], flags, niter, **kw)
            b = oracle.do_quantsmooth(j["coefs"], j["quants"], flags, niter, **kw)
            assert_same_result(a, b, f"{size} {samp} flags={flags} niter={niter}")


@pytest.mark.parametrize("size,samp", [((200, 136), (2, 2)), ((136, 88), (1, 1)), ((176, 72), (2, 1))])
def test_gpu_colour_refresh_pass_sees_unclamped_coefficients(gpu, oracle, synth, size, samp):
    """the reference clamps to +-1023 after its iteration loop, so the refresh-only pass A that
    feeds JOINT_YUV / UPSAMPLE_UV is the IDCT of unclamped coefficients (found on a 1080p frame:
    the job layer used to clamp in the last pass B, i.e. before that refresh)"""
    w, h = size
    j = inject_extreme_blocks(synth.synth_ycc(w, h, samp[0], samp[1], quality=60, seed=5))
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(w, h))
    for flags in (3, 7, 15, 11, 6):
        for niter in (1, 3):
            a = gpu.do_quantsmooth(j["coefs"], j["quants"], flags, niter, **kw)
            b = oracle.do_quantsmooth(j["coefs"], j["quants"], flags, niter, **kw)
            assert b["ret"] == 0
            assert_same_result(a, b, f"{size} {samp} flags={flags} niter={niter}")


@pytest.mark.parametrize("env", [{}, {"QS_HIP_DP": "0"}, {"QS_HIP_DP_GROUPS": "100000"}],
                         ids=["launcher's choice", "one-block-per-lane", "diagonal-parallel"])
def test_gpu_refresh_skip_content(env):
    """The need_refresh skip (reference quantsmooth.h:1407-1409; wave-uniform in the streaming kernel, workgroup-uniform
    in the small-plane kernel) only fires when NO block of a wave changes a coefficient during an anti-diagonal -- which
    the noisy synthetic images of the other tests almost never allow.  Content on which it fires all the time: flat
    planes, gentle gradients, flat planes with a few textured blocks (skipping and non-skipping waves side by side, and
    waves in which a single lane keeps the refresh alive), at several JPEG qualities, gray and 4:2:0, both kernels."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
import jpegqs_pkg
from oracle.oracle import Oracle
from helpers import assert_same_result
pkg = jpegqs_pkg.load(); hip = pkg.HipQS(); O = Oracle(); S = pkg.synth
rng = np.random.default_rng(5)
def planes(w, h):
    x = np.arange(w, dtype=np.float32)[None, :]; y = np.arange(h, dtype=np.float32)[:, None]
    flat = np.full((h, w), 117.0, np.float32)
    grad = 40.0 + 120.0 * x / w + 30.0 * np.sin(y / 97.0)
    spots = flat.copy()
    for _ in range(max(3, w * h // 40000)):
        by, bx = int(rng.integers(0, h // 8)), int(rng.integers(0, w // 8))
        spots[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] += rng.normal(0, 25, (8, 8))
    steps = 128.0 + 60.0 * ((x // 64 + y // 48) % 2)
    return {"flat": flat, "gradient": grad, "flat+textured blocks": spots, "steps": steps + 0 * y}
for (w, h) in ((512, 136), (1024, 512), (2048, 1040)):
    for name, img in planes(w, h).items():
        pix = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        for qual in (30, 75, 95):
            qt = S.quality_table(S.STD_LUMA, qual)
            coef = S.quantise_plane(pix, qt)
            for flags in (0, 1):
                a = hip.do_quantsmooth([coef], [qt], flags, 3)
                b = O.do_quantsmooth([coef], [qt], flags, 3, threads=8)
                assert_same_result(a, b, f"{name} {w}x{h} q{qual} flags={flags}")
# 4:2:0 colour, coupled flags: chroma planes are the smoothest content there is
j = S.synth_ycc(640, 360, 2, 2, quality=60, seed=2)
j["coefs"][1][:] = 0; j["coefs"][2][:, :, 1:] = 0
kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(640, 360))
for flags in (0, 1, 3, 7):
    assert_same_result(hip.do_quantsmooth(j["coefs"], j["quants"], flags, 3, **kw),
                       O.do_quantsmooth(j["coefs"], j["quants"], flags, 3, threads=8, **kw), f"ycc flags={flags}")
print("ok")
'''
    assert "ok" in _run_py(code, env)


def test_gpu_low_quality_gray(gpu, oracle, synth):
    for (w, h, qual) in ((64, 64, 50), (200, 120, 20), (24, 88, 92)):
        coef, quant = synth.synth_gray(w, h, qual, seed=5)
        for flags in (8, 9, 8 | 16):
            a = gpu.do_quantsmooth([coef], [quant], flags, 3)
            b = oracle.do_quantsmooth([coef], [quant], flags, 3)
            assert_same_result(a, b, f"{w}x{h} q{qual} flags={flags}")


def test_gpu_hostile_inputs(gpu, oracle, synth):
    """saturated / degenerate blocks: the a3 == 0 (NaN -> INT_MIN) path, huge
    ratios, zero quantisers, coefficients at the range limits"""
    rng = np.random.default_rng(3)
    coef = rng.integers(-1, 2, (6, 9, 64)).astype(np.int16) * rng.integers(0, 1000, (6, 9, 64)).astype(np.int16)
    quant = np.full(64, 2, np.uint16)
    coef = np.clip(coef, -1023, 1023).astype(np.int16)
    for flags in (0, 1):
        a = gpu.do_quantsmooth([coef], [quant], flags, 2)
        b = oracle.do_quantsmooth([coef], [quant], flags, 2)
        assert_same_result(a, b, f"random q=2 flags={flags}")
    # checkerboard pixels: every difference >= range
    from scipy.fft import dctn
    yy, xx = np.mgrid[0:32, 0:48]
    pix = np.where((yy + xx) & 1, 255.0, 0.0) - 128
    blocks = pix.reshape(4, 8, 6, 8).transpose(0, 2, 1, 3)
    c = np.rint(dctn(blocks, axes=(2, 3), norm="ortho") / 2).astype(np.int16).reshape(4, 6, 64)
    for flags in (0, 1):
        a = gpu.do_quantsmooth([c], [quant], flags, 2)
        b = oracle.do_quantsmooth([c], [quant], flags, 2)
        assert_same_result(a, b, f"checkerboard flags={flags}")
    qz = synth.quality_table(synth.STD_LUMA, 10); qz[1] = 0; qz[8] = 0
    coef2, _ = synth.synth_gray(96, 64, 10)
    a = gpu.do_quantsmooth([coef2], [qz], 1, 2)
    b = oracle.do_quantsmooth([coef2], [qz], 1, 2)
    assert_same_result(a, b, "zero quantisers")


def test_gpu_progress_and_cancel(gpu, oracle, synth):
    coef, quant = synth.synth_gray(64, 64, 50)
    for cancel_at in (None, 0, 2):
        logs = []
        for impl in (gpu, oracle):
            calls = []

            def cb(_u, cur, mx, calls=calls):
                calls.append((cur, mx))
                return 1 if cancel_at is not None and len(calls) - 1 == cancel_at else 0
            res = impl.do_quantsmooth([coef], [quant], 1, 4, progprec=0, progress=cb)
            logs.append((calls, res))
        assert logs[0][0] == logs[1][0]
        assert_same_result(logs[0][1], logs[1][1], f"cancel_at={cancel_at}")


_PROGRESS_CODE = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import jpegqs_pkg
from helpers import assert_same_result
from oracle.oracle import Oracle
pkg = jpegqs_pkg.load(); gpu = pkg.HipQS(); oracle = Oracle()
mode = sys.argv[1]
if mode == "colour":                      # three independent components (4:2:0, --quality 3 flags): one plane set
    j = pkg.synth.synth_ycc(208, 144, 2, 2, quality=40, seed=3)
    coefs, quants = j["coefs"], j["quants"]
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(208, 144))
else:
    c, q = pkg.synth.synth_gray(264, 328, 50, seed=4)
    coefs, quants, kw = [c], [q], {}
niter = 4
for progprec in (0, -1, 7):
    # how many calls does the reference make?  (then: no cancel, and a cancel at every one of them)
    n_calls = []
    def count(_u, cur, mx): n_calls.append((cur, mx)); return 0
    want_full = oracle.do_quantsmooth(coefs, quants, 1, niter, progprec=progprec, progress=count, **kw)
    assert len(n_calls) >= 2
    cancels = [None] + (list(range(len(n_calls))) if progprec == 0 else [0, len(n_calls) - 1])
    for cancel_at in cancels:
        logs = []
        for impl in (gpu, oracle):
            calls = []
            def cb(_u, cur, mx, calls=calls):
                calls.append((cur, mx))
                return 1 if cancel_at is not None and len(calls) - 1 == cancel_at else 0
            res = impl.do_quantsmooth(coefs, quants, 1, niter, progprec=progprec, progress=cb, **kw)
            logs.append((calls, res))
        assert logs[0][0] == logs[1][0], (mode, progprec, cancel_at, logs[0][0], logs[1][0])
        assert_same_result(logs[0][1], logs[1][1], "%%s progprec=%%d cancel_at=%%s" %% (mode, progprec, cancel_at))
# a coefficient that fails the range check (reference quantsmooth.h:2599-2610): the reference leaves the component before
# its first progress call -- the pipelined routes read the flags right behind pass A and make no call of their own
for ci in range(len(coefs)):
    bad = [c.copy() for c in coefs]
    bad[ci][1, 2, 0] = 0x7ff
    logs = []
    for impl in (gpu, oracle):
        calls = []
        def cb(_u, cur, mx, calls=calls):
            calls.append((cur, mx)); return 0
        res = impl.do_quantsmooth(bad, quants, 1, niter, progprec=-1, progress=cb, **kw)
        logs.append((calls, res))
    assert logs[0][0] == logs[1][0], (mode, "bad coefficient in component", ci, logs[0][0][:6], logs[1][0][:6])
    assert logs[1][1]["ret"] == 1 and (ci > 0 or not logs[1][0])
    assert_same_result(logs[0][1], logs[1][1], "%%s bad coefficient in component %%d" %% (mode, ci))
print("PROGRESS_OK")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("mode,env,route", [
    ("gray", {}, "fused"),
    ("colour", {}, "fused"),
    ("gray", {"QS_HIP_SPLIT_BLOCKS": "60", "QS_HIP_BAND_BLOCKS": "40"}, "fused"),              # the plane cut into pipelined bands
    ("gray", {"QS_HIP_DEVICES": "0,0,0", "QS_HIP_SHARD_MIN_BLOCKS": "1"}, "sharded(set)"),      # three logical devices
    ("colour", {"QS_HIP_DEVICES": "0,0", "QS_HIP_SHARD_MIN_BLOCKS": "1"}, "sharded(set)"),
], ids=["plane set", "plane set, three components", "pipelined bands", "three logical devices", "two logical devices, three components"])
def test_gpu_progress_callback_on_the_fast_routes(mode, env, route):
    """A progress callback (reference quantsmooth.h:2474-2482, 2656-2664; example.c:137-149 installs one) no longer
    sends the job to the host-synchronised single-device route: the plane-set route, its pipelined bands and the
    multi-device route make the reference's calls -- same arguments, same order -- as the work completes, and a cancel
    at ANY call gives the reference's result (every call index at the default precision, first and last at two others).
    QS_HIP_TRACE proves which route served the callback."""
    import subprocess
    e = dict(os.environ, QS_HIP_TRACE="1", **env)
    r = subprocess.run([sys.executable, "-c", _PROGRESS_CODE % (str(ROOT), str(ROOT / "tests")), mode],
                       capture_output=True, text=True, timeout=900, cwd=str(ROOT), env=e)
    assert r.returncode == 0 and "PROGRESS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    served = [l for l in r.stderr.splitlines() if l.startswith(f"qs_hip trace: {route}") and "progress: callback served from this route" in l]
    cancelled = [l for l in r.stderr.splitlines() if "cancelled by the" in l and "callback" in l]
    assert served, r.stderr[-3000:]
    assert cancelled, r.stderr[-3000:]


def test_gpu_plane_layer_device_resident(gpu, oracle, synth):
    """the device-pointer entry points (what bench.py times): same result as
    the job layer, inputs resident in HBM, explicit stream"""
    import torch
    coef, quant = synth.synth_gray(256, 192, 50)
    hb, wb = coef.shape[:2]
    dev = torch.device("cuda:0")
    for flags in (0, 1):
        d_coef = torch.from_numpy(coef.copy()).to(dev)
        d_cst = torch.from_numpy(gpu.consts_build(quant, flags)).to(dev)
        d_plane = torch.empty(gpu.plane_bytes(wb, hb), dtype=torch.uint8, device=dev)
        d_status = torch.zeros(1, dtype=torch.int32, device=dev)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            for it in range(3):
                gpu.idct_plane(d_cst.data_ptr(), d_coef.data_ptr(), d_plane.data_ptr(), wb, hb,
                               it == 0, 1, 1, d_status.data_ptr(), stream.cuda_stream)
                gpu.smooth_plane(d_cst.data_ptr(), d_coef.data_ptr(), d_plane.data_ptr(), wb, hb,
                                 flags, 1, it == 2, stream.cuda_stream)
        stream.synchronize()
        assert int(d_status.item()) == 0
        want = oracle.do_quantsmooth([coef], [quant], flags, 3)["coefs"][0]
        assert np.array_equal(d_coef.cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"QS_HIP_DP": "0"}], ids=["launcher's choice (small-plane kernel + pass A behind it)", "one block per lane (fused epilogue)"])
@pytest.mark.parametrize("size", [(256, 192), (1016, 520), (2048, 1032)])
def test_gpu_fused_pass_a_equals_separate_launches(size, env):
    """qs_hip_smooth_plane_next (pass B that also writes the next iteration's pixel plane) against
    qs_hip_smooth_plane followed by qs_hip_idct_plane: same coefficients, and the next plane byte for byte --
    interior, apron columns, replicated apron rows, and with the halo-side apron rows LEFT ALONE when rep_top /
    rep_bot are 0 (bands).  Also with the +-1023 clamp riding on the fused launch: the pixels must be those of the
    UNCLAMPED coefficients (reference :2668-2689 clamps behind the refresh pass).  Fresh process per kernel form."""
    code = r"""
import sys, numpy as np, torch
import jpegqs_pkg
pkg = jpegqs_pkg.load(); gpu = pkg.HipQS()
w, h = %d, %d
coef, quant = pkg.synth.synth_gray(w, h, 50, seed=5)
coef = coef.copy(); coef[::3, ::5, 1:9] *= 6          # some coefficients beyond +-1023 after dequantisation: the clamp matters
quant = quant.copy(); quant[1:9] = np.minimum(quant[1:9] * 3, 255)
hb, wb = coef.shape[:2]
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
for flags in (0, 1):
    for rep_top, rep_bot, clamp in ((1, 1, 0), (0, 1, 0), (1, 0, 1), (0, 0, 1)):
        cst = torch.from_numpy(gpu.consts_build(quant, flags)).to(dev)
        st = torch.zeros(1, dtype=torch.int32, device=dev)
        def fresh():
            c = torch.from_numpy(coef.copy()).to(dev)
            p = torch.full((gpu.plane_bytes(wb, hb),), 77, dtype=torch.uint8, device=dev)
            gpu.idct_plane(cst.data_ptr(), c.data_ptr(), p.data_ptr(), wb, hb, 1, 1, 1, st.data_ptr(), s)
            return c, p
        # separate launches: pass B (no clamp), pass A into a second plane, then the clamp
        c1, p1 = fresh()
        n1 = torch.full_like(p1, 201)
        gpu.smooth_plane(cst.data_ptr(), c1.data_ptr(), p1.data_ptr(), wb, hb, flags, 1, 0, s)
        gpu.idct_plane(cst.data_ptr(), c1.data_ptr(), n1.data_ptr(), wb, hb, 0, rep_top, rep_bot, st.data_ptr(), s)
        if clamp:
            gpu.clamp_plane(c1.data_ptr(), wb, hb, s)
        # fused
        c2, p2 = fresh()
        n2 = torch.full_like(p2, 201)
        gpu.smooth_plane_next(cst.data_ptr(), c2.data_ptr(), p2.data_ptr(), n2.data_ptr(), wb, hb, flags, 1, clamp, rep_top, rep_bot, s)
        torch.cuda.synchronize()
        assert torch.equal(c1, c2), ("coefficients", flags, rep_top, rep_bot, clamp)
        assert torch.equal(p1, p2), "the current plane must not be written"
        assert torch.equal(n1, n2), ("next plane", flags, rep_top, rep_bot, clamp, int((n1 != n2).sum()))
        if clamp:
            assert int(c2.abs().max()) <= 1023
print("ok")
""" % (size[0], size[1])
    assert "ok" in _run_py(code, env)


def test_gpu_large_plane_properties(gpu, oracle, synth):
    """2048x2048 (65,536 blocks): too slow to check everywhere on the scalar
    oracle in CI, so check (a) a band of block rows exactly, using the fact that
    a block's result after n iterations depends only on blocks within n of it,
    and (b) size-independent invariants: every coefficient stays inside the
    quantisation interval of its input, |coef| <= 1023, DC untouched by the loop"""
    coef, quant = synth.synth_gray(2048, 2048, 50, seed=77)
    niter = 2
    got = gpu.do_quantsmooth([coef], [quant], 1, niter)["coefs"][0]
    # (a) exact check of block rows 100..103 from a cropped plane with margin
    lo, hi = 100 - niter - 1, 104 + niter + 1
    sub = oracle.do_quantsmooth([coef[lo:hi]], [quant], 1, niter)["coefs"][0]
    assert np.array_equal(got[100:104], sub[100 - lo:104 - lo])
    # (b) invariants
    q = quant.astype(np.int32)
    deq = coef.astype(np.int32) * q
    g = got.astype(np.int32)
    assert np.abs(g).max() <= 1023
    half = q // 2
    inside = np.abs(g - deq) <= half
    clamped = np.abs(g) == 1023
    assert (inside | clamped).all()


def test_gpu_other_colour_spaces_and_missing_tables(gpu, oracle, synth):
    """4 components (CMYK-like, colorspace != YCbCr => every component is "luma" for
    rebalance, reference quantsmooth.h:2639), RGB JPEGs, a component without a
    quantisation table (skipped, reference :2493)"""
    j = synth.synth_ycc(96, 72, 1, 1, quality=45, seed=2)
    extra, q4 = synth.synth_gray(96, 72, 45, seed=8)
    for cs, coefs, quants in ((4, j["coefs"] + [extra], j["quants"] + [q4]),      # JCS_CMYK
                              (2, j["coefs"], j["quants"]),                        # JCS_RGB
                              (3, j["coefs"], [j["quants"][0], None, j["quants"][2]])):
        n = len(coefs)
        kw = dict(hsamp=[1] * n, vsamp=[1] * n, colorspace=cs, image_size=(96, 72))
        for flags in (0, 7, 32 | 1, 15):
            a = gpu.do_quantsmooth(coefs, quants, flags, 2, **kw)
            b = oracle.do_quantsmooth(coefs, quants, flags, 2, **kw)
            assert_same_result(a, b, f"colorspace={cs} flags={flags}")


def test_gpu_iteration_limits(gpu, oracle, synth):
    """niter is clamped to [0, 100] (reference :2455-2456); niter 0 with UPSAMPLE_UV
    still upsamples (reference :2458)"""
    coef, quant = synth.synth_gray(24, 16, 50)
    a = gpu.do_quantsmooth([coef], [quant], 0, 250)
    b = oracle.do_quantsmooth([coef], [quant], 0, 250)
    assert_same_result(a, b, "niter=250")
    j = synth.synth_ycc(40, 24, 2, 2, quality=50)
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(40, 24))
    a = gpu.do_quantsmooth(j["coefs"], j["quants"], 7, 0, **kw)
    b = oracle.do_quantsmooth(j["coefs"], j["quants"], 7, 0, **kw)
    assert a["up"] and b["up"]
    assert_same_result(a, b, "niter=0 upsample")


def test_gpu_job_layer_is_thread_safe(gpu, oracle, synth):
    """concurrent do_quantsmooth() calls from host threads (a serving process):
    each call leases its own streams and pooled device buffers; results stay bit-exact"""
    import threading
    j = synth.synth_ycc(320, 200, 2, 2, quality=50, seed=31)
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(320, 200))
    want = {fl: oracle.do_quantsmooth(j["coefs"], j["quants"], fl, 2, **kw) for fl in (0, 7)}
    errors = []

    def worker(t):
        try:
            for n in range(6):
                fl = (0, 7)[(t + n) & 1]
                got = gpu.do_quantsmooth(j["coefs"], j["quants"], fl, 2, **kw)
                assert_same_result(got, want[fl], f"thread {t} job {n} flags={fl}")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    gpu.lib.qs_hip_release_cache()


def test_gpu_fuzz_random_jobs(gpu, oracle, synth):
    """seeded random jobs: size, chroma layout, JPEG quality (incl. the extremes:
    quantisers of 1 and of 255), every flag combination, niter 0..3, sparse and
    dense coefficient planes -- GPU vs oracle, bit-exact"""
    rng = np.random.default_rng(20260925)
    layouts = [(1, 1), (2, 2), (2, 1), (1, 2), (4, 1)]
    for trial in range(48):
        w, h = int(rng.integers(8, 180)), int(rng.integers(8, 140))
        qual = int(rng.choice([1, 5, 20, 50, 80, 95, 100]))
        flags = int(rng.integers(0, 64))
        niter = int(rng.integers(0, 4))
        if trial % 3 == 0:
            coef, quant = synth.synth_gray(w, h, qual, seed=trial)
            if trial % 6 == 0:  # sparse plane: mostly zero blocks
                coef = (coef * (rng.random(coef.shape[:2]) < 0.3)[:, :, None]).astype(np.int16)
            a = gpu.do_quantsmooth([coef], [quant], flags, niter)
            b = oracle.do_quantsmooth([coef], [quant], flags, niter)
            assert_same_result(a, b, f"trial {trial}: gray {w}x{h} q{qual} flags={flags} niter={niter}")
        else:
            hs, vs = layouts[int(rng.integers(0, len(layouts)))]
            j = synth.synth_ycc(w, h, hs, vs, quality=qual, seed=trial)
            if trial % 2 and qual >= 20:  # coefficients beyond +-1023 until the final clamp
                j = inject_extreme_blocks(j, seed=trial)
            kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(w, h))
            a = gpu.do_quantsmooth(j["coefs"], j["quants"], flags, niter, **kw)
            b = oracle.do_quantsmooth(j["coefs"], j["quants"], flags, niter, **kw)
            assert_same_result(a, b, f"trial {trial}: ycc {w}x{h} {hs}x{vs} q{qual} flags={flags} niter={niter}")


@pytest.mark.parametrize("samp", [(2, 2), (1, 1)])
def test_gpu_colour_1080p_every_quality(gpu, oracle, synth, samp):
    """a full-HD YCbCr frame through every --quality level (the size at which the clamp-order
    bug surfaced: enough blocks for rare content to occur), default niter"""
    w, h = 1920, 1080
    j = synth.synth_ycc(w, h, samp[0], samp[1], quality=50)
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(w, h))
    for flags in (9, 11, 15, 0, 1, 3, 7):              # --quality 0..6 (reference quantsmooth.c:380-393)
        a = gpu.do_quantsmooth(j["coefs"], j["quants"], flags, 3, **kw)
        b = oracle.do_quantsmooth(j["coefs"], j["quants"], flags, 3, **kw)
        assert_same_result(a, b, f"1080p {samp} flags={flags}")


def test_gpu_plane_set_launch(gpu, oracle, synth):
    """qs_hip_idct_planes / qs_hip_smooth_planes: planes of different sizes, quant tables
    and luma/chroma roles in ONE launch per pass == each plane on its own"""
    import torch
    dev = torch.device("cuda:0")
    specs = [(256, 192, 50, 1), (40, 24, 20, 0), (8, 8, 90, 1), (520, 72, 35, 0), (64, 200, 75, 1), (1000, 16, 50, 0)]
    for flags in (0, 1, 32, 1 | 16):
        planes, keep, want = [], [], []
        for k, (w, h, qual, luma) in enumerate(specs):
            coef, quant = synth.synth_gray(w, h, qual, seed=k)
            hb, wb = coef.shape[:2]
            d_coef = torch.from_numpy(coef.copy()).to(dev)
            d_cst = torch.from_numpy(gpu.consts_build(quant, flags)).to(dev)
            d_plane = torch.empty(gpu.plane_bytes(wb, hb), dtype=torch.uint8, device=dev)
            d_status = torch.zeros(1, dtype=torch.int32, device=dev)
            keep.append((d_coef, d_cst, d_plane, d_status))
            planes.append((d_cst.data_ptr(), d_coef.data_ptr(), d_plane.data_ptr(), d_status.data_ptr(), wb, hb, luma))
            # a chroma plane of a YCbCr job: NO_REBALANCE_UV applies; luma: it does not
            f1 = flags if not luma else flags & ~32
            f1 = f1 | 16 if (not luma and flags & 32) else f1
            want.append(oracle.do_quantsmooth([coef], [quant], f1, 2)["coefs"][0])
        refs = gpu.plane_refs(planes)
        s = torch.cuda.current_stream().cuda_stream
        for it in range(2):
            gpu.idct_planes(refs, it == 0, s)
            gpu.smooth_planes(refs, flags, it == 1, s)
        torch.cuda.synchronize()
        for k, (d_coef, _, _, d_status) in enumerate(keep):
            assert int(d_status.item()) == 0
            assert np.array_equal(d_coef.cpu().numpy(), want[k]), f"flags={flags} plane {k}"
    with pytest.raises(Exception):
        gpu.smooth_planes(refs, 2, 0, s)                      # coupled flags are not a plane-set matter


def _batch_jobs(synth):
    jobs = []
    for k, (w, h, qual) in enumerate([(64, 64, 50), (200, 120, 25), (24, 88, 92), (333, 200, 60)]):
        coef, quant = synth.synth_gray(w, h, qual, seed=k)
        jobs.append(dict(coefs=[coef], quants=[quant]))
    for k, (w, h, hs, vs) in enumerate([(141, 93, 2, 2), (72, 40, 1, 1), (96, 64, 2, 1), (321, 240, 2, 2)]):
        j = synth.synth_ycc(w, h, hs, vs, quality=40 + 10 * k, seed=k)
        jobs.append(dict(coefs=j["coefs"], quants=j["quants"], hsamp=j["hsamp"], vsamp=j["vsamp"],
                         colorspace=3, image_size=(w, h)))
    # special cases inside a batch: range-check failure (careful re-run), all-ones table
    # (iterations skipped), a table entry >= 0x800 (stop), a zero quantiser
    coef, quant = synth.synth_gray(64, 64, 50)
    bad = coef.copy(); bad[3, 4, 0] = 300
    jobs.append(dict(coefs=[bad], quants=[quant]))
    jobs.append(dict(coefs=[coef], quants=[np.ones(64, np.uint16)]))
    qb = quant.copy(); qb[63] = 0x800
    jobs.append(dict(coefs=[coef], quants=[qb]))
    qz = quant.copy(); qz[5] = 0
    jobs.append(dict(coefs=[coef], quants=[qz]))
    return jobs


@pytest.mark.parametrize("flags,niter", [(0, 3), (1, 2), (16, 1), (1 | 32, 2), (7, 2), (9, 1), (0, 0), (4, 0)])
def test_gpu_batch_equals_single_jobs(gpu, oracle, synth, flags, niter):
    """qs_hip_do_quantsmooth_batch: independent jobs fused into plane-set launches, coupled /
    special ones through the general path -- every result as the oracle's for that job alone"""
    jobs = _batch_jobs(synth)
    got = gpu.do_quantsmooth_batch(jobs, flags, niter)
    assert len(got) == len(jobs)
    for k, (j, a) in enumerate(zip(jobs, got)):
        kw = {n: j[n] for n in ("hsamp", "vsamp", "colorspace", "image_size") if n in j}
        b = oracle.do_quantsmooth(j["coefs"], j["quants"], flags, niter, **kw)
        assert_same_result(a, b, f"batch job {k} flags={flags} niter={niter}")


def test_gpu_batch_many_planes_and_empty(gpu, oracle, synth):
    """more planes than one launch takes (QS_HIP_MAX_PLANES): several groups; empty batch"""
    assert gpu.do_quantsmooth_batch([], 0, 3) == []
    jobs = []
    for k in range(40):
        j = synth.synth_ycc(48 + 8 * (k % 5), 32 + 8 * (k % 3), 2, 2, quality=50, seed=k)
        jobs.append(dict(coefs=j["coefs"], quants=j["quants"], hsamp=j["hsamp"], vsamp=j["vsamp"],
                         colorspace=3, image_size=(48 + 8 * (k % 5), 32 + 8 * (k % 3))))
    got = gpu.do_quantsmooth_batch(jobs, 1, 2)
    for k, (j, a) in enumerate(zip(jobs, got)):
        b = oracle.do_quantsmooth(j["coefs"], j["quants"], 1, 2, hsamp=j["hsamp"], vsamp=j["vsamp"],
                                  colorspace=3, image_size=j["image_size"])
        assert_same_result(a, b, f"job {k}")


@pytest.mark.gpu
@pytest.mark.parametrize("flags,niter,couple_blocks", [(7, 3, None), (3, 2, None), (6, 1, "2000"), (7, 2, "1"), (2 | 32, 2, None)])
def test_gpu_batch_coupled_groups(gpu, oracle, synth, flags, niter, couple_blocks, monkeypatch):
    """coupled YCbCr jobs (JOINT_YUV / UPSAMPLE_UV, CLI --quality 5/6) of a batch advance in groups
    (run_coupled: luma planes as one plane set, then chroma planes as one): every result as the oracle's
    for that job alone -- more jobs than one group takes, subsampled and 4:4:4 jobs in one group, a job
    that trips the range check inside a group (careful re-run, the others unaffected), group sizes forced
    down to several groups / one job per task"""
    if couple_blocks:
        monkeypatch.setenv("QS_HIP_COUPLE_BLOCKS", couple_blocks)
    jobs = []
    layouts = [(2, 2), (1, 1), (2, 1), (1, 2), (2, 2)]
    for k in range(34):
        w, h = 40 + 8 * (k % 7) + (k % 3), 24 + 16 * (k % 4) + (k % 5)
        hs, vs = layouts[k % len(layouts)]
        j = synth.synth_ycc(w, h, hs, vs, quality=35 + (k % 4) * 15, seed=100 + k)
        jobs.append(dict(coefs=j["coefs"], quants=j["quants"], hsamp=j["hsamp"], vsamp=j["vsamp"],
                         colorspace=3, image_size=(w, h)))
    bad = [c.copy() for c in jobs[5]["coefs"]]
    bad[1][0, 0, 1] = 1500                                          # chroma coefficient far out of range
    jobs[5] = dict(jobs[5], coefs=bad)
    got = gpu.do_quantsmooth_batch(jobs, flags, niter)
    for k, (j, a) in enumerate(zip(jobs, got)):
        b = oracle.do_quantsmooth(j["coefs"], j["quants"], flags, niter, hsamp=j["hsamp"], vsamp=j["vsamp"],
                                  colorspace=3, image_size=j["image_size"])
        assert_same_result(a, b, f"coupled batch job {k} flags={flags} niter={niter}")


def test_gpu_fuzz_corpus():
    """tests/golden/fuzz_s2.jsonl: 400 seeded trials (959 jobs: every flag combination, sizes up to
    1400x1050, all chroma layouts, extreme blocks, batches) whose expected output hashes were written
    by the oracle (`tools/fuzz_gpu.py gen`); the GPU side regenerates the inputs and compares"""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "fuzz_gpu.py"), "run", str(root / "tests" / "golden" / "fuzz_s2.jsonl")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "400 trials, 959 jobs" in r.stdout and " 0 failures" in r.stdout


def _run_py(code, env_extra, timeout=900, with_stderr=False):
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, cwd=str(root), env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout + r.stderr if with_stderr else r.stdout


_BAND_ENV = {"QS_HIP_SPLIT_BLOCKS": "60", "QS_HIP_BAND_BLOCKS": "40"}


def test_gpu_banded_planes_fuzz_corpus():
    """the job layer cuts very large planes into bands with niter halo rows (upload, kernels and
    download of the bands overlap); with the two size thresholds lowered through the environment
    the same code runs on every plane of the committed corpus -- results must not change"""
    out = _run_py("import runpy, sys; sys.argv = ['fuzz_gpu.py', 'run', 'tests/golden/fuzz_s2.jsonl']; "
                  "runpy.run_path('tools/fuzz_gpu.py', run_name='__main__')", _BAND_ENV)
    assert "400 trials, 959 jobs" in out and " 0 failures" in out


def test_gpu_banded_plane_range_check_trips_late():
    """a bad coefficient in the LAST band of a banded job: earlier bands have already been written
    back when it is found; they must be restored and the job re-run in the careful order, giving the
    reference's stop semantics -- also inside a batch next to healthy jobs"""
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
import jpegqs_pkg
from oracle.oracle import Oracle
from helpers import assert_same_result
pkg = jpegqs_pkg.load(); hip = pkg.HipQS(); O = Oracle()
coef, quant = pkg.synth.synth_gray(64, 512, 50, seed=3)          # 8 x 64 blocks -> 13 bands of <= 5 rows + halo
bad = coef.copy(); bad[60, 2, 0] = 300                           # 300 * 16 > 0x7ff, in the last band
good, q2 = pkg.synth.synth_gray(96, 200, 60, seed=4)
for flags, niter in ((0, 3), (1, 2)):
    a = hip.do_quantsmooth([bad], [quant], flags, niter)
    b = O.do_quantsmooth([bad], [quant], flags, niter)
    assert b["ret"] == 1
    assert_same_result(a, b, f"single flags={flags}")
    jobs = [dict(coefs=[good], quants=[q2]), dict(coefs=[bad], quants=[quant]), dict(coefs=[coef], quants=[quant])]
    got = hip.do_quantsmooth_batch(jobs, flags, niter)
    for j, g in zip(jobs, got):
        assert_same_result(g, O.do_quantsmooth(j["coefs"], j["quants"], flags, niter), f"batch flags={flags}")
print("ok")
'''
    assert "ok" in _run_py(code, _BAND_ENV)


@pytest.mark.gpu
def test_gpu_vs_other_reference_orderings_information(gpu, oracle, synth):
    """Information, not a gate: the GPU matches oracle A = the reference's scalar build
    exactly; the reference's SIMD builds sum in other orders (and fuse multiply-adds), so a few
    blocks differ from the scalar path there.  Counted at 1080p q3/q4 and printed (pytest -s / the
    recorded output in profiles/); asserted only: GPU == scalar reference."""
    from oracle import oracle as om
    j = synth.synth_ycc(1920, 1080, 2, 2, quality=50, seed=5)
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(1920, 1080))
    total = sum(int(c.shape[0] * c.shape[1]) for c in j["coefs"])
    for quality in (3, 4):
        flags = 1 if quality == 4 else 0
        got = gpu.do_quantsmooth(j["coefs"], j["quants"], flags, 3, **kw)
        counts = {}
        for v in ("none", "sse2", "avx2", "avx512"):
            if not om.have_ref(v) or (v == "avx2" and not om.cpu_has("avx2")) or (v == "avx512" and not om.cpu_has("avx512bw")):
                continue
            ref = om.Reference(v).do_quantsmooth(j["coefs"], j["quants"], flags, 3, threads=0, **kw)
            counts[v] = sum(int((a != b).any(axis=2).sum()) for a, b in zip(got["coefs"], ref["coefs"]))
        print(f"[info] 1920x1080 4:2:0 q{quality} niter 3: blocks (of {total}) where the GPU result differs from the "
              f"reference build: {counts}")
        if "none" in counts:
            assert counts["none"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"QS_HIP_DP": "0"}, {"QS_HIP_DP_GROUPS": "100000"},
                                 {"QS_HIP_DP_GROUPS": "0", "QS_HIP_DP_GROUPS2": "100000"}],
                         ids=["one-block-per-lane", "diagonal-parallel-4-waves", "diagonal-parallel-2-waves"])
def test_gpu_fuzz_corpus_every_pass_b_form(env):
    """Pass B exists in two forms -- one block per lane (large planes) and the diagonal-parallel
    kernel with 4 or 2 waves per 64 blocks (small planes); the launcher picks by size, so the
    default corpus run mostly sees the small-plane form.  Here each form is forced on EVERY plane of
    the committed corpus (all flags, layouts, sizes up to 1400x1050, batches): 0 mismatches each."""
    out = _run_py("import runpy, sys; sys.argv = ['fuzz_gpu.py', 'run', 'tests/golden/fuzz_s2.jsonl']; "
                  "runpy.run_path('tools/fuzz_gpu.py', run_name='__main__')", env)
    assert "400 trials, 959 jobs" in out and " 0 failures" in out


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"QS_HIP_UPLOAD_STAGE": "0"}, {"QS_HIP_UPLOAD_STAGE": "1"}],
                         ids=["large-uploads-never-staged", "uploads-always-staged"])
def test_gpu_transfer_modes_full_size(env):
    """the two upload policies behind QS_HIP_UPLOAD_STAGE (default: staged only when a pinned block is pooled, i.e. the
    first call of a process uploads straight from caller memory and lands every result before writing any): an
    8192 x 6144 plane through the banded fused route, three times in one process, and over three logical devices -- every
    block against the oracle"""
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
import jpegqs_pkg
from oracle.oracle import Oracle
pkg = jpegqs_pkg.load(); hip = pkg.HipQS(); O = Oracle()
coef, quant = pkg.synth.synth_gray(8192, 6144, 50, seed=9)            # 786,432 blocks: three pipelined bands of 32 MiB
want = O.do_quantsmooth([coef], [quant], 1, 2, threads=16)["coefs"][0]
for rep in range(3):
    got = hip.do_quantsmooth([coef], [quant], 1, 2)
    assert got["ret"] == 0 and np.array_equal(got["coefs"][0], want), rep
got = hip.do_quantsmooth([coef], [quant], 1, 2, devices=[0, 0, 0])
assert np.array_equal(got["coefs"][0], want)
print("ok")
'''
    assert "ok" in _run_py(code, env)


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(65500, 8), (8, 65500), (65500, 24), (8, 8), (16, 16), (24, 8)])
def test_gpu_extreme_geometry(gpu, oracle, synth, size):
    """JPEG's limits: the widest (8,188 blocks in one block row) and the tallest (8,188 block rows of
    one block) image, and the smallest ones -- gray and 4:2:0 (chroma planes of a single block),
    every route: one call, sharded over three logical devices, --quality 3 / 4 / 6"""
    w, h = size
    coef, quant = synth.synth_gray(w, h, 50, seed=w + h)
    for flags in (0, 1):
        want = oracle.do_quantsmooth([coef], [quant], flags, 2, threads=0)
        assert_same_result(gpu.do_quantsmooth([coef], [quant], flags, 2), want, f"gray {size} flags={flags}")
        assert_same_result(gpu.do_quantsmooth([coef], [quant], flags, 2, devices=[0, 0, 0]), want, f"gray {size} flags={flags} sharded")
    if w <= 4096 or h <= 24:
        j = synth.synth_ycc(w, h, 2, 2, quality=50, seed=3)
        kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(w, h))
        for flags in (0, 7):
            want = oracle.do_quantsmooth(j["coefs"], j["quants"], flags, 2, threads=0, **kw)
            assert_same_result(gpu.do_quantsmooth(j["coefs"], j["quants"], flags, 2, **kw), want, f"ycc {size} flags={flags}")
            assert_same_result(gpu.do_quantsmooth(j["coefs"], j["quants"], flags, 2, devices=[0, 0], **kw), want,
                               f"ycc {size} flags={flags} sharded")
