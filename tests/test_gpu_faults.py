"""Error paths of the job layer under injected allocation failures (QS_HIP_TEST_FAIL_ALLOC=N: the N-th device
buffer request fails once; QS_HIP_TEST_FAIL_PINNED=N: the N-th pinned host buffer request, csrc/qs_xfer.h).  Whatever route a job takes -- plane sets, the
general per-component route, coupled groups from worker threads, bands over logical devices -- a failed
allocation must surface as an error code (or as the reference's own fall-back: a component whose pixel
plane cannot be allocated is dequantised only, reference quantsmooth.h:2551-2566), never as a crash, a
hang or a wrong result, and the library must work normally afterwards (streams drained before pooled
buffers are reused, compute slots released, worker threads joined)."""
import numpy as np
import pytest

from helpers import assert_same_result

pytestmark = pytest.mark.gpu


def _dequant_only(coef, quant):
    q = np.asarray(quant, dtype=np.int32).reshape(1, 1, 64)
    return np.clip(coef.astype(np.int32) * q, -1023, 1023).astype(np.int16)


def _colour_jobs(synth, n, seed0):
    jobs = []
    for k in range(n):
        w, h = 96 + 16 * (k % 3), 64 + 8 * (k % 4)
        j = synth.synth_ycc(w, h, 2, 2, quality=40 + 10 * (k % 3), seed=seed0 + k)
        jobs.append(dict(coefs=j["coefs"], quants=j["quants"], hsamp=j["hsamp"], vsamp=j["vsamp"],
                         colorspace=3, image_size=(w, h)))
    return jobs


def _want(oracle, j, flags, niter):
    kw = {n: j[n] for n in ("hsamp", "vsamp", "colorspace", "image_size") if n in j}
    # (threads: the box reports 256 logical CPUs on a 16-core quota; an OpenMP team of 256 is slower than one thread here)
    nblk = sum(int(c.shape[0] * c.shape[1]) for c in j["coefs"])
    return oracle.do_quantsmooth(j["coefs"], j["quants"], flags, niter, threads=8 if nblk > 30000 else 1, **kw)


@pytest.mark.parametrize("nth", [1, 2, 3, 4, 5, 7])
def test_single_job_allocation_failure(gpu, oracle, synth, monkeypatch, nth):
    """one gray job (plane-set route) and one 4:2:0 --quality 6 job (general route)"""
    coef, quant = synth.synth_gray(200, 136, 50, seed=3)
    gray = dict(coefs=[coef], quants=[quant])
    colour = _colour_jobs(synth, 1, 40)[0]
    for job, flags in ((gray, 0), (colour, 7)):
        want = _want(oracle, job, flags, 2)
        kw = {n: job[n] for n in ("hsamp", "vsamp", "colorspace", "image_size") if n in job}
        monkeypatch.setenv("QS_HIP_TEST_FAIL_ALLOC", f"{nth}")
        try:
            got = gpu.do_quantsmooth(job["coefs"], job["quants"], flags, 2, **kw)
        except Exception:
            got = None                                            # the failure was reported: fine
        finally:
            monkeypatch.delenv("QS_HIP_TEST_FAIL_ALLOC")
        if nth == 1 and flags == 0:
            assert got is None, "the very first allocation of the plane-set route failed: the call must say so"
        if got is not None and flags == 0:
            # no exception: either the fault was never reached, or the reference's fall-back applied
            assert got["ret"] == 0
            ok = np.array_equal(got["coefs"][0], want["coefs"][0]) or \
                np.array_equal(got["coefs"][0], _dequant_only(coef, quant))
            assert ok, f"nth={nth}: neither the recovered nor the dequantised-only plane"
        # and the library is healthy afterwards
        again = gpu.do_quantsmooth(job["coefs"], job["quants"], flags, 2, **kw)
        assert_same_result(again, want, f"after an injected failure (nth={nth}, flags={flags})")


@pytest.mark.parametrize("nth", [1, 2, 3, 5, 6, 9, 12])
@pytest.mark.parametrize("flags", [0, 7])
def test_batch_allocation_failure(gpu, oracle, synth, monkeypatch, nth, flags):
    """batches: plane-set groups (flags 0) and coupled groups on worker threads (flags 7)"""
    jobs = _colour_jobs(synth, 7, 60)
    monkeypatch.setenv("QS_HIP_COUPLE_BLOCKS", "1500")            # several coupled groups -> several worker threads
    monkeypatch.setenv("QS_HIP_TEST_FAIL_ALLOC", f"{nth}")
    try:
        got = gpu.do_quantsmooth_batch(jobs, flags, 2)
    except Exception:
        got = None
    finally:
        monkeypatch.delenv("QS_HIP_TEST_FAIL_ALLOC")
    if nth == 1:
        assert got is None or any(a["ret"] < 0 for a in got), "an allocation failed: somebody must report it"
    if got is not None:
        for k, (j, a) in enumerate(zip(jobs, got)):
            if a["ret"] < 0:
                continue                                          # this job reported the failure
            if flags == 0:                                        # (a coupled job may have taken the dequantise-only fall-back)
                assert_same_result(a, _want(oracle, j, flags, 2), f"nth={nth} job {k}")
    again = gpu.do_quantsmooth_batch(jobs, flags, 2)
    for k, (j, a) in enumerate(zip(jobs, again)):
        assert_same_result(a, _want(oracle, j, flags, 2), f"after an injected failure (nth={nth}) job {k}")


@pytest.mark.parametrize("nth", [1, 2, 4, 6, 9])
def test_sharded_allocation_failure(gpu, oracle, synth, monkeypatch, nth):
    """bands over two logical devices on one GPU (C-side sharding): set route and coupled-colour route"""
    coef, quant = synth.synth_gray(264, 328, 50, seed=4)
    gray = dict(coefs=[coef], quants=[quant])
    colour = _colour_jobs(synth, 1, 80)[0]
    for job, flags in ((gray, 1), (colour, 7)):
        kw = {n: job[n] for n in ("hsamp", "vsamp", "colorspace", "image_size") if n in job}
        want = _want(oracle, job, flags, 2)
        monkeypatch.setenv("QS_HIP_TEST_FAIL_ALLOC", f"{nth}")
        raised = False
        try:
            gpu.do_quantsmooth(job["coefs"], job["quants"], flags, 2, devices=[0, 0], **kw)
        except Exception:
            raised = True
        finally:
            monkeypatch.delenv("QS_HIP_TEST_FAIL_ALLOC")
        if nth == 1:
            assert raised, "the first allocation of a band failed: the call must say so"
        again = gpu.do_quantsmooth(job["coefs"], job["quants"], flags, 2, devices=[0, 0], **kw)
        assert_same_result(again, want, f"after an injected failure (nth={nth}, flags={flags})")


@pytest.mark.parametrize("nth", [1, 2, 3, 4, 6, 8])
@pytest.mark.parametrize("flags", [1, 7])
def test_pinned_allocation_failure(gpu, oracle, synth, monkeypatch, nth, flags):
    """pinned staging buffers: without one a transfer takes its pageable fall-back (the result is still
    exact) or the call reports the failure; planes large enough for the staged paths (> 1 MiB)"""
    big = synth.synth_ycc(1024, 768, 2, 2, quality=50, seed=7)
    job = dict(coefs=big["coefs"], quants=big["quants"], hsamp=big["hsamp"], vsamp=big["vsamp"],
               colorspace=3, image_size=(1024, 768))
    small = _colour_jobs(synth, 5, 90)
    kw = {n: job[n] for n in ("hsamp", "vsamp", "colorspace", "image_size")}
    want = _want(oracle, job, flags, 1)
    monkeypatch.setenv("QS_HIP_TEST_FAIL_PINNED", f"{nth}")
    try:
        got = gpu.do_quantsmooth(job["coefs"], job["quants"], flags, 1, **kw)
    except Exception:
        got = None
    finally:
        monkeypatch.delenv("QS_HIP_TEST_FAIL_PINNED")
    if got is not None:
        assert_same_result(got, want, f"single job with a failed pinned allocation (nth={nth}, flags={flags})")
    monkeypatch.setenv("QS_HIP_TEST_FAIL_PINNED", f"{nth}")
    try:
        res = gpu.do_quantsmooth_batch([job] + small, flags, 1)
    except Exception:
        res = None
    finally:
        monkeypatch.delenv("QS_HIP_TEST_FAIL_PINNED")
    if res is not None:
        for k, (j, a) in enumerate(zip([job] + small, res)):
            if a["ret"] >= 0:
                assert_same_result(a, _want(oracle, j, flags, 1), f"batch job {k} (nth={nth}, flags={flags})")
    again = gpu.do_quantsmooth(job["coefs"], job["quants"], flags, 1, **kw)
    assert_same_result(again, want, f"after an injected pinned failure (nth={nth}, flags={flags})")


def _raw_call(gpu, job, flags, niter, devices=None):
    """the C entry point on arrays we keep: -> (return code, the arrays the library worked on, the job struct)"""
    import ctypes as C
    from jpeg_quantsmooth_amd.hipqs import PROGRESS_FN
    kw = {n: job[n] for n in ("hsamp", "vsamp", "colorspace", "image_size") if n in job}
    j, work = gpu._make_job(job["coefs"], job["quants"], kw.get("hsamp"), kw.get("vsamp"), kw.get("colorspace"), kw.get("image_size"))
    if devices:
        arr = (C.c_int * len(devices))(*devices)
        rc = gpu.lib.qs_hip_do_quantsmooth_sharded(C.byref(j), flags, niter, arr, len(devices))
    else:
        rc = gpu.lib.qs_hip_do_quantsmooth(C.byref(j), flags, niter, 0, C.cast(None, PROGRESS_FN), None)
    return rc, work, j


@pytest.mark.parametrize("route", ["general", "sharded-set", "sharded-colour"])
@pytest.mark.parametrize("size", [(1024, 768), (2048, 1536)], ids=["bands-under-1MiB", "staged-bands"])
@pytest.mark.parametrize("nth", [1, 2])
def test_reported_failure_leaves_the_image_untouched(gpu, oracle, synth, monkeypatch, route, size, nth):
    """results are scattered to caller memory piece by piece (per component, per band); a
    failure AFTER some pieces have been written must put the original blocks back -- the reference's
    applications ignore do_quantsmooth's return value and would write a half-smoothed, dequantised image
    with unchanged quant tables.  QS_HIP_TEST_FAIL_FINISH=N: the N-th scatter reports a HIP error after
    writing its pieces (where a restore copy exists: the pinned upload staging of transfers >= 1 MiB), or, where
    none exists (small bands uploaded straight from caller memory), before anything is written: everything lands
    in library-owned memory first."""
    big = synth.synth_ycc(size[0], size[1], 2, 2, quality=50, seed=11)
    job = dict(coefs=big["coefs"], quants=big["quants"], hsamp=big["hsamp"], vsamp=big["vsamp"], colorspace=3, image_size=size)
    nb = 3 if size[0] == 1024 else 2
    flags, devices = {"general": (7, None), "sharded-set": (1, [0] * nb), "sharded-colour": (7, [0] * nb)}[route]
    monkeypatch.setenv("QS_HIP_TEST_FAIL_FINISH", f"{nth}")
    try:
        rc, work, j = _raw_call(gpu, job, flags, 2, devices)
    finally:
        monkeypatch.delenv("QS_HIP_TEST_FAIL_FINISH")
    assert rc < 0, "the injected transfer failure must be reported"
    for ci in range(3):
        assert np.array_equal(work[ci], job["coefs"][ci]), f"{route}: component {ci} was left modified after a reported failure"
        assert list(j.quant[ci][:]) == [int(v) for v in job["quants"][ci]], f"{route}: quant table {ci} changed"
    assert j.up_wblk == 0
    kw = {n: job[n] for n in ("hsamp", "vsamp", "colorspace", "image_size")}
    again = gpu.do_quantsmooth(job["coefs"], job["quants"], flags, 2, devices=devices, **kw)
    assert_same_result(again, _want(oracle, job, flags, 2), f"{route}: the call after the injected failure")


@pytest.mark.parametrize("width,band_blocks,kind", [(64, 128, "small bands: no restore copy, everything lands first"),
                                                     (2048, 8192, "1 MiB bands: written early, restored from the upload staging")])
def test_reported_failure_leaves_a_banded_image_untouched(width, band_blocks, kind):
    """the same for the fused route with the plane cut into pipelined bands (run_fused): with a pinned upload staging
    copy the bands are written back while later bands are still in flight and a failure restores them; without one
    (bands under 1 MiB) they are held back until every band has landed.  The failing scatter is the 2nd / 3rd / last."""
    from test_gpu_parity import _run_py
    code = r'''
import sys, ctypes as C, os, numpy as np
sys.path.insert(0, "tests")
import jpegqs_pkg
from oracle.oracle import Oracle
from helpers import assert_same_result
pkg = jpegqs_pkg.load(); hip = pkg.HipQS(); O = Oracle()
from jpeg_quantsmooth_amd.hipqs import PROGRESS_FN
width = int(sys.argv[1])
coef, quant = pkg.synth.synth_gray(width, 1024, 50, seed=3)       # 128 block rows -> 8 / 4 bands of 16 / 32 rows + halo
want = O.do_quantsmooth([coef], [quant], 1, 1, threads=8)
for nth in (2, 3, int(sys.argv[2])):
    j, work = hip._make_job([coef], [quant])
    os.environ["QS_HIP_TEST_FAIL_FINISH"] = str(nth)
    rc = hip.lib.qs_hip_do_quantsmooth(C.byref(j), 1, 1, 0, C.cast(None, PROGRESS_FN), None)
    del os.environ["QS_HIP_TEST_FAIL_FINISH"]
    assert rc < 0, (nth, rc)
    assert np.array_equal(work[0], coef), f"nth={nth}: rows left modified after a reported failure"
    assert list(j.quant[0][:]) == [int(v) for v in quant]
    assert_same_result(hip.do_quantsmooth([coef], [quant], 1, 1), want, f"after nth={nth}")
print("ok")
'''
    nbands = (width // 8) * 128 // band_blocks
    env = {"QS_HIP_SPLIT_BLOCKS": "60", "QS_HIP_BAND_BLOCKS": str(band_blocks), "QS_HIP_TEST_HOOKS": "1"}
    assert "ok" in _run_py("import sys; sys.argv = ['x', '%d', '%d']\n" % (width, nbands) + code, env), kind


@pytest.mark.parametrize("route", ["general", "fused", "sharded-set", "sharded-colour"])
def test_direct_download_path_failure_leaves_the_image_untouched(gpu, oracle, synth, monkeypatch, route):
    """an UNSTAGED download whose caller holds a restore copy goes straight into the caller's arrays
    (Download::finish, direct path: a few blocking copies instead of a full-size temporary) and can fail half-way.
    That combination arises when the pinned block of the DOWNLOAD cannot be had while the upload was staged.  Drive it
    on every route: the k-th pinned request fails (k = 1..10 walks through upload staging, status words and download
    staging of every component / band) and the 1st or 2nd landing reports a failure after writing.  Whatever happens,
    either the call fails and the caller's arrays and tables are the input's, or it succeeds with the exact result."""
    size = (2048, 1536)
    big = synth.synth_ycc(size[0], size[1], 2, 2, quality=50, seed=13)
    job = dict(coefs=big["coefs"], quants=big["quants"], hsamp=big["hsamp"], vsamp=big["vsamp"], colorspace=3, image_size=size)
    flags, devices = {"general": (7, None), "fused": (1, None), "sharded-set": (1, [0, 0]), "sharded-colour": (7, [0, 0])}[route]
    want = _want(oracle, job, flags, 2)
    failed = ok = 0
    for k in range(1, 11):
        for nth in (1, 2):
            monkeypatch.setenv("QS_HIP_TEST_FAIL_PINNED", str(k))
            monkeypatch.setenv("QS_HIP_TEST_FAIL_FINISH", str(nth))
            try:
                rc, work, j = _raw_call(gpu, job, flags, 2, devices)
            finally:
                monkeypatch.delenv("QS_HIP_TEST_FAIL_PINNED")
                monkeypatch.delenv("QS_HIP_TEST_FAIL_FINISH")
            if rc < 0:
                failed += 1
                for ci in range(3):
                    assert np.array_equal(work[ci], job["coefs"][ci]), f"{route} k={k} nth={nth}: component {ci} left modified after a reported failure"
                    assert list(j.quant[ci][:]) == [int(v) for v in job["quants"][ci]], f"{route} k={k} nth={nth}: quant table {ci} changed"
                assert j.up_wblk == 0
            else:
                ok += 1
                got = gpu._job_result(j, work, job["quants"], rc)
                assert_same_result(got, want, f"{route} k={k} nth={nth}: reported success")
    assert failed > 0, "the injected landing failure never fired"
    kw = {n: job[n] for n in ("hsamp", "vsamp", "colorspace", "image_size")}
    again = gpu.do_quantsmooth(job["coefs"], job["quants"], flags, 2, devices=devices, **kw)
    assert_same_result(again, want, f"{route}: the call after the injected failures")
