"""One job spread over several (logical) GPUs inside the C ABI (csrc/qs_shard.cpp):
qs_hip_do_quantsmooth_sharded / qs_hip_set_devices.  On the one-GPU test box every band
lives on device 0 (`devices=[0, 0, ...]`): the band split, the halo pulls, their event
ordering and the band-local colour stages are the code that runs on a multi-GPU node, only
the copies are device-local.  Everything is compared bit for bit with the unsharded oracle."""
import numpy as np
import pytest

from helpers import assert_same_result, inject_extreme_blocks, load_golden


def test_shard_abi_argument_checks(hip):
    """no GPU needed: the entry points exist and validate their arguments"""
    import ctypes as C
    assert hip.lib.qs_hip_set_devices(None, 3) == -2            # QS_HIP_EINVAL
    assert hip.lib.qs_hip_set_devices(None, 0) == 0             # back to the default list
    bogus = (C.c_int * 2)(0, 99)
    assert hip.lib.qs_hip_set_devices(bogus, 2) == -2           # no such device
    assert b"no HIP device" in hip.lib.qs_hip_last_error()
    job, _ = hip._make_job([np.zeros((2, 2, 64), np.int16)], [np.full(64, 4, np.uint16)])
    assert hip.lib.qs_hip_do_quantsmooth_sharded(C.byref(job), 0, 3, None, 0) == -2


@pytest.mark.gpu
@pytest.mark.parametrize("nbands", [2, 3, 5, 8])
def test_sharded_gray_equals_unsharded(gpu, oracle, synth, nbands):
    coef, quant = synth.synth_gray(264, 520, 50, seed=4)        # 65 x 33 blocks
    for flags in (0, 1, 16):
        want = oracle.do_quantsmooth([coef], [quant], flags, 3)
        got = gpu.do_quantsmooth([coef], [quant], flags, 3, devices=[0] * nbands)
        assert_same_result(got, want, f"gray {nbands} bands flags={flags}")


def test_band_entry_argument_checks(hip):
    """no GPU needed: qs_hip_do_quantsmooth_band validates rank / communicator; librccl is NOT a link-time dependency"""
    import ctypes as C
    import subprocess
    from jpeg_quantsmooth_amd.hipqs import lib_path
    job, _ = hip._make_job([np.zeros((8, 2, 64), np.int16)], [np.full(64, 4, np.uint16)])
    f = hip.lib.qs_hip_do_quantsmooth_band
    assert f(C.byref(job), 0, 3, 2, 2, None) == -2               # rank out of range
    assert f(C.byref(job), 0, 3, 0, 2, None) == -2               # two ranks need a communicator
    assert f(C.byref(job), 0, 0, 0, 1, None) == 0                # niter 0: the reference's early-out, before any device
    needed = subprocess.run(["readelf", "-d", str(lib_path())], capture_output=True, text=True).stdout
    assert "librccl" not in needed and "libamdhip64" in needed


def test_shard_schedule_setter(hip):
    """no GPU needed: qs_hip_set_shard_schedule validates its argument"""
    assert hip.lib.qs_hip_set_shard_schedule(2) == -2 and hip.lib.qs_hip_set_shard_schedule(-2) == -2
    for v in (1, 0, -1):
        assert hip.lib.qs_hip_set_shard_schedule(v) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("nbands", [2, 3, 5, 8])
def test_sharded_deep_halo_schedule_equals_unsharded(gpu, oracle, synth, nbands):
    """the COMMUNICATION-AVOIDING schedule (qs_hip_set_shard_schedule(1)): every cut side of a band carries niter block
    rows of its neighbour, no halo exchange at all -- the same coefficients as the unsharded run, for gray and for
    independent YCbCr components (each cut on its own rows), with the row-pointer entry point's in-place write-back
    limited to the rows a band owns; niter too large for the bands falls back to the exchange schedule"""
    coef, quant = synth.synth_gray(264, 1040, 50, seed=4)       # 130 x 33 blocks
    j = synth.synth_ycc(333, 777, 2, 2, quality=45, seed=9)
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(333, 777))
    gpu.set_shard_schedule(1)
    try:
        for flags, niter in ((0, 3), (1, 2), (16, 1), (0, 5), (1, 40)):
            want = oracle.do_quantsmooth([coef], [quant], flags, min(niter, 6))
            got = gpu.do_quantsmooth([coef], [quant], flags, min(niter, 6), devices=[0] * nbands)
            assert_same_result(got, want, f"gray deep halo {nbands} bands flags={flags} niter={niter}")
        for flags, niter in ((0, 3), (1, 2), (32, 2)):
            want = oracle.do_quantsmooth(j["coefs"], j["quants"], flags, niter, **kw)
            got = gpu.do_quantsmooth(j["coefs"], j["quants"], flags, niter, devices=[0] * nbands, **kw)
            assert_same_result(got, want, f"ycc deep halo {nbands} bands flags={flags}")
        # coupled flags keep their own schedule (one-off exchanges), whatever is selected here
        want = oracle.do_quantsmooth(j["coefs"], j["quants"], 7, 2, **kw)
        assert_same_result(gpu.do_quantsmooth(j["coefs"], j["quants"], 7, 2, devices=[0] * min(nbands, 4), **kw), want, "q6 with deep selected")
        # a tripped range check in a halo copy of a block is still the job's range check
        job, wantb = load_golden("gray64_badcoef_q3_n2")
        assert_same_result(gpu.do_quantsmooth(job["coefs"], job["quants"], job["flags"], job["niter"], devices=[0, 0], **job["kw"]), wantb, "bad coefficient")
    finally:
        gpu.set_shard_schedule(-1)


@pytest.mark.gpu
def test_sharded_deep_halo_transparent_route_and_trace(gpu):
    """QS_HIP_SHARD_SCHEDULE=deep in the environment: the fuzz corpus through qs_hip_do_quantsmooth with three logical
    devices, every shardable job on the deep-halo schedule (the trace line says which one ran)"""
    from test_gpu_parity import _run_py
    out = _run_py("import runpy, sys; sys.argv = ['fuzz_gpu.py', 'run', 'tests/golden/fuzz_s2.jsonl']; "
                  "runpy.run_path('tools/fuzz_gpu.py', run_name='__main__')",
                  {"QS_HIP_DEVICES": "0,0,0", "QS_HIP_SHARD_MIN_BLOCKS": "1", "QS_HIP_TRACE": "1", "QS_HIP_SHARD_SCHEDULE": "deep"},
                  with_stderr=True)
    assert "400 trials, 959 jobs" in out and " 0 failures" in out
    assert "schedule: deep halo, no exchange" in out


@pytest.mark.gpu
@pytest.mark.parametrize("size,samp", [((256, 160), (2, 2)), ((333, 517), (2, 2)), ((321, 200), (1, 1)),
                                       ((208, 328), (2, 1)), ((200, 264), (1, 2)), ((320, 264), (4, 1))])
def test_sharded_colour_independent_components(gpu, oracle, synth, size, samp):
    """--quality 3/4 on YCbCr: one plane set per band, every component cut on its own rows"""
    w, h = size
    j = synth.synth_ycc(w, h, samp[0], samp[1], quality=45, seed=9)
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(w, h))
    for flags, niter in ((0, 3), (1, 2), (32, 2)):
        want = oracle.do_quantsmooth(j["coefs"], j["quants"], flags, niter, **kw)
        for nb in (2, 4):
            got = gpu.do_quantsmooth(j["coefs"], j["quants"], flags, niter, devices=[0] * nb, **kw)
            assert_same_result(got, want, f"{size} {samp} flags={flags} bands={nb}")


@pytest.mark.gpu
@pytest.mark.parametrize("size,samp,nbands", [((256, 320), (2, 2), 2), ((333, 517), (2, 2), 3), ((321, 400), (2, 2), 4),
                                               ((160, 200), (1, 1), 2), ((208, 264), (2, 1), 2), ((200, 520), (1, 2), 3)])
def test_sharded_colour_coupled_flags(gpu, oracle, synth, size, samp, nbands):
    """--quality 5/6 (JOINT_YUV / UPSAMPLE_UV, also with LOW_QUALITY): bands cut on chroma block
    rows, halos of luma, low-res luma and chroma, band-local downsample / upsample / re-FDCT"""
    w, h = size
    # extreme blocks: coefficients beyond +-1023 before the final clamp (the refresh passes must see them unclamped)
    j = inject_extreme_blocks(synth.synth_ycc(w, h, samp[0], samp[1], quality=40, seed=12))
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(w, h))
    for flags, niter in ((7, 2), (3, 2), (7, 1), (7 | 32, 1), (15, 2), (5, 1), (11, 1), (6 | 16, 2)):
        want = oracle.do_quantsmooth(j["coefs"], j["quants"], flags, niter, **kw)
        got = gpu.do_quantsmooth(j["coefs"], j["quants"], flags, niter, devices=[0] * nbands, **kw)
        assert_same_result(got, want, f"{size} {samp} flags={flags} niter={niter} bands={nbands}")


@pytest.mark.gpu
def test_sharded_range_check_and_unsupported(gpu, oracle, synth):
    """a coefficient out of range in one band: the whole job is re-run in the careful order
    (reference stop semantics); flag combinations without a sharded route say so"""
    job, want = load_golden("gray64_badcoef_q3_n2")
    got = gpu.do_quantsmooth(job["coefs"], job["quants"], job["flags"], job["niter"], devices=[0, 0], **job["kw"])
    assert_same_result(got, want, "bad coefficient, 2 bands")
    coef, quant = synth.synth_gray(64, 64, 50)
    with pytest.raises(Exception) as e:
        gpu.do_quantsmooth([coef], [quant], 8, 2, devices=[0, 0])      # LOW_QUALITY gray: one device only
    assert getattr(e.value, "code", None) == -4


@pytest.mark.gpu
def test_sharded_route_taken_transparently(gpu):
    """qs_hip_do_quantsmooth itself shards once several devices are configured and the job is
    large enough: the committed fuzz corpus with QS_HIP_DEVICES=0,0,0 and the size threshold at 1
    (every job that has a sharded route takes it, batches included)"""
    from test_gpu_parity import _run_py
    out = _run_py("import runpy, sys; sys.argv = ['fuzz_gpu.py', 'run', 'tests/golden/fuzz_s2.jsonl']; "
                  "runpy.run_path('tools/fuzz_gpu.py', run_name='__main__')",
                  {"QS_HIP_DEVICES": "0,0,0", "QS_HIP_SHARD_MIN_BLOCKS": "1", "QS_HIP_TRACE": "1"})
    assert "400 trials, 959 jobs" in out and " 0 failures" in out


def test_rows_entry_point_argument_checks(hip):
    import ctypes as C
    job, _ = hip._make_job([np.zeros((2, 2, 64), np.int16)], [np.full(64, 4, np.uint16)])
    assert hip.lib.qs_hip_do_quantsmooth_rows(C.byref(job), None, 0, 3, 0, C.cast(None, hip.lib.qs_hip_do_quantsmooth_rows.argtypes[5]), None) == -2


@pytest.mark.gpu
@pytest.mark.parametrize("padded", [False, True])
def test_job_given_as_rows_equals_contiguous(gpu, oracle, synth, padded):
    """qs_hip_do_quantsmooth_rows: the blocks handed over as one pointer per block row (libjpeg's
    JBLOCKROWs; `padded` = allocated rows wider than width_in_blocks, i.e. rows not adjacent) are
    processed in place, for the fused, the general (coupled flags) and the sharded route"""
    import ctypes as C
    from jpeg_quantsmooth_amd.hipqs import PROGRESS_FN
    j = synth.synth_ycc(200, 136, 2, 2, quality=45, seed=3)
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(200, 136))
    for flags, devices in ((0, None), (1, None), (7, None), (11, None), (0, [0, 0]), (7, [0, 0])):
        want = oracle.do_quantsmooth(j["coefs"], j["quants"], flags, 2, **kw)
        job, work = gpu._make_job(j["coefs"], j["quants"], j["hsamp"], j["vsamp"], 3, (200, 136))
        store, tables = [], []
        for ci, a in enumerate(work):
            hb, wb = a.shape[:2]
            pad = 3 if padded else 0
            buf = np.zeros((hb, wb + pad, 64), np.int16)
            buf[:, :wb] = a
            store.append(buf)
            tbl = (C.c_void_p * hb)(*[buf[y].ctypes.data for y in range(hb)])
            tables.append(tbl)
            job.coef[ci] = None
        rows = (C.POINTER(C.c_void_p) * 4)(*[C.cast(t, C.POINTER(C.c_void_p)) for t in tables])
        import os
        if devices:                                              # transparent sharding: device list + size threshold
            gpu.set_devices(devices)
            os.environ["QS_HIP_SHARD_MIN_BLOCKS"] = "1"
        try:
            ret = gpu.lib.qs_hip_do_quantsmooth_rows(C.byref(job), rows, flags, 2, 0, C.cast(None, PROGRESS_FN), None)
        finally:
            gpu.set_devices([])
            os.environ.pop("QS_HIP_SHARD_MIN_BLOCKS", None)
        assert ret == want["ret"] == 0
        for ci in range(3):
            got = store[ci][:, :work[ci].shape[1]]
            if want["up"] and ci:
                cnt = job.up_wblk * job.up_hblk * 64
                up = np.frombuffer((C.c_int16 * cnt).from_address(job.coef_up[ci - 1]), dtype=np.int16).reshape(job.up_hblk, job.up_wblk, 64)
                assert np.array_equal(up, want["coefs"][ci]), (flags, ci)
                gpu.lib.qs_hip_free(job.coef_up[ci - 1])
            else:
                assert np.array_equal(got, want["coefs"][ci]), (flags, ci, padded)
            if padded:
                assert not store[ci][:, work[ci].shape[1]:].any()      # nothing written beyond width_in_blocks


@pytest.mark.gpu
@pytest.mark.parametrize("flags,niter", [(0, 3), (1, 2), (7, 2), (11, 1)])
def test_batch_spread_over_devices(gpu, oracle, synth, flags, niter):
    """qs_hip_do_quantsmooth_batch with several devices configured: whole jobs go to different devices
    (independent objects, no exchange) -- plane-set jobs as one group per device, coupled jobs from
    worker threads per device; three logical devices on the one GPU here"""
    jobs = []
    for n, (w, h, samp) in enumerate([(200, 136, (2, 2)), (96, 64, (1, 1)), (333, 217, (2, 2)), (64, 64, (2, 1)),
                                      (160, 120, (2, 2)), (72, 40, (1, 1)), (256, 160, (2, 2))]):
        j = synth.synth_ycc(w, h, samp[0], samp[1], quality=40 + 5 * n, seed=30 + n)
        jobs.append(dict(coefs=j["coefs"], quants=j["quants"], hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(w, h)))
    coef, quant = synth.synth_gray(120, 88, 50, seed=5)
    jobs.append(dict(coefs=[coef], quants=[quant]))
    gpu.set_devices([0, 0, 0])
    try:
        got = gpu.do_quantsmooth_batch(jobs, flags, niter)
    finally:
        gpu.set_devices([])
    for n, (j, g) in enumerate(zip(jobs, got)):
        kw = {k: j[k] for k in ("hsamp", "vsamp", "colorspace", "image_size") if k in j}
        assert_same_result(g, oracle.do_quantsmooth(j["coefs"], j["quants"], flags, niter, **kw), f"job {n} flags={flags}")
