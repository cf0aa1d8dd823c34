"""Every BASELINE.json configuration at FULL size, every block compared with the CPU reference.

`truth` is the compiled, unmodified reference (oracle/_ref/libqsref_none.so: the scalar build,
OpenMP over block rows, reference quantsmooth.h:2587-2640) when it travelled with the tree, the
plain-C port otherwise (the two are pinned to each other by tests/test_oracle.py).  The summation
order of reference quantsmooth.h:1517-1549 decides single blocks at a rate of about 1e-5 per
block-iteration between orderings, i.e. tens of blocks per 8192^2 plane: sampling
a few rows would be the wrong economy, so these tests compare WHOLE planes -- a 16-core host does
an 8192^2 plane in a few seconds.

  configs[2] + the headline metric : 8192^2 luma, --quality 3 and 4, niter 3
      plane layer (single launches, and the plane-set launches bench.py times), job layer
      (qs_hip_do_quantsmooth: the banded fused route), 8 logical devices (qs_shard.cpp),
      8 logical bands of the torch.distributed driver (bands.py)
  configs[3] : 16384^2 luma, --quality 3 -- job layer, 8 logical devices, 8 logical bands
  configs[4] : 8192^2 4:2:0, --quality 6 (JOINT_YUV + UPSAMPLE_UV), niter 5 -- job layer,
      8 logical devices, 8 logical colour bands; reference quantsmooth.h:2691-2815
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def truth():
    from oracle import oracle as om
    if om.have_ref("none"):
        return om.Reference("none")
    return om.Oracle()


def _diff(got, want, what):
    """whole-array comparison with a useful message"""
    assert got.shape == want.shape, f"{what}: shape {got.shape} != {want.shape}"
    if np.array_equal(got, want):
        return
    bad = np.argwhere((got != want).any(axis=2))
    raise AssertionError(f"{what}: {len(bad)} of {got.shape[0] * got.shape[1]} blocks differ from the reference, "
                         f"first at (by, bx) = {tuple(bad[0])}")


def _luma_band_driver(gpu, torch, bands, coef_dev, quant, flags, niter, nbands):
    """bands.py with N logical bands on one device: the plane-set launches and the packed halo rows
    of bench.py --gpus N, the exchange done by device copies instead of RCCL"""
    hblk = int(coef_dev.shape[0])
    engines, topos = [], []
    for r in range(nbands):
        r0, r1 = bands.band_rows(hblk, nbands, r)
        topos.append(bands.BandTopology(r, nbands, r0, r1))
        engines.append(bands.HipBandEngine(gpu, torch, coef_dev[r0:r1].clone(), quant, flags))
    for it in range(niter):
        for e, t in zip(engines, topos):
            band = (1 if t.up is not None else 0) | (2 if t.down is not None else 0)
            refs = gpu.plane_refs([(e.cst.data_ptr(), e.coef.data_ptr(), e.plane.data_ptr(), e.status.data_ptr(),
                                    e.wblk, e.hblk, 1, band)])
            gpu.idct_planes(refs, it == 0, e._s())
        bands.exchange_halo_local(engines)
        for e, t in zip(engines, topos):
            band = (1 if t.up is not None else 0) | (2 if t.down is not None else 0)
            refs = gpu.plane_refs([(e.cst.data_ptr(), e.coef.data_ptr(), e.plane.data_ptr(), e.status.data_ptr(),
                                    e.wblk, e.hblk, 1, band)])
            gpu.smooth_planes(refs, flags, it == niter - 1, e._s())
    torch.cuda.synchronize()
    assert not any(e.bad_coef() for e in engines)
    return torch.cat([e.coef for e in engines], dim=0)


@pytest.mark.parametrize("quality", [3, 4])
def test_full_8192_luma_every_block_vs_reference(gpu, pkg, truth, big_plane, quality):
    """the metric's configuration (q=3) and configs[2] (q=4): 1,048,576 blocks, all compared"""
    import torch
    from jpeg_quantsmooth_amd import bands
    coef, quant = big_plane
    flags = pkg.flags_for_quality(quality)
    t0 = time.time()
    want = truth.do_quantsmooth([coef], [quant], flags, 3, threads=0)
    print(f"[info] reference ({type(truth).__name__}) 8192^2 q{quality}: {time.time() - t0:.1f} s")
    assert want["ret"] == 0
    want = want["coefs"][0]
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(coef).to(dev)

    # (i) plane layer, one launch per pass and plane
    eng = bands.HipBandEngine(gpu, torch, d_in.clone(), quant, flags, luma=1, device=dev)
    for it in range(3):
        eng.idct(it == 0, 1, 1)
        eng.smooth(it == 2)
    torch.cuda.synchronize()
    assert not eng.bad_coef()
    _diff(eng.coef.cpu().numpy(), want, f"q{quality} plane layer")

    # (ii) plane-set launches over three planes at once: what bench.py times
    engs = [bands.HipBandEngine(gpu, torch, d_in.clone(), quant, flags, luma=1, device=dev) for _ in range(3)]
    bands.run_bands_batched_sets(gpu, engs, bands.BandTopology(0, 1, 0, coef.shape[0]), 3, lambda: None)
    torch.cuda.synchronize()
    assert torch.equal(engs[0].coef, engs[1].coef) and torch.equal(engs[0].coef, engs[2].coef)
    _diff(engs[0].coef.cpu().numpy(), want, f"q{quality} plane-set launch")
    del engs, eng

    # (iii) 8 logical bands of the torch.distributed driver
    got = _luma_band_driver(gpu, torch, bands, d_in, quant, flags, 3, 8)
    _diff(got.cpu().numpy(), want, f"q{quality} bands.py, 8 logical bands")
    del got, d_in
    torch.cuda.empty_cache()

    # (iv) job layer on host arrays (fused route, the plane cut into pipelined bands), (v) 8 logical devices
    one = gpu.do_quantsmooth([coef], [quant], flags, 3)
    assert one["ret"] == 0 and (one["quants"][0] == 1).all()
    _diff(one["coefs"][0], want, f"q{quality} qs_hip_do_quantsmooth")
    many = gpu.do_quantsmooth([coef], [quant], flags, 3, devices=[0] * 8)
    assert many["ret"] == 0
    _diff(many["coefs"][0], want, f"q{quality} qs_hip_do_quantsmooth_sharded over 8 logical devices")


def test_full_16384_luma_every_block_vs_reference(gpu, pkg, truth):
    """configs[3]: 16384 x 16384 luma (4,194,304 blocks, 512 MiB of coefficients), --quality 3 niter 3,
    row-sharded 8 ways -- behind the C ABI (8 logical devices: both sides of all 7 band edges are part
    of the whole-plane comparison) and through the band driver bench.py --gpus 8 uses"""
    import torch
    import bench
    from jpeg_quantsmooth_amd import bands
    dev = torch.device("cuda:0")
    d_in, quant = bench.synth_input_gpu(torch, pkg, 16384, 50, dev)
    coef = d_in.cpu().numpy()
    t0 = time.time()
    want = truth.do_quantsmooth([coef], [quant], 0, 3, threads=0)
    print(f"[info] reference ({type(truth).__name__}) 16384^2 q3: {time.time() - t0:.1f} s")
    assert want["ret"] == 0
    want = want["coefs"][0]

    got = _luma_band_driver(gpu, torch, bands, d_in, quant, 0, 3, 8)
    _diff(got.cpu().numpy(), want, "bands.py, 8 logical bands")
    del got, d_in
    torch.cuda.empty_cache()

    many = gpu.do_quantsmooth([coef], [quant], 0, 3, devices=[0] * 8)
    assert many["ret"] == 0
    _diff(many["coefs"][0], want, "qs_hip_do_quantsmooth_sharded over 8 logical devices")
    del many
    one = gpu.do_quantsmooth([coef], [quant], 0, 3)
    assert one["ret"] == 0
    _diff(one["coefs"][0], want, "qs_hip_do_quantsmooth")
    # size-independent properties on top: inside the quantisation interval or clamped
    g = one["coefs"][0].astype(np.int32)
    q = quant.astype(np.int32)
    assert np.abs(g).max() <= 1023
    deq = coef.astype(np.int32) * q
    assert ((np.abs(g - deq) <= q // 2) | (np.abs(g) == 1023)).all()


def test_full_8192_420_q6_n5_every_block_vs_reference(gpu, pkg, truth):
    """configs[4]: 8192 x 8192 4:2:0 YCbCr, --quality 6 (DIAGONALS + JOINT_YUV + UPSAMPLE_UV), niter 5:
    the luma -> low-res luma -> chroma ordering and the upsampled 1024 x 1024-block chroma arrays of
    reference quantsmooth.h:2691-2815 at scale; one device, 8 logical devices, 8 logical colour bands"""
    import torch
    import bench
    from jpeg_quantsmooth_amd import bands as B
    dev = torch.device("cuda:0")
    size, flags, niter = 8192, pkg.flags_for_quality(6), 5
    d_coefs, quants = bench.synth_colour_gpu(torch, pkg, size, 50, dev)
    coefs = [t.cpu().numpy() for t in d_coefs]
    hsamp, vsamp = [2, 1, 1], [2, 1, 1]
    kw = dict(hsamp=hsamp, vsamp=vsamp, colorspace=3, image_size=(size, size))
    t0 = time.time()
    want = truth.do_quantsmooth(coefs, quants, flags, niter, threads=0, **kw)
    print(f"[info] reference ({type(truth).__name__}) 8192^2 4:2:0 q6 n5: {time.time() - t0:.1f} s")
    assert want["ret"] == 0 and want["up"] and (want["hsamp0"], want["vsamp0"]) == (1, 1)
    assert all(c.shape == (1024, 1024, 64) for c in want["coefs"])

    # 8 logical bands of the torch.distributed colour driver (bench.py --quality 6 --gpus 8)
    hby, hbc = size // 8, size // 16
    bl = []
    for r, (y0, y1, c0, c1) in enumerate(B.colour_band_split(hby, hbc, 2, 8)):
        topo = B.BandTopology(r, 8, c0, c1)
        mine = [d_coefs[0][y0:y1].clone(), d_coefs[1][c0:c1].clone(), d_coefs[2][c0:c1].clone()]
        b = B.ColourBand(gpu, torch, mine, quants, hsamp, vsamp, (size, size), flags, niter, topo, dev)
        b.chroma_row0 = c0
        bl.append(b)
    B.run_colour_bands(bl, B.exchange_rows_local)
    torch.cuda.synchronize()
    assert not any(e.bad_coef() for b in bl for e in b.eng)
    _diff(torch.cat([b.eng[0].coef for b in bl]).cpu().numpy(), want["coefs"][0], "bands.py Y")
    for ci in (1, 2):
        _diff(torch.cat([b.up[ci - 1] for b in bl]).cpu().numpy(), want["coefs"][ci], f"bands.py upsampled component {ci}")
    del bl, d_coefs
    torch.cuda.empty_cache()

    from helpers import assert_same_result
    one = gpu.do_quantsmooth(coefs, quants, flags, niter, **kw)
    assert_same_result(one, want, "qs_hip_do_quantsmooth")
    del one
    many = gpu.do_quantsmooth(coefs, quants, flags, niter, devices=[0] * 8, **kw)
    assert_same_result(many, want, "qs_hip_do_quantsmooth_sharded over 8 logical devices")


def test_32768_luma_beyond_the_baseline_sizes(gpu, pkg, truth):
    """Four times the largest BASELINE plane: 32768 x 32768 luma (16,777,216 blocks, 2 GiB of coefficients; libjpeg's own
    limit is 65500 x 65500), --quality 3 niter 3, through the job layer (the plane travels as pipelined halo bands) and over
    8 logical devices on both band schedules.  The reference would need minutes for the whole plane, so it runs on crops:
    16 windows of 8 block rows -- the top, the bottom and 14 seeded random positions -- each with niter + 1 margin rows
    (a block's result depends only on blocks within niter rows of it); the three routes must agree on EVERY block."""
    import torch
    import bench
    dev = torch.device("cuda:0")
    tile, quant = bench.synth_input_gpu(torch, pkg, 8192, 50, dev)
    coef = np.tile(tile.cpu().numpy(), (4, 4, 1))            # seams: real neighbours, nothing special-cased
    del tile
    torch.cuda.empty_cache()
    assert coef.shape == (4096, 4096, 64)
    flags, niter, m = 0, 3, 4
    t0 = time.time()
    one = gpu.do_quantsmooth([coef], [quant], flags, niter)
    print(f"[info] 32768^2 q3 through qs_hip_do_quantsmooth: {time.time() - t0:.2f} s (incl. the 2 GiB input copy of the binding)")
    assert one["ret"] == 0
    got = one["coefs"][0]
    rng = np.random.default_rng(32768)
    starts = [0, 4096 - 8] + [int(v) for v in rng.integers(8, 4096 - 16, 14)]
    bad = 0
    for a in starts:
        lo, hi = max(0, a - m), min(4096, a + 8 + m)
        want = truth.do_quantsmooth([np.ascontiguousarray(coef[lo:hi])], [quant], flags, niter, threads=0)["coefs"][0][a - lo:a - lo + 8]
        bad += int((got[a:a + 8] != want).any(axis=2).sum())
    assert bad == 0, f"{bad} blocks differ from the reference in the 16 checked windows"
    g = got.astype(np.int32)
    assert np.abs(g).max() <= 1023
    for sched in (0, 1):
        gpu.set_shard_schedule(sched)
        try:
            many = gpu.do_quantsmooth([coef], [quant], flags, niter, devices=[0] * 8)
        finally:
            gpu.set_shard_schedule(-1)
        assert many["ret"] == 0
        assert np.array_equal(many["coefs"][0], got), f"8 logical devices, schedule {sched}: differs from the one-device result"
        del many


def test_65500_luma_the_largest_jpeg(gpu, pkg, truth):
    """The largest image a JPEG file can hold: 65500 x 65500 (8188 x 8188 = 67,043,344 blocks, 8.0 GiB of coefficients),
    luma, --quality 3 niter 3, through qs_hip_do_quantsmooth on one device.  Reference on 10 windows of 8 block rows (top,
    bottom, 8 seeded random positions), each over the full width; every coefficient inside its interval or clamped."""
    import torch
    import bench
    dev = torch.device("cuda:0")
    tile, quant = bench.synth_input_gpu(torch, pkg, 8192, 50, dev)
    coef = np.ascontiguousarray(np.tile(tile.cpu().numpy(), (8, 8, 1))[:8188, :8188])
    del tile
    torch.cuda.empty_cache()
    flags, niter, m = 0, 3, 4
    t0 = time.time()
    one = gpu.do_quantsmooth([coef], [quant], flags, niter, image_size=(65500, 65500))
    print(f"[info] 65500^2 q3 through qs_hip_do_quantsmooth: {time.time() - t0:.2f} s (incl. the 8 GiB input copy of the binding)")
    assert one["ret"] == 0
    got = one["coefs"][0]
    rng = np.random.default_rng(65500)
    bad = 0
    for a in [0, 8188 - 8] + [int(v) for v in rng.integers(8, 8188 - 16, 8)]:
        lo, hi = max(0, a - m), min(8188, a + 8 + m)
        want = truth.do_quantsmooth([np.ascontiguousarray(coef[lo:hi])], [quant], flags, niter, threads=0)["coefs"][0][a - lo:a - lo + 8]
        bad += int((got[a:a + 8] != want).any(axis=2).sum())
    assert bad == 0, f"{bad} blocks differ from the reference in the 10 checked windows"
    q = quant.astype(np.int32)
    for r in range(0, 8188, 1024):                            # (in slices: int32 copies of the whole plane would be 34 GB)
        g = got[r:r + 1024].astype(np.int32)
        deq = coef[r:r + 1024].astype(np.int32) * q
        assert np.abs(g).max() <= 1023
        assert ((np.abs(g - deq) <= q // 2) | (np.abs(g) == 1023)).all()
