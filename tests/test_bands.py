"""Band sharding: arithmetic, the N > 1 loop under gloo (world_size 2 and 3, CPU
engine from the oracle), and N logical bands on one GPU with the real kernels."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_band_rows_python_restatement_equals_c(pkg):
    """bands.py's fall-back for boxes where the product library cannot be loaded is pinned to the C definition"""
    from jpeg_quantsmooth_amd import bands as B
    lib = B._lib()
    assert lib is not None, "the product library must load here (it carries the definition)"
    for hblk in (0, 1, 7, 8, 25, 128, 1024, 2047):
        for world in (1, 2, 3, 4, 8, 13):
            for align in (1, 2, 4):
                for r in range(world):
                    assert B._band_rows_py(hblk, world, r, align) == lib.band_rows(hblk, world, r, align)
    for hy, hc, vs in ((1024, 512, 2), (135, 68, 2), (9, 9, 1), (17, 5, 4), (64, 64, 1)):
        for world in (1, 2, 3, 8):
            for r in range(world):
                assert B._colour_band_rows_py(hy, hc, vs, world, r) == lib.colour_band_rows(hy, hc, vs, world, r)


def test_band_rows_cover_and_align(pkg):
    from jpeg_quantsmooth_amd.bands import band_rows
    for hblk in (1, 2, 7, 64, 135, 1024, 2048):
        for world in (1, 2, 3, 4, 8):
            for align in (1, 2):
                rows = [band_rows(hblk, world, r, align) for r in range(world)]
                assert rows[0][0] == 0 and rows[-1][1] == hblk
                for (a0, a1), (b0, b1) in zip(rows[:-1], rows[1:]):
                    assert a1 == b0 and a0 <= a1
                    assert a1 % align == 0 or a1 == hblk


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, flags, niter, tmp, overlapped=False):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import torch
    import torch.distributed as dist
    import jpegqs_pkg
    from oracle.oracle import Oracle
    from band_cpu_engine import OracleBandEngine
    pkg = jpegqs_pkg.load()
    from jpeg_quantsmooth_amd import bands
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    coef, quant = pkg.synth.synth_gray(136, 200, 45, seed=21)     # 17 x 25 blocks
    hblk = coef.shape[0]
    r0, r1 = bands.band_rows(hblk, world, rank)
    topo = bands.BandTopology(rank, world, r0, r1)
    eng = OracleBandEngine(Oracle(), pkg.HipQS(), coef[r0:r1].copy(), quant, flags)
    runner = bands.run_band_overlapped if overlapped else bands.run_band
    runner(eng, topo, niter, lambda: bands.exchange_halo_dist(eng, topo, dist))
    assert not eng.bad_coef()
    np.save(os.path.join(tmp, f"band{rank}.npy"), eng.coef)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,overlapped", [(2, False), (3, False), (8, False), (2, True), (4, True)])
@pytest.mark.parametrize("flags", [0, 1])
def test_bands_gloo_equal_unsharded(world, overlapped, flags, oracle, synth, tmp_path):
    """world_size > 1, CPU, gloo: bands + halo exchange reproduce the unsharded result
    bit for bit, with the simple and with the communication-hiding schedule"""
    import torch.multiprocessing as mp
    niter = 3
    port = _free_port()
    mp.spawn(_worker, args=(world, port, flags, niter, str(tmp_path), overlapped), nprocs=world, join=True)
    coef, quant = synth.synth_gray(136, 200, 45, seed=21)
    want = oracle.do_quantsmooth([coef], [quant], flags, niter)["coefs"][0]
    got = np.concatenate([np.load(tmp_path / f"band{r}.npy") for r in range(world)], axis=0)
    assert np.array_equal(got, want)


def _deep_worker(rank, world, port, flags, niter, tmp):
    """one rank of the communication-avoiding schedule: holds niter extra block rows per cut side, exchanges NOTHING
    during the iterations (the process group only carries the final barrier)"""
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import torch
    import torch.distributed as dist
    import jpegqs_pkg
    from oracle.oracle import Oracle
    from band_cpu_engine import OracleBandEngine
    pkg = jpegqs_pkg.load()
    from jpeg_quantsmooth_amd import bands
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    coef, quant = pkg.synth.synth_gray(136, 520, 45, seed=21)     # 65 x 17 blocks
    r0, r1, e0, e1 = bands.deep_band_rows(coef.shape[0], world, rank, niter)
    eng = OracleBandEngine(Oracle(), pkg.HipQS(), coef[e0:e1].copy(), quant, flags)
    if rank & 1:
        del OracleBandEngine.smooth_next                             # odd ranks: the unfused loop of run_band_deep
    bands.run_band_deep(eng, niter)
    assert not eng.bad_coef()
    np.save(os.path.join(tmp, f"band{rank}.npy"), eng.coef[r0 - e0:r1 - e0])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,niter", [(2, 3), (3, 2), (4, 5), (8, 1)])
@pytest.mark.parametrize("flags", [0, 1])
def test_deep_halo_bands_gloo_equal_unsharded(world, niter, flags, oracle, synth, tmp_path):
    """world_size > 1, CPU, gloo: the communication-avoiding schedule (bands.deep_band_rows / run_band_deep, the
    one-process-per-GPU form of qs_hip_set_shard_schedule(1)) reproduces the unsharded result bit for bit with ZERO halo
    exchanges -- niter extra block rows per cut side absorb the error of treating the cuts as image edges"""
    import torch.multiprocessing as mp
    mp.spawn(_deep_worker, args=(world, _free_port(), flags, niter, str(tmp_path)), nprocs=world, join=True)
    coef, quant = synth.synth_gray(136, 520, 45, seed=21)
    want = oracle.do_quantsmooth([coef], [quant], flags, niter)["coefs"][0]
    got = np.concatenate([np.load(tmp_path / f"band{r}.npy") for r in range(world)], axis=0)
    assert np.array_equal(got, want)


def test_deep_band_rows_arithmetic(pkg):
    from jpeg_quantsmooth_amd import bands
    for hblk, world, niter in ((128, 8, 3), (65, 4, 5), (1024, 8, 3), (17, 2, 20)):
        for rank in range(world):
            r0, r1, e0, e1 = bands.deep_band_rows(hblk, world, rank, niter)
            assert (r0, r1) == bands.band_rows(hblk, world, rank)
            assert e0 == max(0, r0 - niter) and e1 == min(hblk, r1 + niter) and e0 <= r0 <= r1 <= e1


def _batched_worker(rank, world, port, tmp, packed=False):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import torch
    import torch.distributed as dist
    import jpegqs_pkg
    from oracle.oracle import Oracle
    from band_cpu_engine import OracleBandEngine
    pkg = jpegqs_pkg.load()
    from jpeg_quantsmooth_amd import bands
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    engs = []
    hip = pkg.HipQS()
    r0, r1 = bands.band_rows(25, world, rank)
    planes2d = torch.zeros((3, (hip.plane_bytes(17, r1 - r0) + 255) & ~255), dtype=torch.uint8)
    for n, seed in enumerate((21, 22, 23)):                      # three independent planes = one batch
        coef, quant = pkg.synth.synth_gray(136, 200, 45, seed=seed)
        assert coef.shape[:2] == (25, 17)
        engs.append(OracleBandEngine(Oracle(), hip, coef[r0:r1].copy(), quant, 1, plane=planes2d[n] if packed else None))
    topo = bands.BandTopology(rank, world, r0, r1)
    if packed == "fused":   # bench.py's schedule: pass A once, then per iteration one packed exchange of the CURRENT planes + pass B
        planes2d_b = torch.zeros_like(planes2d)
        for n, e in enumerate(engs):
            e.plane2 = planes2d_b[n]

        def exch():
            cur = planes2d if engs[0].plane.data_ptr() == planes2d[0].data_ptr() else planes2d_b
            bands.exchange_halo_packed(hip, cur, 17, r1 - r0, topo, dist)
        bands.run_bands_batched_fused(engs, topo, 3, exch)
    elif packed:      # the exchange bench.py uses: one packed send + receive per neighbour for the whole batch
        bands.run_bands_batched(engs, topo, 3, lambda: bands.exchange_halo_packed(hip, planes2d, 17, r1 - r0, topo, dist))
    else:
        bands.run_bands_batched(engs, topo, 3, lambda: bands.exchange_halo_dist_many(engs, topo, dist))
    for n, e in enumerate(engs):
        np.save(os.path.join(tmp, f"plane{n}_band{rank}.npy"), e.coef)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("packed", [False, True, "fused"])
def test_batched_bands_gloo_equal_unsharded(packed, oracle, synth, tmp_path):
    """the schedule bench.py uses for N > 1: the planes of a batch advance together, ONE batched
    halo exchange per iteration for all of them -- per-plane messages, or (bench.py) the rows of the
    whole batch packed into one message per neighbour (world_size 3, gloo, CPU engine)"""
    import torch.multiprocessing as mp
    world = 3
    mp.spawn(_batched_worker, args=(world, _free_port(), str(tmp_path), packed), nprocs=world, join=True)
    for n, seed in enumerate((21, 22, 23)):
        coef, quant = synth.synth_gray(136, 200, 45, seed=seed)
        want = oracle.do_quantsmooth([coef], [quant], 1, 3)["coefs"][0]
        got = np.concatenate([np.load(tmp_path / f"plane{n}_band{r}.npy") for r in range(world)], axis=0)
        assert np.array_equal(got, want), f"plane {n}"


@pytest.mark.gpu
@pytest.mark.parametrize("nbands", [2, 5])
def test_bands_on_one_gpu_equal_unsharded(gpu, pkg, oracle, synth, nbands):
    """the GPU band path (rep_top/rep_bot flags + halo rows) with N logical bands on one device"""
    import torch
    from jpeg_quantsmooth_amd import bands
    coef, quant = synth.synth_gray(264, 328, 50, seed=4)          # 41 x 33 blocks
    hblk = coef.shape[0]
    dev = torch.device("cuda:0")
    for flags in (0, 1):
        engines, topos = [], []
        for r in range(nbands):
            r0, r1 = bands.band_rows(hblk, nbands, r)
            topos.append(bands.BandTopology(r, nbands, r0, r1))
            engines.append(bands.HipBandEngine(gpu, torch, torch.from_numpy(coef[r0:r1].copy()).to(dev), quant, flags))
        niter = 3
        for it in range(niter):
            for e, t in zip(engines, topos):
                e.idct(it == 0, t.rep_top, t.rep_bot)
            # interior rows first (they do not read the halo), then the exchange, then the edge rows:
            # the order run_band_overlapped produces on a real multi-GPU run
            for e, t in zip(engines, topos):
                lo = 1 if t.up is not None else 0
                hi = e.hblk - 1 if t.down is not None else e.hblk
                e.smooth_rows(lo, hi, it == niter - 1)
            bands.exchange_halo_local(engines)
            for e, t in zip(engines, topos):
                if t.up is not None:
                    e.smooth_rows(0, 1, it == niter - 1)
                if t.down is not None:
                    e.smooth_rows(e.hblk - 1, e.hblk, it == niter - 1)
        torch.cuda.synchronize()
        got = np.concatenate([e.coef.cpu().numpy() for e in engines], axis=0)
        want = oracle.do_quantsmooth([coef], [quant], flags, niter)["coefs"][0]
        assert np.array_equal(got, want), f"flags={flags}"


@pytest.mark.gpu
@pytest.mark.parametrize("size,samp,nbands", [((256, 160), (2, 2), 2), ((333, 517), (2, 2), 3), ((321, 200), (2, 2), 2),
                                               ((160, 96), (1, 1), 2), ((208, 128), (2, 1), 2)])
def test_colour_bands_on_one_gpu_equal_unsharded(gpu, pkg, oracle, synth, size, samp, nbands):
    """BASELINE config 4 shape (YCbCr, JOINT_YUV + UPSAMPLE_UV) cut into N logical
    bands on one device: luma/chroma halos per iteration, one-time halo of the
    low-res luma and of the refreshed chroma, band-local downsample / upsample /
    re-FDCT -- bit-exact against the unsharded oracle"""
    import torch
    from jpeg_quantsmooth_amd import bands as B
    w, h = size
    hs_, vs_ = samp
    from helpers import inject_extreme_blocks
    # extreme blocks: coefficients beyond +-1023 before the final clamp (the refresh passes must see them unclamped)
    j = inject_extreme_blocks(synth.synth_ycc(w, h, hs_, vs_, quality=40, seed=12))
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(w, h))
    dev = torch.device("cuda:0")
    hby, hbc = j["hblk"][0], j["hblk"][1]
    split = B.colour_band_split(hby, hbc, vs_, nbands)
    for flags, niter in ((7, 2), (3, 2), (7, 0), (7 | 32, 1), (15, 2), (5, 1)):
        bl = []
        for r, (y0, y1, c0, c1) in enumerate(split):
            topo = B.BandTopology(r, nbands, c0, c1)
            coefs = [torch.from_numpy(j["coefs"][0][y0:y1].copy()).to(dev),
                     torch.from_numpy(j["coefs"][1][c0:c1].copy()).to(dev),
                     torch.from_numpy(j["coefs"][2][c0:c1].copy()).to(dev)]
            b = B.ColourBand(gpu, torch, coefs, j["quants"], j["hsamp"], j["vsamp"], (w, h), flags, niter, topo, dev)
            b.chroma_row0 = c0
            bl.append(b)
        B.run_colour_bands(bl, B.exchange_rows_local)
        torch.cuda.synchronize()
        want = oracle.do_quantsmooth(j["coefs"], j["quants"], flags, niter, **kw)
        gotY = np.concatenate([b.eng[0].coef.cpu().numpy() for b in bl], axis=0)
        assert np.array_equal(gotY, want["coefs"][0]), f"Y flags={flags} niter={niter}"
        for ci in (1, 2):
            if want["up"]:
                got = np.concatenate([b.up[ci - 1].cpu().numpy() for b in bl], axis=0)
            else:
                got = np.concatenate([b.eng[ci].coef.cpu().numpy() for b in bl], axis=0)
            assert got.shape == want["coefs"][ci].shape, (got.shape, want["coefs"][ci].shape)
            assert np.array_equal(got, want["coefs"][ci]), f"comp {ci} flags={flags} niter={niter}"


def _rows_worker(rank, world, port, tmp):
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist
    import jpegqs_pkg
    pkg = jpegqs_pkg.load()
    from jpeg_quantsmooth_amd import bands as B
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hip = pkg.HipQS()
    wblk, hblk = 5, 2 + rank
    t = torch.full((hip.plane_bytes(wblk, hblk),), 100 + rank, dtype=torch.uint8)
    rows = B.PlaneRows(hip, t, wblk, hblk)
    rows.row(0)[:] = 10 + rank            # first pixel row
    rows.row(hblk * 8 - 1)[:] = 50 + rank   # last pixel row
    topo = B.BandTopology(rank, world, 0, hblk)
    B.exchange_rows_dist(rows, topo, dist)
    top, bot = int(rows.row(-1)[20]), int(rows.row(hblk * 8)[20])
    exp_top = 50 + rank - 1 if rank > 0 else 100 + rank          # neighbour's last row, or untouched
    exp_bot = 10 + rank + 1 if rank < world - 1 else 100 + rank
    assert (top, bot) == (exp_top, exp_bot), (rank, top, bot)
    dist.barrier(); dist.destroy_process_group()


def test_plane_rows_exchange_gloo(tmp_path):
    """the generic row exchange used by the colour band driver, world_size 3, bands of unequal height"""
    import torch.multiprocessing as mp
    mp.spawn(_rows_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--edge-first"], ["--overlap", "--no-extras"], ["--backend", "nccl", "--no-extras"]])
def test_bench_sharded_path_two_processes_one_gpu(gpu, extra):
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one
    process per rank, band split, comm side stream, overlapped schedule), but on ONE
    GPU with the gloo back end and host-staged halo rows; the rows on both sides of
    the band edge are checked against the oracle inside bench.py (--verify)"""
    import json
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device",
           "--size", "1024", "--steps", "2", "--warmup", "1", "--verify", "--no-cpu-baseline", *extra]
    # (the "--backend nccl" case: two ranks on one device is something RCCL refuses.  A run that asked for RCCL must then
    # END, non-zero and without a result line -- a host-staged gloo number must never look like an RCCL result)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    if "nccl" in extra:
        assert r.returncode != 0, r.stdout[-2000:]
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")], r.stdout[-2000:]
        assert "RCCL did not come up" in r.stderr and "--backend gloo" in r.stderr, r.stderr[-2000:]
        return
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["verify_band_edges_ok"] is True and d["verify_ok"] is True
    if "--no-extras" not in extra:
        # the extra legs: single-plane steps over the same bands, and the product's own route
        # (qs_hip_do_quantsmooth_sharded over two logical devices) run as a child process of rank 0
        assert d["single_plane_ms"] > 0 and d["value_batch1"] > 0 and d["planes_identical"] is True
        pr = d["product_route"]
        assert pr.get("error") is None, pr
        assert pr["entry"] == "qs_hip_do_quantsmooth_sharded" and pr["devices"] == [0, 0]
        assert pr["equals_one_device_result"] is True and pr["verify_ok"] is True
        # ... and the communication-avoiding schedule: same rows as the exchange schedule, without any exchange
        dh = d["deep_halo_schedule"]
        assert dh["equals_exchange_schedule"] is True and dh["value_batch1"] > 0 and dh["rows_owned_and_held"] == [64, 67]
        # ... and the latency-hiding one (edge rows + exchange on a side stream, interior rows on the main stream)
        ef = d["edge_first_schedule"]
        assert ef["equals_exchange_schedule"] is True and ef["value_batch1"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("nbands", [2, 5])
@pytest.mark.parametrize("size", [(264, 328), (4096, 4096)])
def test_fused_bands_on_one_gpu_equal_unsharded(gpu, pkg, oracle, synth, nbands, size):
    """the FUSED band schedule (pass A once; every pass B but the last writes the next iteration's plane, halo rows are
    exchanged on whichever plane is current) with N logical bands on one device: per-plane launches
    (qs_hip_smooth_plane_next through HipBandEngine.smooth_next; the small-plane kernel except for the two 131 k-block
    bands of the 4096^2 plane, which take the one-block-per-lane kernel) -- bit-exact against the unsharded oracle"""
    import torch
    from jpeg_quantsmooth_amd import bands
    coef, quant = synth.synth_gray(size[0], size[1], 50, seed=4)
    hblk = coef.shape[0]
    dev = torch.device("cuda:0")
    niter = 3
    for flags in (0, 1):
        engines, topos = [], []
        for r in range(nbands):
            r0, r1 = bands.band_rows(hblk, nbands, r)
            topos.append(bands.BandTopology(r, nbands, r0, r1))
            engines.append(bands.HipBandEngine(gpu, torch, torch.from_numpy(coef[r0:r1].copy()).to(dev), quant, flags))
        for e, t in zip(engines, topos):
            e.idct(True, t.rep_top, t.rep_bot)
        for it in range(niter):
            bands.exchange_halo_local(engines)                     # rows of the engines' CURRENT planes
            for e, t in zip(engines, topos):
                e.smooth_next(it == niter - 1, it < niter - 1, t.rep_top, t.rep_bot)
        torch.cuda.synchronize()
        got = np.concatenate([e.coef.cpu().numpy() for e in engines], axis=0)
        want = oracle.do_quantsmooth([coef], [quant], flags, niter, threads=8)["coefs"][0]
        assert np.array_equal(got, want), f"flags={flags}"


@pytest.mark.gpu
@pytest.mark.parametrize("nbands", [2, 8])
def test_deep_halo_bands_on_one_gpu_equal_unsharded(gpu, pkg, oracle, synth, nbands):
    """the communication-avoiding band schedule through HipBandEngine (bands.run_band_deep) with N logical bands on one
    device: no halo row moves, every band runs niter extra block rows per cut side -- bit-exact against the unsharded oracle"""
    import torch
    from jpeg_quantsmooth_amd import bands
    coef, quant = synth.synth_gray(520, 1040, 50, seed=4)
    hblk = coef.shape[0]
    dev = torch.device("cuda:0")
    for flags, niter in ((0, 3), (1, 2), (0, 5)):
        parts = []
        for r in range(nbands):
            r0, r1, e0, e1 = bands.deep_band_rows(hblk, nbands, r, niter)
            eng = bands.HipBandEngine(gpu, torch, torch.from_numpy(coef[e0:e1].copy()).to(dev), quant, flags)
            bands.run_band_deep(eng, niter)
            assert not eng.bad_coef()
            parts.append(eng.coef[r0 - e0:r1 - e0].cpu().numpy())
        want = oracle.do_quantsmooth([coef], [quant], flags, niter, threads=8)["coefs"][0]
        assert np.array_equal(np.concatenate(parts, axis=0), want), f"flags={flags} niter={niter}"


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(264, 328), (1024, 2048), (520, 16)])
def test_edge_first_schedule_whole_plane_equals_unsharded(gpu, pkg, oracle, synth, size):
    """bands.run_band_edge_first on a band without neighbours (the whole plane): three VIEWS of one coefficient array and
    one pair of pixel planes -- first block row, last block row, interior -- run as two concurrent launches per iteration
    on two streams; image edges replicate.  Must equal the unsharded oracle (the two-rank form runs in
    test_bench_sharded_path_two_processes_one_gpu); a two-row plane takes the fall-back."""
    import torch
    from jpeg_quantsmooth_amd import bands
    coef, quant = synth.synth_gray(size[0], size[1], 50, seed=4)
    dev = torch.device("cuda:0")
    main, side = torch.cuda.current_stream(), torch.cuda.Stream(device=dev)
    topo = bands.BandTopology(0, 1, 0, coef.shape[0])
    for flags, niter in ((0, 3), (1, 2), (16, 1)):
        eng = bands.HipBandEngine(gpu, torch, torch.from_numpy(coef.copy()).to(dev), quant, flags)
        bands.run_band_edge_first(gpu, eng, topo, niter, lambda: None, main, side, torch)
        torch.cuda.synchronize()
        assert not eng.bad_coef()
        want = oracle.do_quantsmooth([coef], [quant], flags, niter, threads=8)["coefs"][0]
        assert np.array_equal(eng.coef.cpu().numpy(), want), f"flags={flags} niter={niter}"


@pytest.mark.gpu
def test_bench_gpus2_plain_invocation_self_launches(gpu):
    """`python3 bench.py --gpus 2 ...` exactly as a driver types it -- NO launcher, a clean environment (no RANK /
    WORLD_SIZE / MASTER_*): bench.py re-executes itself under torch.distributed.run, one rank per band, and prints ONE
    JSON line with the single-image legs and the product's own route next to `value`."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device",
           "--size", "1024", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(ROOT), env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 2 and d["warmup"] == 1
    assert d["verify_band_edges_ok"] is True and d["verify_ok"] is True
    assert d["value_batch1"] > 0 and d["single_plane_ms"] > 0
    assert d["product_route"].get("error") is None and d["product_route"]["verify_ok"] is True
    assert d["config"]["rccl_ranks"] == 0            # gloo on one device; RCCL runs report rccl_ranks == n_gpus


def test_bench_self_launch_command_line(monkeypatch):
    """CPU: the command bench.py replaces itself with for --gpus N (no launcher in the environment)"""
    import bench
    seen = {}

    def fake_execve(exe, argv, env):
        seen.update(exe=exe, argv=argv, env=env)
        raise SystemExit(0)
    monkeypatch.setattr(bench.os, "execve", fake_execve)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit):
        bench.main()
    a = seen["argv"]
    assert a[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in a
    assert a[a.index("--nproc-per-node") + 1] == "4" and a[a.index("--master-addr") + 1] == "127.0.0.1"
    assert a[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and a[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under a launcher (WORLD_SIZE / RANK set) and for --gpus 1 it must NOT re-launch
    import argparse
    assert bench._needs_self_launch(argparse.Namespace(gpus=4), {}) is True
    assert bench._needs_self_launch(argparse.Namespace(gpus=4), {"WORLD_SIZE": "4"}) is False
    assert bench._needs_self_launch(argparse.Namespace(gpus=4), {"RANK": "0"}) is False
    assert bench._needs_self_launch(argparse.Namespace(gpus=1), {}) is False


@pytest.mark.gpu
@pytest.mark.parametrize("quality,size", [(6, 512), (5, 384)])
def test_bench_sharded_colour_two_processes_one_gpu(gpu, quality, size):
    """BASELINE configs[4] shape through bench.py as the driver launches it for N = 2: one process
    per rank, ColourBand + run_colour_band_dist order (halos of luma, low-res luma and chroma over
    torch.distributed), real kernels, both ranks on ONE GPU with gloo and host-staged halo rows.
    Every rank compares its whole band (upsampled chroma included) with the oracle's result."""
    import json
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device", "--quality", str(quality),
           "--size", str(size), "--niter", "2", "--steps", "1", "--warmup", "1", "--batch", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "4:2:0" in d["config"]["workload"]
    assert d["verify_band_edges_ok"] is True and d["verify_ok"] is True
