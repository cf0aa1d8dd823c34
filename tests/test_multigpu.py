"""First contact with more than one GPU (Next 3): the three steps of tools/first_contact.sh as tests.

Nothing in this repository has run on two devices: the development box and the driver's GPU test box have one.  These
tests make a multi-GPU `pytest -m gpu` exercise the real transports by itself:
  (a) the peer-copy + cross-device event protocol of csrc/qs_shard.cpp alone (tools/first_contact_p2p),
  (b) qs_hip_do_quantsmooth_sharded over real devices, every block against the compiled reference,
  (c) bench.py --gpus N over RCCL, band edges verified, `rccl_ranks == N`.
With fewer than two devices the N >= 2 cases SKIP (reported with -rs); each step's degenerate N = 1 form runs
everywhere, so the scripts themselves cannot rot.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _ndev(gpu):
    return gpu.device_count()


def _counts(gpu):
    n, out = 2, []
    while n <= _ndev(gpu):
        out.append(n)
        n *= 2
    return out


def _p2p_binary():
    exe = ROOT / "tools" / "first_contact_p2p"
    if not exe.exists():
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", str(ROOT / "tools" / "first_contact_p2p.hip"), "-o", str(exe)],
                       check=True, timeout=600)
    return exe


def _run_p2p(n):
    r = subprocess.run([str(_p2p_binary()), str(n), "32"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "first_contact_p2p: PASS" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]
    assert "row ring: 0 wrong words" in r.stdout and "bulk ring: 0 wrong sampled words" in r.stdout
    return r.stdout


def test_multigpu_p2p_ring_degenerate_one_device(gpu):
    """(a) at N = 1: the A / X event protocol with a device copying to itself"""
    _run_p2p(1)


def test_multigpu_p2p_ring_real_devices(gpu):
    """(a): hipDeviceEnablePeerAccess + hipMemcpyPeerAsync on the receiver's stream behind the sender's event, for
    every power-of-two device count the node has"""
    if _ndev(gpu) < 2:
        pytest.skip(f"needs >= 2 HIP devices, {_ndev(gpu)} visible")
    for n in _counts(gpu):
        print(_run_p2p(n))


def _run_shard(devices, small):
    cmd = [sys.executable, str(ROOT / "tools" / "first_contact_shard.py"), "--devices", ",".join(map(str, devices))]
    if small:
        cmd.append("--small")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1700, cwd=str(ROOT))
    assert r.returncode == 0 and "first_contact_shard: PASS" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count(": OK --") == 3, r.stdout[-3000:]
    return r.stdout


def test_multigpu_sharded_route_degenerate_one_device(gpu):
    """(b) at N = 1 and reduced sizes: one band per configuration, no exchange; every block vs the reference"""
    _run_shard([0], small=True)


def test_multigpu_sharded_route_real_devices_every_block(gpu):
    """(b): qs_hip_do_quantsmooth_sharded over devices 0..N-1 at FULL size (8192^2 q3, 16384^2 q3, 8192^2 4:2:0 q6 n5),
    every block against oracle/_ref/libqsref_none.so and against the one-device result"""
    if _ndev(gpu) < 2:
        pytest.skip(f"needs >= 2 HIP devices, {_ndev(gpu)} visible")
    n = _counts(gpu)[-1]
    print(_run_shard(list(range(n)), small=False))
    if n > 2:
        print(_run_shard([0, 1], small=True))


def _bench(extra, timeout=900):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *extra], capture_output=True, text=True, timeout=timeout, cwd=str(ROOT), env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_multigpu_bench_rccl_real_devices(gpu):
    """(c): `python bench.py --gpus N` over RCCL (the default back end; it ENDS the run if RCCL does not come up):
    band edges verified against the reference, rccl_ranks == N, single-image leg and the product's own route present;
    then q6 (colour bands) at the largest count"""
    if _ndev(gpu) < 2:
        pytest.skip(f"needs >= 2 HIP devices, {_ndev(gpu)} visible")
    for n in _counts(gpu):
        d = _bench(["--gpus", str(n), "--steps", "5", "--warmup", "2", "--no-cpu-baseline"])
        assert d["n_gpus"] == n and d["config"]["rccl_ranks"] == n and d["scaling"] == "strong"
        assert d["verify_ok"] is True and d["verify_band_edges_ok"] is True
        assert d["value"] > 0 and d["value_batch1"] > 0
        pr = d["product_route"]
        assert pr.get("error") is None and pr["verify_ok"] is True and pr["devices"] == list(range(n)), pr
        print(f"[info] N = {n}: value {d['value'] / 1e6:.1f} M blocks/s, batch 1 {d['value_batch1'] / 1e6:.1f}, product route {pr['ms_per_image']:.2f} ms")
    n = _counts(gpu)[-1]
    d = _bench(["--gpus", str(n), "--quality", "6", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert d["n_gpus"] == n and d["config"]["rccl_ranks"] == n and d["verify_ok"] is True


def test_multigpu_first_contact_script_runs_on_this_box(gpu, tmp_path):
    """the one-command script itself, on whatever this box has (N = 1: every step in its degenerate form) -- only the
    cheap steps: (a), and a syntax check of the whole script; (b)-(d) are the tests above"""
    subprocess.run(["bash", "-n", str(ROOT / "tools" / "first_contact.sh")], check=True)
    text = (ROOT / "tools" / "first_contact.sh").read_text()
    for piece in ("first_contact_p2p", "first_contact_shard.py", "bench.py --gpus", "tests/test_multigpu.py", "SUMMARY.txt"):
        assert piece in text


# ---- qs_hip_do_quantsmooth_band: one band per process / thread, halo rows through RCCL behind the C ABI ----------------------

def _rccl():
    import ctypes as C
    import importlib.util
    # ONE RCCL (and one HIP runtime) per process: a PyTorch wheel bundles its own librccl.so next to its own libamdhip64;
    # loading /opt/rocm's copy first and torch's later ends in a double free at exit.  Take torch's when torch is installed.
    cand = []
    spec = importlib.util.find_spec("torch")
    if spec and spec.origin:
        cand.append(str(Path(spec.origin).parent / "lib" / "librccl.so"))
    cand += ["librccl.so.1", "librccl.so"]
    lib = None
    for name in cand:
        if name.startswith("/") and not Path(name).exists():
            continue
        try:
            lib = C.CDLL(name, mode=os.RTLD_GLOBAL)
            break
        except OSError:
            continue
    if lib is None:
        pytest.skip("librccl is not loadable on this box")
    lib.ncclCommInitAll.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    return lib


def _band_job_and_truth(synth, oracle, flags, niter):
    j = synth.synth_ycc(264, 1040, 1, 1, quality=45, seed=17)   # 130 x 33 blocks per component, 4:4:4: every component cut alike
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(264, 1040))
    return j, kw, oracle.do_quantsmooth(j["coefs"], j["quants"], flags, niter, threads=8, **kw)


def test_multigpu_rccl_band_entry_degenerate_one_rank(gpu, synth, oracle):
    """qs_hip_do_quantsmooth_band with nranks = 1: the whole image is the band, no neighbour -- once without a
    communicator, once with a real one-rank RCCL communicator (librccl looked up in the caller's copy, the range-check
    flags all-reduced through it).  Bit-exact; the coupled flags are refused; a tripped range check writes nothing."""
    import ctypes as C
    from helpers import assert_same_result, load_golden
    rccl = _rccl()
    comm = C.c_void_p()
    dev = (C.c_int * 1)(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, dev) == 0
    try:
        for flags, niter in ((0, 3), (1, 2), (32, 2)):
            j, kw, want = _band_job_and_truth(synth, oracle, flags, niter)
            for c in (None, comm):
                got = gpu.do_quantsmooth_band(j["coefs"], j["quants"], flags, niter, 0, 1, c, **kw)
                assert_same_result(got, want, f"band entry, one rank, flags {flags}, comm {c is not None}")
        with pytest.raises(Exception) as e:
            gpu.do_quantsmooth_band(j["coefs"], j["quants"], 7, 2, 0, 1, comm, **kw)
        assert getattr(e.value, "code", None) == -4                     # QS_HIP_ENOTSUP
        job, _ = load_golden("gray64_badcoef_q3_n2")
        got = gpu.do_quantsmooth_band(job["coefs"], job["quants"], job["flags"], job["niter"], 0, 1, comm, **job["kw"])
        assert got["ret"] == 2                                            # QS_HIP_BAND_RANGE_CHECK
        assert all((a == b).all() for a, b in zip(got["coefs"], job["coefs"]))   # nothing written
    finally:
        rccl.ncclCommDestroy(comm)


def test_multigpu_rccl_band_entry_real_devices(gpu, synth, oracle):
    """N threads, one per device, each with its band of the image and its rank of one RCCL clique (ncclCommInitAll):
    ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd between the iterations -- the bands put together equal the
    unsharded result"""
    import ctypes as C
    import threading
    import numpy as np
    if _ndev(gpu) < 2:
        pytest.skip(f"needs >= 2 HIP devices, {_ndev(gpu)} visible")
    rccl = _rccl()
    # the HIP runtime this process already uses (no torch here: importing it AFTER a communicator has lived in the process
    # ended in a double free at interpreter exit on the one-GPU box)
    try:
        hip_rt = C.CDLL(None); hip_rt.hipSetDevice
    except (OSError, AttributeError):
        hip_rt = C.CDLL("libamdhip64.so")
    hip_rt.hipSetDevice.argtypes = [C.c_int]
    n = _counts(gpu)[-1]
    comms = (C.c_void_p * n)()
    devs = (C.c_int * n)(*range(n))
    assert rccl.ncclCommInitAll(comms, n, devs) == 0
    try:
        for flags, niter in ((0, 3), (1, 2)):
            j, kw, want = _band_job_and_truth(synth, oracle, flags, niter)
            hb = j["coefs"][0].shape[0]
            parts, errs = [None] * n, []

            def work(r):
                try:
                    assert hip_rt.hipSetDevice(r) == 0
                    r0, r1 = gpu.band_rows(hb, n, r)
                    parts[r] = gpu.do_quantsmooth_band([c[r0:r1] for c in j["coefs"]], j["quants"], flags, niter, r, n, comms[r], **kw)
                except Exception as ex:  # noqa: BLE001
                    errs.append((r, ex))
            th = [threading.Thread(target=work, args=(r,)) for r in range(n)]
            [t.start() for t in th]; [t.join() for t in th]
            assert not errs, errs
            for ci in range(3):
                got = np.concatenate([p["coefs"][ci] for p in parts], axis=0)
                assert np.array_equal(got, want["coefs"][ci]), (flags, ci)
    finally:
        for c in comms:
            rccl.ncclCommDestroy(c)
