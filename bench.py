#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X jpeg-quantsmooth hot path.

Metric (BASELINE.json): 8x8 blocks/s at q=3 niter=3 on a synthetic 8192x8192 luma
plane, inputs resident in HBM.  A "step" is one pass of the hot path over one BATCH of
synthetic input: `--batch` (default 12) independent 8192x8192 planes, each taken through
a complete do_quantsmooth -- the IDCT-to-plane kernel once, then niter x {[halo exchange],
recovery kernel}: every recovery launch but the last also writes the next iteration's pixel
planes (pass A fused into pass B, LABNOTES.md 4.2f); final clamp fused into the last launch.  The planes of a step travel
together as one plane set: one launch per pass covers all of them (the job layer's
qs_hip_idct_planes / qs_hip_smooth_planes), at every N.  (20 driver steps of 12 planes cover
about a second, i.e. the power-capped steady state; `value` counts blocks.)

  python bench.py [--gpus N --steps K --warmup W]      (N > 1 without a launcher: re-executes itself under
                                                        torch.distributed.run, one rank per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: every plane is split into N contiguous block-row bands (strong scaling, total
work fixed, as BASELINE.json's north_star asks); after each IDCT pass a band swaps one
pixel row with each neighbour over RCCL (torch.distributed send/recv, ONE batched call for
the 12 planes of the step), which is the only data-path communication the algorithm has
(SURVEY.md section 8e).

Workloads (BASELINE.json configs): default = 8192^2 luma, --quality 3 (the metric) or 4
(configs[2]); --size 16384 (configs[3]); --quality 5/6 = 8192^2 4:2:0 YCbCr with
JOINT_YUV (+ UPSAMPLE_UV), niter 5 (configs[4], cross-component stages, colour bands).

One JSON line is printed by rank 0.  `roofline` is for the dominant kernel
(qs_smooth_set_kernel, one launch = the 12 planes of a step): achieved = algorithmic bytes
(256 B per block per launch: read + write of 64 int16) / mean launch time measured with HIP
events on the launch stream.  The kernel is FP32-VALU-bound, so `roofline_valu` gives the fraction of the
non-FMA FP32 vector peak, which is the roofline that actually binds -- `frac` against the nominal
78.6 T (2.0 cycles per instruction at 2.4 GHz), `frac_of_sustained` against the 68 T the chip sustains for a pure
stream of this kernel's term instructions (profiles/valu_sustained.json, tools/ubench_clock.hip).

Input (`--data jpeg`, the default): the synthetic image of SURVEY.md 8d ENCODED BY LIBJPEG (Pillow) and read back with
jpeg_read_coefficients (tools/jpeg_coefs.c) -- BASELINE.md section 3; `--data synth` (and the fall-back when Pillow or
the helper is missing): float32-DCT coefficients built on the GPU.  `config.workload` says which.

`verify_ok`: the planes of the last timed step are compared with each other on the GPU, and the last one with the
compiled reference (oracle/_ref/libqsref_none.so; the plain-C port when it did not travel) -- EVERY block at N = 1 up to
8192^2 (`verify_rows` 1024), 16 block rows at the top / middle / bottom for larger planes, the top rows and both sides of
every band edge for N > 1.  `cpu_baseline` times the compiled reference (AVX-512 and AVX2 builds, OpenMP) over a sweep
of thread counts on a bounded sample of the same workload on this box's host cores.

Extra legs of the luma workload (not part of `value`; `--no-extras` skips them):
  single_plane_ms / value_batch1   ONE plane per step instead of twelve: single-image latency (N = 1) and single-image
                                   strong scaling (N > 1: the RCCL halo exchange is paid per plane)
  scaling_emulation                N = 1: one rank's share of a SINGLE-image run on 2 / 4 / 8 GPUs emulated on this GPU (the
                                   middle 1/N band of one plane, halo rows as device copies): ms per step, pass-B launch time, speed-up
                                   -- also with 25 / 50 / 100 us of injected exchange latency, and for the communication-avoiding
                                   schedule (niter extra block rows per cut side, no exchange)
  deep_halo_schedule               N > 1: the single image on the communication-avoiding schedule, timed like value_batch1; its owned
                                   rows must equal the exchange schedule's
  edge_first_schedule              N > 1: the single image on the latency-hiding schedule (edge rows + exchange on a side stream while the
                                   interior rows run), timed like value_batch1; same result required
  smooth_input                     N = 1: the same workload on the smooth variant of the image (periods x10, no noise),
                                   where the wave-uniform need_refresh skip applies (LABNOTES.md 4.2c)
  product_route                    the PRODUCT's own multi-GPU route over the same N devices -- qs_hip_do_quantsmooth_sharded
                                   (csrc/qs_shard.cpp: one process, peer copies), host arrays in and out, run as a child
                                   process of rank 0 while the other ranks wait on a host-side (gloo) barrier
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
import jpegqs_pkg  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TFLOPS = 78.6        # 157.3 TF packed-FMA spec / 2: separate mul/add, no FMA allowed
FLOP_PER_BLOCK_ITER = {0: 75e3, 1: 130e3}   # q3 / q4 (DIAGONALS)
ALGO_BYTES_PER_BLOCK_ITER = 256
RESIDENT_BUDGET = 64 << 30     # HBM the resident input planes of all steps may take


KERNEL_SOURCES = ("qs_kernels.hip", "qs_smooth_kernel.inc", "qs_smooth_dp_kernel.inc", "qs_devfn.h", "qs_device.h", "qs_launch.h",
                  "strip_asm_nops.py", "build_stripped.sh")


def kernel_sources_sha1():
    """what the device code of the recovery kernels is built from (jpeg-quantsmooth_amd/csrc): profiles/pmc_traffic.json
    records this hash next to the PMC counters it holds, so a `roofline.traffic` measured on OTHER kernel sources shows in
    the bench line (`traffic_kernel_hash.match` false) instead of passing as current"""
    import hashlib
    h = hashlib.sha1()
    for name in KERNEL_SOURCES:
        f = ROOT / "jpeg-quantsmooth_amd" / "csrc" / name
        h.update(name.encode() + b"\0" + (f.read_bytes() if f.exists() else b"<missing>") + b"\0")
    return h.hexdigest()[:16]


def _sustained(achieved_tf):
    """the same fraction against what the chip SUSTAINS for a pure stream of this kernel's term instructions
    (profiles/valu_sustained.json, measured with tools/ubench_clock.hip): the shader clock under a full VALU load
    is ~2.2 GHz, not the 2.4 GHz the nominal peak assumes, and a SIMD issues one wave-instruction per 2.11 cycles"""
    try:
        j = json.loads((ROOT / "profiles" / "valu_sustained.json").read_text())
        peak = float(j["lane_instr_per_s"]) / 1e12
        return {"sustained_peak": peak, "frac_of_sustained": achieved_tf / peak, "sustained_source": j["_source"]}
    except Exception:
        return {}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=8192, help="the image is size x size pixels")
    ap.add_argument("--quality", type=int, default=3, choices=(3, 4, 5, 6),
                    help="jpegqs --quality: 3/4 = luma plane (the metric), 5/6 = 4:2:0 YCbCr with the cross-component stages")
    ap.add_argument("--niter", type=int, default=None, help="default 3 (quality 3/4), 5 (quality 5/6: BASELINE configs[4])")
    ap.add_argument("--batch", type=int, default=0, help="planes per step (0 = 12, fewer if the resident inputs would exceed 64 GiB)")
    ap.add_argument("--jpeg-quality", type=int, default=50, help="JPEG quality of the synthetic input")
    ap.add_argument("--weak", action="store_true", help="give every rank a full size x size plane")
    ap.add_argument("--overlap", action="store_true",
                    help="N > 1: interior rows on the main stream while halo exchange + edge rows run on a side stream "
                         "(measured slower on MI355X than the default in-order schedule, see LABNOTES.md section 8)")
    ap.add_argument("--no-overlap", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline sweep")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the last timed step")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra legs of the luma workload (single-plane latency, smooth input, the product's own multi-GPU route)")
    ap.add_argument("--data", default="jpeg", choices=("jpeg", "synth"),
                    help="jpeg (default): the synthetic image encoded by libjpeg and read back with jpeg_read_coefficients "
                         "(falls back to synth when Pillow or tools/jpeg_coefs is missing); synth: float32-DCT coefficients built on the GPU")
    ap.add_argument("--input", default="survey", choices=("survey", "smooth"),
                    help="synthetic image: the SURVEY.md 8d formula (default) or its noise-free, ten times smoother variant")
    ap.add_argument("--verify", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"), help="torch.distributed backend (gloo: functional test of the sharded path)")
    ap.add_argument("--single-device", action="store_true", help="all ranks use cuda:0 (functional test of the sharded path on a 1-GPU box, with --backend gloo)")
    ap.add_argument("--edge-first", action="store_true",
                    help="N > 1: also time the single image on the latency-hiding band schedule (bands.run_band_edge_first): edge rows + "
                         "exchange on a side stream, interior rows on the main stream")
    ap.add_argument("--print-kernel-hash", action="store_true", help="print kernel_sources_sha1() (what profiles/pmc_traffic.json is tied to) and exit")
    a = ap.parse_args()
    if a.print_kernel_hash:
        print(kernel_sources_sha1())
        sys.exit(0)
    if a.niter is None:
        a.niter = 5 if a.quality >= 5 else 3
    return a


def _synth_pixels_gpu(torch, w, h, dev, variant=0, seed=1234, squeeze=False, smooth=False):
    """the synthetic image of SURVEY.md 8d (formula of synth.py) as a float tensor of integer pixel values 0..255"""
    g = torch.Generator(device=dev); g.manual_seed(seed + 7919 * variant)
    x = torch.arange(w, device=dev, dtype=torch.float32)[None, :]
    y = torch.arange(h, device=dev, dtype=torch.float32)[:, None]
    # smooth: the same formula with every period ten times longer and no sensor noise -- large flat / gently shaded
    # regions, where the reference's need_refresh skips most refresh IDCTs (quantsmooth.h:1407-1409)
    k = 10 if smooth else 1
    px, py = (17.0 + 5 * variant) * k, (23.0 + 3 * variant) * k
    img = 128.0 + 60.0 * torch.sin(x / px) + 50.0 * torch.cos(y / py)
    checker = ((torch.arange(w, device=dev) // ((37 + 4 * variant) * k))[None, :] + (torch.arange(h, device=dev) // ((29 + 2 * variant) * k))[:, None]) & 1
    img = img + 40.0 * (checker.float() - 0.5)
    if not smooth:
        img = img + torch.randn(h, w, device=dev, generator=g) * 6.0
    y0, x0 = h // 5, w // 3
    img[y0:y0 + h // 7, x0:x0 + w // 4] *= 0.45
    img = img.round().clamp(0, 255)
    if squeeze:   # chroma is smoother and nearer mid-grey in natural images (synth.synth_ycc)
        img = (128 + torch.div(img - 128, 3, rounding_mode="floor")).clamp(0, 255)
    return img


def _synth_plane_gpu(torch, pkg, w, h, quant, dev, variant=0, seed=1234, squeeze=False, smooth=False):
    """int16 [h/8, w/8, 64] quantised coefficients of the synthetic plane, float32 DCT via two small matmuls on the
    GPU (the fall-back input when the libjpeg route below is not available, and what the tests use)"""
    synth = pkg.synth
    img = _synth_pixels_gpu(torch, w, h, dev, variant, seed, squeeze, smooth) - 128.0
    d = torch.from_numpy(synth._dct_matrix().astype(np.float32)).to(dev)
    blk = img.reshape(h // 8, 8, w // 8, 8).permute(0, 2, 1, 3)
    c = d @ blk @ d.T
    q = torch.from_numpy(quant.astype(np.float32)).to(dev).reshape(8, 8)
    return torch.round(c / q).to(torch.int16).reshape(h // 8, w // 8, 64).contiguous()


JPEG_COEFS = ROOT / "tools" / "jpeg_coefs"     # tools/jpeg_coefs.c, built by __graft_entry__.build()


def jpeg_input(torch, size, jpeg_quality, dev, colour=False, smooth=False):
    """BASELINE.md section 3's input: the synthetic image ENCODED BY LIBJPEG (Pillow: libjpeg-turbo, standard IJG tables at
    `jpeg_quality`, baseline Huffman, integer DCT) and read back with jpeg_read_coefficients() -- what the jpegqs CLI hands to
    do_quantsmooth.  -> ([coef tensors on the device], [quant tables]) or None when Pillow / the helper is missing."""
    import subprocess
    import tempfile
    try:
        from PIL import Image
    except ImportError:
        return None
    if not JPEG_COEFS.exists():
        return None
    Image.MAX_IMAGE_PIXELS = None
    planes = [_synth_pixels_gpu(torch, size, size, dev, variant=v, squeeze=v > 0, smooth=smooth).to(torch.uint8).cpu().numpy()
              for v in (range(3) if colour else range(1))]
    with tempfile.TemporaryDirectory() as tmp:
        jpg, raw = Path(tmp) / "in.jpg", Path(tmp) / "in.bin"
        if colour:
            Image.merge("YCbCr", [Image.fromarray(p, "L") for p in planes]).save(jpg, quality=jpeg_quality, subsampling=2)
        else:
            Image.fromarray(planes[0], "L").save(jpg, quality=jpeg_quality)
        if subprocess.run([str(JPEG_COEFS), str(jpg), str(raw)]).returncode != 0:
            return None
        b = raw.read_bytes()
    hdr = np.frombuffer(b[:20], np.int32)
    if hdr[0] != 0x51534a43:
        return None
    ncomp, off = int(hdr[1]), 20
    geo, quants = [], []
    for _ in range(ncomp):
        geo.append(np.frombuffer(b[off:off + 20], np.int32)); off += 20
        quants.append(np.frombuffer(b[off:off + 128], np.uint16).copy()); off += 128
    coefs = []
    for g in geo:
        n = int(g[0]) * int(g[1]) * 64
        coefs.append(torch.from_numpy(np.frombuffer(b[off:off + 2 * n], np.int16).reshape(int(g[1]), int(g[0]), 64).copy()).to(dev))
        off += 2 * n
    return coefs, quants


def synth_input_gpu(torch, pkg, size, jpeg_quality, dev, smooth=False):
    """luma workload: (coef int16 [hblk, wblk, 64] on the device, quant uint16[64])"""
    quant = pkg.synth.quality_table(pkg.synth.STD_LUMA, jpeg_quality)
    return _synth_plane_gpu(torch, pkg, size, size, quant, dev, smooth=smooth), quant


def synth_colour_gpu(torch, pkg, size, jpeg_quality, dev):
    """4:2:0 YCbCr workload: ([Y, Cb, Cr] coefficient tensors, [qY, qC, qC])"""
    qy = pkg.synth.quality_table(pkg.synth.STD_LUMA, jpeg_quality)
    qc = pkg.synth.quality_table(pkg.synth.STD_CHROMA, jpeg_quality)
    y = _synth_plane_gpu(torch, pkg, size, size, qy, dev)
    cb = _synth_plane_gpu(torch, pkg, size // 2, size // 2, qc, dev, variant=1, squeeze=True)
    cr = _synth_plane_gpu(torch, pkg, size // 2, size // 2, qc, dev, variant=2, squeeze=True)
    return [y, cb, cr], [qy, qc, qc.copy()]


# ---------------------------------------------------------------------------
# CPU baseline: the compiled reference on this box's host cores

def _host_info():
    info = {"os_cpu_count": os.cpu_count()}
    try:
        info["affinity_cpus"] = len(os.sched_getaffinity(0))
    except AttributeError:
        info["affinity_cpus"] = os.cpu_count()
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            info["cgroup_cpu_max"] = Path(p).read_text().strip()
            break
        except OSError:
            pass
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                info["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return info


def cpu_baseline(args, run_sample, total_units, unit_name, describe):
    """Sweep thread counts for the AVX-512 and AVX2 builds of the UNMODIFIED reference
    (oracle/_ref, OpenMP); the scalar oracle port only when no reference build travels with
    the tree.  run_sample(impl, threads, frac) runs `frac` of the workload and returns the
    number of blocks it processed.  About args.cpu_seconds of CPU work in total."""
    from oracle import oracle as om
    host = _host_info()
    usable = max(1, int(host.get("affinity_cpus") or 1))
    cg = str(host.get("cgroup_cpu_max", "max")).split()
    if len(cg) == 2 and cg[0].isdigit() and int(cg[1]) > 0:       # cgroup v2 "quota period"
        usable = max(1, min(usable, int(cg[0]) // int(cg[1])))
    variants = [v for v in ("avx512", "avx2") if om.have_ref(v) and
                all(om.cpu_has(f) for f in ({"avx512": ("avx512f", "avx512bw", "avx512dq", "fma"), "avx2": ("avx2", "fma")}[v]))]
    impls = [(f"reference {v}+openmp", "reference", om.Reference(v)) for v in variants]
    if not impls:
        impls = [("oracle port (scalar C + openmp)", "port", om.Oracle())]
    threads = sorted({t for t in (1, 8, 32, 64, 128, usable) if t <= usable})
    budget = args.cpu_seconds / max(1, len(impls) * len(threads))
    sweep, best, one_thread = {}, None, None
    for name, kind, impl in impls:
        sweep[name] = {}
        rate_guess = 0.08e6                                        # blocks/s of one thread (AVX2/AVX-512 class)
        for t in threads:
            # a sample sized for about a third of this point's budget, then best of two runs
            frac = min(1.0, max(1.0 / 512, rate_guess * t ** 0.85 * budget / 3 / total_units))
            runs = []
            t_point = time.time()
            while len(runs) < 2 or (time.time() - t_point < budget * 0.5 and len(runs) < 4):
                t0 = time.time()
                n = run_sample(impl, t, frac)
                runs.append(n / (time.time() - t0))
            rate = max(runs)
            sweep[name][str(t)] = round(rate)
            if t == 1:
                rate_guess = rate
                if one_thread is None or rate > one_thread[0]:
                    one_thread = (rate, name)
            if best is None or rate > best[0]:
                best = (rate, name, kind, t, frac)
    rate, name, kind, t, frac = best
    return {"value": rate, "unit": f"{unit_name}/s", "cores": t, "kind": kind,
            "sample": f"{describe(frac)}, q={args.quality} niter={args.niter}, {name}, {t} OpenMP threads "
                      f"(best point of the sweep; samples sized for ~{args.cpu_seconds:.0f} s of CPU work in total)",
            "one_thread": {"value": one_thread[0], "impl": one_thread[1]} if one_thread else None,
            "sweep_blocks_per_s": sweep, "host": host, "usable_cpus": usable}


# ---------------------------------------------------------------------------

def _self_launch(args):
    """`python bench.py --gpus N` started plainly (no launcher in the environment) becomes the N-rank job itself: the
    process is REPLACED by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`
    (one rank per GPU over RCCL; rendezvous on 127.0.0.1 and a free port), so that one command is the whole benchmark,
    like the reference's one call that fans out over OpenMP threads (quantsmooth.h:2587-2640)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")                 # (what torchrun would set, without its warning)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    print("bench.py: --gpus %d without a launcher: re-executing as `%s`" % (args.gpus, " ".join(cmd[1:])), file=sys.stderr, flush=True)
    os.execve(sys.executable, cmd, env)


def _needs_self_launch(args, environ):
    return args.gpus > 1 and "WORLD_SIZE" not in environ and "RANK" not in environ


def main():
    args = parse_args()
    if _needs_self_launch(args, os.environ):
        _self_launch(args)           # does not return
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world            # started by a launcher: its world size is the truth
    if args.single_device:
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:
        print(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} HIP device(s) visible "
              f"(--single-device --backend gloo runs all ranks on one GPU as a functional test)", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)                      # brings RCCL up (or fails) before anything is timed
                torch.cuda.synchronize()
                assert int(probe.item()) == world
            except Exception as e:                          # noqa: BLE001
                # A host-staged gloo number must never look like an RCCL result: the run was asked
                # for RCCL over xGMI, so it ends here, non-zero, with the reason -- `--backend gloo` is the explicit way to
                # time (or functionally test) the host-staged transport.
                print(f"bench.py rank {rank}: RCCL did not come up ({type(e).__name__}: {str(e)[:300]}); no result is printed. "
                      f"Re-run with --backend gloo to use the host-staged transport on purpose.", file=sys.stderr, flush=True)
                try:
                    if dist.is_initialized():
                        dist.destroy_process_group()
                finally:
                    sys.exit(4)
        if args.backend != "nccl":
            dist.init_process_group("gloo", rank=rank, world_size=world)
    # a host-only barrier (an RCCL barrier keeps a kernel spinning on every waiting GPU): used while rank 0 runs a leg alone
    host_group = dist.new_group(backend="gloo") if (world > 1 and args.backend == "nccl") else None

    pkg = jpegqs_pkg.load()
    hip = pkg.HipQS()          # raises if the HIP library is missing: no fallback
    from jpeg_quantsmooth_amd import bands
    flags = pkg.flags_for_quality(args.quality)
    colour = args.quality >= 5
    size = args.size
    sharded = world > 1 and not args.weak
    verify = not args.no_verify
    nsteps = args.steps + args.warmup
    ctx = dict(args=args, torch=torch, dist=dist, pkg=pkg, hip=hip, bands=bands, flags=flags, size=size, world=world,
               rank=rank, dev=dev, sharded=sharded, verify=verify, nsteps=nsteps, host_group=host_group)
    run = run_colour if colour else run_luma
    res = run(ctx)

    if rank == 0:
        value = res["total_blocks"] * args.steps / res["elapsed"]
        kern_ms = res["kern_ms"]
        if kern_ms is None:   # sharded run: no per-kernel events; derive from the step time (comm included)
            kern_ms = res["elapsed"] / args.steps / res["launches_per_step"] * 1e3
        kblocks = res["kernel_blocks"]
        achieved_gbs = kblocks * ALGO_BYTES_PER_BLOCK_ITER / (kern_ms * 1e-3) / 1e9
        achieved_tf = kblocks * FLOP_PER_BLOCK_ITER[flags & 1] / (kern_ms * 1e-3) / 1e12
        # `traffic` is NOT measured by this run: it is the PMC figure (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes with the
        # gfx950 corrections of MI355X_MICROARCH.md) of an earlier profiled run of the same command, kept in
        # profiles/pmc_traffic.json; `traffic_measured_in` names the profile folder it came from, so a stale figure is visible.
        traffic, traffic_src, traffic_hash = None, None, None
        pmc = ROOT / "profiles" / "pmc_traffic.json"
        if pmc.exists() and not colour and world == 1:
            try:
                j = json.loads(pmc.read_text())
                tree = kernel_sources_sha1()
                traffic_hash = {"measured_on": j.get("kernel_sources_sha1"), "this_tree": tree, "match": j.get("kernel_sources_sha1") == tree}
                ppl = res.get("planes_per_launch", 1)
                det = j.get(f"set{ppl}_detail", {}).get(f"q{args.quality}", {}) if ppl > 1 else {}
                traffic = j.get(f"q{args.quality}_{size}_set{ppl}") if ppl > 1 else None
                traffic_src = det.get("measured_in") or j.get("_source")
                if traffic is None:
                    traffic = j.get(f"q{args.quality}_{size}")
                    traffic_src = j.get("_source")
                    if traffic is not None and ppl > 1:           # only single-plane counters on file
                        traffic *= ppl
                        traffic_src = f"{ppl} x the per-plane figure of " + str(traffic_src)
            except Exception:
                traffic = None
        out = {
            "metric": "8x8 blocks/s at q=%d niter=%d" % (args.quality, args.niter),
            "value": value, "unit": "blocks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["elapsed"] / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",   # (a synthetic IMAGE; see config.workload for how it became coefficients)
            "mpixels_per_s": value * 64 / 1e6,
            "timed_region_s": res["elapsed"],
            "config": {"workload": res["workload"] +
                       f", jpegqs --quality {args.quality} (flags={flags}) --niter {args.niter}; input = the synthetic image of SURVEY.md 8d"
                       f"{' (smooth variant: periods x10, no noise)' if args.input == 'smooth' else ''}, "
                       + (f"ENCODED BY LIBJPEG (Pillow / libjpeg-turbo, standard tables at quality {args.jpeg_quality}, baseline, integer DCT) and "
                          f"read back with jpeg_read_coefficients (tools/jpeg_coefs.c), as BASELINE.md section 3 asks"
                          if ctx.get("data_note") == "libjpeg" else
                          f"as quantised float32-DCT coefficients at JPEG quality {args.jpeg_quality} built on the GPU (not a libjpeg-encoded "
                          f"file: Pillow or tools/jpeg_coefs missing, or --data synth)")
                       + "; the CPU baseline consumes the same arrays",
                       "planes_per_step": res["batch"],
                       "sharding": "none" if world == 1 else f"{world} block-row bands, 1-pixel-row halo over "
                                                               f"{'RCCL' if args.backend == 'nccl' else 'gloo (host-staged)'} per iteration, "
                                                               + ("exchange overlapped with the interior rows" if args.overlap
                                                                  else "the planes of a step advance together as a plane set: one launch per "
                                                                       "pass and one packed send + receive per neighbour and iteration, "
                                                                       "in stream order"),
                       "rccl_ranks": world if (world > 1 and args.backend == "nccl") else 0,
                       "blocks_per_gpu": res["blocks_per_gpu"]},
            "roofline": {"bound": "hbm", "kernel": res["kernel"], "achieved": achieved_gbs,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_measured_in": traffic_src,
                         "traffic_kind": "PMC counters of an earlier profiled run of this command (profiles/pmc_traffic.json); not re-measured by this run"
                                         + ("" if (traffic_hash or {}).get("match") else " -- STALE: taken on other kernel sources than this tree's (traffic_kernel_hash)"),
                         "traffic_kernel_hash": traffic_hash,
                         "kernel_ms": kern_ms,
                         "kernel_launches_timed": res["kernel_launches"],
                         "algorithmic_bytes_per_launch": kblocks * ALGO_BYTES_PER_BLOCK_ITER,
                         "planes_per_launch": res.get("planes_per_launch", 1),
                         "note": "kernel is FP32-VALU-bound (~290 flop/B); see roofline_valu"},
            "roofline_valu": {"bound": "fp32-valu (separate mul/add, FMA forbidden by bit-exactness)",
                              "achieved": achieved_tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": achieved_tf / VALU_PEAK_TFLOPS,
                              "flop_per_block_iter": FLOP_PER_BLOCK_ITER[flags & 1],
                              **_sustained(achieved_tf)},
        }
        for k in ("single_plane_ms", "value_batch1", "planes_identical", "smooth_input", "scaling_emulation", "deep_halo_schedule", "edge_first_schedule", "product_route", "verify_against"):
            if res.get(k) is not None:
                out[k] = res[k]
        for k in ("verify_ok", "verify_rows", "verify_detail", "verify_band_edges_ok"):
            if res.get(k) is not None:
                out[k] = res[k]
        if res.get("cpu_baseline"):
            out["cpu_baseline"] = res["cpu_baseline"]
            out["speedup_vs_cpu_baseline"] = value / res["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _fence(torch, dist, world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()


def _max_over_ranks(torch, dist, world, elapsed, dev, backend):
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def _batch_size(args, nsteps, bytes_per_plane):
    want = args.batch if args.batch > 0 else 12
    return max(1, min(want, RESIDENT_BUDGET // max(1, nsteps * bytes_per_plane)))


def _device_delay(torch, stream):
    """-> f(us): queue a device-side wait of about `us` microseconds on the stream (a spinning kernel, torch.cuda._sleep,
    calibrated here with HIP events), or None when it cannot be had.  Stands in for the latency of a halo exchange: the
    receiving stream is idle until the neighbour's row has arrived."""
    sleep = getattr(torch.cuda, "_sleep", None)
    if sleep is None:
        return None
    try:
        sleep(1000); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 2_000_000
        e0.record(stream); sleep(n); e1.record(stream); torch.cuda.synchronize()
        per_us = n / (e0.elapsed_time(e1) * 1e3)
        if not per_us > 0:
            return None
    except Exception:  # noqa: BLE001
        return None
    return lambda us: sleep(int(us * per_us)) if us > 0 else None


def _scaling_emulation(torch, hip, bands, full, quant, flags, niter, dev, hblk_total, steps=8):
    """ms per single-image step of the middle 1/N band for N = 1, 2, 4, 8 (batch = 1: one plane, one launch per pass),
    the mean pass-B launch time by HIP events, and the implied speed-up over N = 1 -- for the per-iteration halo
    exchange with an injected exchange latency of 0 / 25 / 50 / 100 us, and for the communication-avoiding schedule
    (niter extra block rows per cut side, no exchange: bands.deep_band_rows)"""
    delays = (0, 25, 50, 100)
    out = {"what": "one rank's share of a single-image N-GPU step on THIS GPU: middle 1/N band of one plane.  exchange schedule: "
                   "pass A once, niter x {2 halo rows in + out as device copies, pass B}, each exchange followed by a device-side "
                   "wait of 0 / 25 / 50 / 100 us (stand-in for the interconnect latency; speedup_vs_1 = no wait).  deep_halo: "
                   "the communication-avoiding schedule, niter extra block rows per cut side, no exchange at all",
           "ms_per_step": {}, "pass_b_us": {}, "speedup_vs_1": {}, "exchange_latency_us": {}, "deep_halo": {"ms_per_step": {}, "speedup_vs_1": {}, "rows": {}}}
    stream = torch.cuda.current_stream()
    delay = _device_delay(torch, stream)

    def time_band(src, topo, wait_us, exchange=True):
        work = [src.clone() for _ in range(steps + 2)]
        eng = bands.HipBandEngine(hip, torch, work[0], quant, flags, luma=1, device=dev)
        h = eng.hblk * 8
        is_band = exchange and topo.world > 1

        def fake_exchange():
            if is_band:                                  # same bytes as the real exchange, both directions
                eng.row(-1).copy_(eng.row(0)); eng.row(h).copy_(eng.row(h - 1))
                if wait_us and delay:
                    delay(wait_us)
        pairs, pend = [], []

        def mark(which):
            ev = torch.cuda.Event(enable_timing=True); ev.record(stream)
            if which == 0:
                pend.append(ev)
            else:
                pairs.append((pend.pop(), ev))
        for i, p in enumerate(work):
            if i == 2:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.rebind(p)
            bands.run_bands_batched_sets(hip, [eng], topo, niter, fake_exchange, mark=mark if i >= 2 else None)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        return ms, float(np.mean([a.elapsed_time(b) for a, b in pairs])) * 1e3
    side = torch.cuda.Stream(device=dev)

    def time_edge_first(src, topo, wait_us):
        """the latency-hiding schedule (bands.run_band_edge_first): edge block rows + exchange on a side stream, interior
        rows on the main stream; the exchange is two device copies + the injected wait, on the side stream"""
        work = [src.clone() for _ in range(steps + 2)]
        eng = bands.HipBandEngine(hip, torch, work[0], quant, flags, luma=1, device=dev)
        h = eng.hblk * 8

        def fake_exchange():
            eng.row(-1).copy_(eng.row(0)); eng.row(h).copy_(eng.row(h - 1))
            if wait_us and delay:
                delay(wait_us)
        for i, p in enumerate(work):
            if i == 2:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.rebind(p)
            bands.run_band_edge_first(hip, eng, topo, niter, fake_exchange, stream, side, torch)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    out["edge_first"] = {"what": "bands.run_band_edge_first: pass B of the band's first and last block row (one small launch) and the "
                                 "halo exchange on a side stream, the interior rows on the main stream -- the exchange hides behind them",
                         "ms_per_step": {}, "speedup_vs_1": {}}
    for n in (1, 2, 4, 8):
        if n == 1:
            topo = bands.BandTopology(0, 1, 0, hblk_total)
        else:
            r = n // 2
            r0, r1 = bands.band_rows(hblk_total, n, r)
            topo = bands.BandTopology(r, n, r0, r1)
        src = full[topo.r0:topo.r1].contiguous()
        ms, pb = time_band(src, topo, 0)
        out["ms_per_step"][str(n)] = ms
        out["pass_b_us"][str(n)] = pb
        if n > 1:
            if delay:
                out["exchange_latency_us"][str(n)] = {str(d): out["ms_per_step"]["1"] / time_band(src, topo, d)[0] for d in delays[1:]}
            out["edge_first"]["ms_per_step"][str(n)] = {str(d): time_edge_first(src, topo, d) for d in (delays if delay else delays[:1])}
            r0, r1, e0, e1 = bands.deep_band_rows(hblk_total, n, n // 2, niter)
            dms, _ = time_band(full[e0:e1].contiguous(), bands.BandTopology(0, 1, e0, e1), 0, exchange=False)
            out["deep_halo"]["ms_per_step"][str(n)] = dms
            out["deep_halo"]["rows"][str(n)] = [r1 - r0, e1 - e0]
        del src
    for n in ("2", "4", "8"):
        out["speedup_vs_1"][n] = out["ms_per_step"]["1"] / out["ms_per_step"][n]
        out["deep_halo"]["speedup_vs_1"][n] = out["ms_per_step"]["1"] / out["deep_halo"]["ms_per_step"][n]
        out["edge_first"]["speedup_vs_1"][n] = {d: out["ms_per_step"]["1"] / ms for d, ms in out["edge_first"]["ms_per_step"][n].items()}
    if not delay:
        out["exchange_latency_us"] = None
    return out


# ---------------------------------------------------------------------------
# luma plane (BASELINE configs[1..3]: the metric)

def run_luma(c):
    args, torch, dist, pkg, hip, bands = c["args"], c["torch"], c["dist"], c["pkg"], c["hip"], c["bands"]
    flags, size, world, rank, dev = c["flags"], c["size"], c["world"], c["rank"], c["dev"]
    hblk_total, wblk = size // 8, size // 8
    if c["sharded"]:
        r0, r1 = bands.band_rows(hblk_total, world, rank)
        topo = bands.BandTopology(rank, world, r0, r1)
    else:
        topo = bands.BandTopology(0, 1, 0, hblk_total)       # a whole plane per rank
    r0, r1, hblk = topo.r0, topo.r1, topo.hblk
    total_blocks_plane = (hblk_total * wblk) * (world if args.weak else 1)

    got = jpeg_input(torch, size, args.jpeg_quality, dev, smooth=args.input == "smooth") if args.data == "jpeg" else None
    if got is not None:
        full, quant = got[0][0], got[1][0]
        c["data_note"] = "libjpeg"
    else:
        full, quant = synth_input_gpu(torch, pkg, size, args.jpeg_quality, dev, smooth=args.input == "smooth")
        c["data_note"] = "synth"
    pristine = full[r0:r1].contiguous()
    nsteps = c["nsteps"]
    batch = _batch_size(args, nsteps, pristine.numel() * 2)
    edge_checks = []
    if c["verify"] and c["sharded"]:
        # the two block rows on our side of each band edge, checked against the oracle run on a
        # crop around the edge (a block after n iterations depends only on blocks within n of it)
        m = args.niter + 2
        for edge, mine in [e for e in ((r1, (r1 - 2, r1)) if topo.down is not None else None,
                                       (r0, (r0, r0 + 2)) if topo.up is not None else None) if e]:
            lo, hi = max(0, edge - 2 - m), min(hblk_total, edge + 2 + m)
            edge_checks.append((lo, mine, full[lo:hi].contiguous().cpu().numpy()))
    keep_full = full if (rank == 0 and (c["verify"] or not args.no_cpu_baseline)) else None
    # N > 1 extra leg: the communication-avoiding schedule (niter extra block rows per cut side, no exchange)
    deep_rows = bands.deep_band_rows(hblk_total, world, rank, args.niter) if (c["sharded"] and not args.no_extras) else None
    deep_src = full[deep_rows[2]:deep_rows[3]].contiguous() if deep_rows else None
    del full

    work = [[pristine.clone() for _ in range(batch)] for _ in range(nsteps)]   # resident inputs, one set per step
    # the pixel planes of the batch live in one [batch, plane_bytes] tensor: the halo rows of all of them
    # are then packed / unpacked with one strided copy each (bands.exchange_halo_packed)
    # (two of them: pass B writes the next iteration's pixel planes into the other one, fused pass A)
    planes2d = torch.zeros((batch, (hip.plane_bytes(wblk, hblk) + 255) & ~255), dtype=torch.uint8, device=dev)
    planes2d_b = torch.zeros_like(planes2d)
    eng = bands.HipBandEngine(hip, torch, work[0][0], quant, flags, luma=1, device=dev, plane=planes2d[0], plane2=planes2d_b[0])
    stream = torch.cuda.current_stream()
    ev_pairs = []
    is_band = topo.up is not None or topo.down is not None
    comm = eng.comm_scope() if (is_band and args.overlap) else None
    exch = bands.exchange_halo_dist if args.backend == "nccl" else bands.exchange_halo_dist_hostcopy
    # Default schedule at every N: the planes of a step advance together as ONE plane set -- one launch
    # per pass for all of them (a lone 1/8 band leaves the chip two-thirds idle; at N = 1 it saves the
    # twelve launch tails) and, for N > 1, ONE batched halo exchange per iteration (latency-bound: 2 rows
    # of 8 KB per plane).  --overlap keeps the older per-plane schedule.
    engs = [eng] + [bands.HipBandEngine(hip, torch, work[0][b], quant, flags, luma=1, device=dev, plane=planes2d[b], plane2=planes2d_b[b])
                    for b in range(1, batch)] if not args.overlap else None

    def exch_many(part, lo):
        # the planes the coming pass B reads: whichever of the two tensors the engines' `plane` points into right now
        cur = planes2d if part[0].plane.data_ptr() == planes2d[lo].data_ptr() else planes2d_b
        bands.exchange_halo_packed(hip, cur[lo:lo + len(part)], wblk, hblk, topo, dist, hostcopy=args.backend != "nccl")
    pending = []

    def mark(which):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        if which == 0:
            pending.append(ev)
        else:
            ev_pairs.append((pending.pop(), ev))

    def one_step_sharded(planes, timed=False):
        for e, p in zip(engs, planes):
            e.rebind(p)
        for lo in range(0, len(engs), 48):               # (a plane set holds up to 56 planes)
            part = engs[lo:lo + 48]
            bands.run_bands_batched_sets(hip, part, topo, args.niter,
                                         (lambda: exch_many(part, lo)) if is_band else (lambda: None),
                                         mark=mark if timed else None)

    def one_plane(coef, timed):
        """--overlap only: one plane at a time, bands.run_band_overlapped (interior rows on the main
        stream, halo exchange + edge rows on a side stream)"""
        eng.rebind(coef)
        if is_band:
            bands.run_band_overlapped(eng, topo, args.niter, lambda: exch(eng, topo, dist), comm=comm)
            return
        for it in range(args.niter):
            eng.idct(it == 0, topo.rep_top, topo.rep_bot)
            if timed:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            eng.smooth(it == args.niter - 1)
            if timed:
                e1.record(stream)
                ev_pairs.append((e0, e1))

    for i in range(args.warmup):
        if engs:
            one_step_sharded(work[i])
        else:
            for p in work[i]:
                one_plane(p, False)
    _fence(torch, dist, world)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if engs:
            one_step_sharded(work[args.warmup + i], timed=True)
        else:
            for bi, p in enumerate(work[args.warmup + i]):
                one_plane(p, bi == 0)                    # HIP events around the first plane's launches of every step
    _fence(torch, dist, world)
    elapsed = _max_over_ranks(torch, dist, world, time.perf_counter() - t0, dev, args.backend)
    assert not any(e.bad_coef() for e in (engs or [eng])), "range check tripped on synthetic input"
    last = work[-1][-1]                                   # the last plane of the last timed step
    # every plane of the last timed step must be the same result (they had the same input)
    planes_identical = all(bool(torch.equal(work[-1][0], p)) for p in work[-1][1:])

    # ---- extra legs (not part of `value`) ------------------------------------------------------------------
    single_plane_ms = value_batch1 = smooth_res = deep_res = edge_res = None
    if engs and not args.no_extras:
        # (1) ONE plane per step (batch = 1): the latency of a single image and, for N > 1, single-image strong scaling
        k1 = max(3, min(args.steps, 20))
        w1 = [pristine.clone() for _ in range(k1 + 2)]

        def one_single(p):
            engs[0].rebind(p)
            bands.run_bands_batched_sets(hip, engs[:1], topo, args.niter,
                                         (lambda: exch_many(engs[:1], 0)) if is_band else (lambda: None))
        for p in w1[:2]:
            one_single(p)
        _fence(torch, dist, world)
        t1 = time.perf_counter()
        for p in w1[2:]:
            one_single(p)
        _fence(torch, dist, world)
        e1 = _max_over_ranks(torch, dist, world, time.perf_counter() - t1, dev, args.backend)
        single_plane_ms = e1 / k1 * 1e3
        value_batch1 = total_blocks_plane / (e1 / k1)
        planes_identical = planes_identical and bool(torch.equal(w1[-1], last))
        del w1
        # (1b) N > 1: the same single image on the COMMUNICATION-AVOIDING schedule: every rank runs its band plus niter block
        # rows per cut side, nothing is exchanged during the iterations (csrc/qs_shard.cpp: qs_hip_set_shard_schedule(1));
        # its owned rows must equal what the exchange schedule produced
        def timed_single(run_one, inputs):
            """k1 steps of one image each, timed like value_batch1 -> (seconds, the last result tensor)"""
            for p in inputs[:2]:
                run_one(p)
            _fence(torch, dist, world)
            t = time.perf_counter()
            for p in inputs[2:]:
                run_one(p)
            _fence(torch, dist, world)
            return _max_over_ranks(torch, dist, world, time.perf_counter() - t, dev, args.backend), inputs[-1]

        def all_ranks_agree(ok):
            flag = torch.tensor([int(ok)], dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())

        def leg_deep():
            o0, o1, e0, e1 = deep_rows
            deng = bands.HipBandEngine(hip, torch, deep_src, quant, flags, luma=1, device=dev)
            dtopo = bands.BandTopology(0, 1, e0, e1)

            def one(p):
                deng.rebind(p)
                bands.run_bands_batched_sets(hip, [deng], dtopo, args.niter, lambda: None)
            sec, res = timed_single(one, [deep_src.clone() for _ in range(k1 + 2)])
            return {"single_plane_ms": sec / k1 * 1e3, "value_batch1": total_blocks_plane / (sec / k1),
                    "rows_owned_and_held": [o1 - o0, e1 - e0],
                    "equals_exchange_schedule": all_ranks_agree(torch.equal(res[o0 - e0:o1 - e0], last)),
                    "what": "one image per step, every rank holds niter extra block rows per cut side and exchanges nothing"}

        def leg_edge_first():
            eeng = bands.HipBandEngine(hip, torch, pristine, quant, flags, luma=1, device=dev)
            side = torch.cuda.Stream(device=dev)

            def one(p):
                eeng.rebind(p)
                bands.run_band_edge_first(hip, eeng, topo, args.niter, lambda: exch(eeng, topo, dist), stream, side, torch)
            sec, res = timed_single(one, [pristine.clone() for _ in range(k1 + 2)])
            return {"single_plane_ms": sec / k1 * 1e3, "value_batch1": total_blocks_plane / (sec / k1),
                    "equals_exchange_schedule": all_ranks_agree(torch.equal(res, last)),
                    "what": "one image per step; pass B of the band's first / last block row and the halo exchange on a side stream, "
                            "the interior rows on the main stream (bands.run_band_edge_first)"}
        # (an extra leg must not cost the headline line: a failure is recorded, the run goes on)
        if deep_src is not None and world > 1:
            try:
                deep_res = leg_deep()
            except Exception as ex:  # noqa: BLE001
                deep_res = {"error": repr(ex)[:300]}
        # (1c) N > 1, opt-in (--edge-first): the single image on the latency-hiding schedule.  It issues RCCL operations from a
        # second stream next to running kernels -- a pattern nothing in this repository has met real links with:
        # tools/first_contact.sh asks for it, the driver's plain command does not.
        if is_band and world > 1 and args.edge_first:
            try:
                edge_res = leg_edge_first()
            except Exception as ex:  # noqa: BLE001
                edge_res = {"error": repr(ex)[:300]}
        # (2) N = 1: the same workload on the smooth variant of the image (what the wave-uniform need_refresh skip is
        # worth on content that is not sensor noise; the headline input never lets a whole wave skip)
        if world == 1 and args.input == "survey":
            got_s = jpeg_input(torch, size, args.jpeg_quality, dev, smooth=True) if c["data_note"] == "libjpeg" else None
            sm = got_s[0][0] if got_s is not None else synth_input_gpu(torch, pkg, size, args.jpeg_quality, dev, smooth=True)[0]
            ks = max(2, min(args.steps, 5))
            ws = [[sm.clone() for _ in range(batch)] for _ in range(ks + 1)]
            one_step_sharded(ws[0])
            _fence(torch, dist, world)
            t2 = time.perf_counter()
            for st in ws[1:]:
                one_step_sharded(st)
            _fence(torch, dist, world)
            e2 = time.perf_counter() - t2
            smooth_res = {"value": total_blocks_plane * batch * ks / e2, "unit": "blocks/s", "ms_per_step": e2 / ks * 1e3,
                          "steps": ks, "input": "SURVEY.md 8d formula with every period x10 and no noise"}
            del ws, sm

    # (3) N = 1: what ONE rank of an N-GPU run of a SINGLE image does per step, emulated here -- the middle 1/N band of one
    # plane, niter x {pass A, halo rows, pass B}, the halo exchange replaced by device copies of the same rows.  Device
    # side only (no interconnect latency): an upper bound on single-image strong scaling, reported every round.
    scaling_emu = None
    if engs and not args.no_extras and world == 1 and hblk_total >= 64:
        scaling_emu = _scaling_emulation(torch, hip, bands, pristine, quant, flags, args.niter, dev, hblk_total)

    set_launch = engs is not None                         # the timed launches cover all planes of the step
    res = dict(elapsed=elapsed, batch=batch, total_blocks=total_blocks_plane * batch, blocks_per_gpu=hblk * wblk,
               kernel=(f"qs_smooth_set_kernel<{'true' if flags & 1 else 'false'}> (one launch = the {batch} planes of a step)"
                       if set_launch else f"qs_smooth_plane_kernel<{'true' if flags & 1 else 'false'}>"),
               kernel_blocks=hblk * wblk * (min(batch, 48) if set_launch else 1),
               planes_per_launch=min(batch, 48) if set_launch else 1,
               launches_per_step=args.niter * (-(-batch // 48) if set_launch else batch),
               kern_ms=float(np.mean([a.elapsed_time(b) for a, b in ev_pairs])) if ev_pairs else None,
               kernel_launches=len(ev_pairs),
               workload=f"{size}x{size} luma plane ({hblk_total * wblk} blocks)",
               planes_identical=planes_identical, single_plane_ms=single_plane_ms, value_batch1=value_batch1,
               smooth_input=smooth_res, scaling_emulation=scaling_emu, deep_halo_schedule=deep_res, edge_first_schedule=edge_res)
    if c["verify"] and c["sharded"]:
        from oracle.oracle import Oracle
        got = last.cpu().numpy()
        ok = True
        for lo, (a0, a1), crop in edge_checks:
            want = Oracle().do_quantsmooth([crop], [quant], flags, args.niter, threads=0)["coefs"][0]
            ok &= bool(np.array_equal(got[a0 - r0:a1 - r0], want[a0 - lo:a1 - lo]))
        flag = torch.tensor([int(ok)], dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        res["verify_band_edges_ok"] = bool(flag.item())
    if c["verify"]:
        # The checker: the compiled, unmodified reference (oracle/_ref, scalar build, OpenMP) when it travelled with the
        # tree, the plain-C port otherwise.  N = 1: EVERY block of the plane (the reference does 8192^2 in ~5 s on 16
        # cores); larger planes and N > 1 (rank 0 holds one band): 16 block rows at the top, middle, bottom / at the top,
        # the band edges were checked above.
        from oracle import oracle as om
        from oracle.oracle import RowSource, verify_bands
        if rank == 0:
            truth = om.Reference("none") if om.have_ref("none") else om.Oracle()
            res["verify_against"] = ("compiled reference (oracle/_ref/libqsref_none.so)" if om.have_ref("none")
                                     else "plain-C port (oracle/libqs_oracle.so)")
            if c["sharded"]:
                n = min(16, hblk)
                crop = keep_full[: min(hblk_total, n + args.niter + 1)].cpu().numpy()
                want = truth.do_quantsmooth([crop], [quant], flags, args.niter, threads=0)["coefs"][0][:n]
                bad = int((last[:n].cpu().numpy() != want).any(axis=2).sum())
                detail = [dict(where="top", row0=0, row1=n, bad_blocks=bad)]
            elif hblk_total * wblk <= (1 << 20) + 1:
                want = truth.do_quantsmooth([keep_full.cpu().numpy()], [quant], flags, args.niter, threads=0)["coefs"][0]
                bad = int((last.cpu().numpy() != want).any(axis=2).sum())
                detail = [dict(where="whole plane", row0=0, row1=hblk_total, bad_blocks=bad)]
            else:
                detail = verify_bands(truth, RowSource(keep_full), quant, flags, args.niter, RowSource(last), rows=16)
            res["verify_detail"] = detail
            res["verify_rows"] = sum(d["row1"] - d["row0"] for d in detail)
            res["verify_ok"] = all(d["bad_blocks"] == 0 for d in detail) and planes_identical
    # (3) the PRODUCT's own multi-GPU route (csrc/qs_shard.cpp behind qs_hip_do_quantsmooth_sharded: one process, peer
    # copies) over the same devices, host arrays in and out -- a child process of rank 0 while the other ranks wait on the
    # host; its JSON is merged as `product_route`
    if not args.no_extras and not args.weak and size <= 16384:
        hg = c.get("host_group")
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier(group=hg) if hg is not None else dist.barrier()
        if rank == 0:
            import subprocess
            devs = ",".join(str(d) for d in (range(world) if not args.single_device else [0] * world))
            cmd = [sys.executable, str(ROOT / "tools" / "bench_product_route.py"), "--devices", devs, "--size", str(size),
                   "--quality", str(args.quality), "--niter", str(args.niter), "--jpeg-quality", str(args.jpeg_quality)]
            try:
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                res["product_route"] = json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-400:]}
            except Exception as e:  # noqa: BLE001 -- the headline line must not depend on this leg
                res["product_route"] = {"error": repr(e)[:300]}
        if world > 1:
            dist.barrier(group=hg) if hg is not None else dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        host_plane = keep_full.cpu().numpy()

        def run_sample(impl, threads, frac):
            rows = max(8, min(hblk_total, int(round(hblk_total * frac))))
            crop = np.ascontiguousarray(host_plane[:rows])
            impl.do_quantsmooth([crop], [quant], flags, args.niter, threads=threads)
            return rows * wblk
        res["cpu_baseline"] = cpu_baseline(
            args, run_sample, hblk_total * wblk, "blocks",
            lambda f: f"the top {max(8, min(hblk_total, int(round(hblk_total * f))))} of {hblk_total} block rows of the same plane")
    return res


# ---------------------------------------------------------------------------
# 4:2:0 YCbCr with the cross-component stages (BASELINE configs[4])

def run_colour(c):
    args, torch, dist, pkg, hip, B = c["args"], c["torch"], c["dist"], c["pkg"], c["hip"], c["bands"]
    flags, size, world, rank, dev = c["flags"], c["size"], c["world"], c["rank"], c["dev"]
    hby, hbc, wby = size // 8, size // 16, size // 8
    got = jpeg_input(torch, size, args.jpeg_quality, dev, colour=True) if args.data == "jpeg" else None
    if got is not None:
        coefs, quants = got
        c["data_note"] = "libjpeg"
    else:
        coefs, quants = synth_colour_gpu(torch, pkg, size, args.jpeg_quality, dev)
        c["data_note"] = "synth"
    hsamp, vsamp = [2, 1, 1], [2, 1, 1]
    if c["sharded"]:
        y0, y1, c0, c1 = B.colour_band_split(hby, hbc, 2, world)[rank]
        topo = B.BandTopology(rank, world, c0, c1)
    else:
        y0, y1, c0, c1 = 0, hby, 0, hbc
        topo = B.BandTopology(0, 1, 0, hbc)
    mine = [coefs[0][y0:y1].contiguous(), coefs[1][c0:c1].contiguous(), coefs[2][c0:c1].contiguous()]
    nsteps = c["nsteps"]
    batch = _batch_size(args, nsteps, sum(t.numel() for t in mine) * 2 * 3)    # (+ the upsampled outputs)
    total_blocks_img = (hby * wby + 2 * hbc * (wby // 2)) * (world if args.weak else 1)
    small = size <= 1024                                    # small enough for the oracle to do the whole image
    keep = coefs if ((rank == 0 or (small and c["sharded"])) and (c["verify"] or not args.no_cpu_baseline)) else None
    del coefs
    work = [[[t.clone() for t in mine] for _ in range(batch)] for _ in range(nsteps)]
    stream = torch.cuda.current_stream()
    ev_pairs = []
    is_band = topo.up is not None or topo.down is not None

    xrows = B.exchange_rows_dist if args.backend == "nccl" else B.exchange_rows_dist_hostcopy

    def exchange(rows_list):
        if is_band:
            xrows(rows_list[0], topo, dist)
    # one band object (constants, planes, low-res luma), re-pointed at each resident image
    band = B.ColourBand(hip, torch, work[0][0], quants, hsamp, vsamp, (size, size), flags, args.niter, topo, dev)
    band.chroma_row0 = c0
    timing = [False]
    plain_next = band.eng[0].smooth_next

    def luma_smooth_next(*a, **kw):                         # HIP events around the luma recovery launches
        if not timing[0]:
            return plain_next(*a, **kw)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream); plain_next(*a, **kw); e1.record(stream)
        ev_pairs.append((e0, e1))
    band.eng[0].smooth_next = luma_smooth_next

    def one_image(cf, timed):
        for ci in range(3):
            band.eng[ci].rebind(cf[ci])
        timing[0] = timed
        B.run_colour_bands([band], exchange)

    for i in range(args.warmup):
        for cf in work[i]:
            one_image(cf, False)
    _fence(torch, dist, world)
    t0 = time.perf_counter()
    for i in range(args.steps):
        for bi, cf in enumerate(work[args.warmup + i]):
            one_image(cf, bi == 0 and not is_band)
    _fence(torch, dist, world)
    elapsed = _max_over_ranks(torch, dist, world, time.perf_counter() - t0, dev, args.backend)
    for e in band.eng:
        assert not e.bad_coef(), "range check tripped on synthetic input"
    res = dict(elapsed=elapsed, batch=batch, total_blocks=total_blocks_img * batch,
               blocks_per_gpu=sum(int(t.shape[0] * t.shape[1]) for t in mine),
               kernel=f"qs_smooth_plane_kernel<{'true' if flags & 1 else 'false'}> (luma plane)",
               kernel_blocks=(y1 - y0) * wby,
               kern_ms=float(np.mean([a.elapsed_time(b) for a, b in ev_pairs])) if ev_pairs else None,
               kernel_launches=len(ev_pairs), launches_per_step=args.niter * batch,
               workload=f"{size}x{size} 4:2:0 YCbCr image ({hby * wby} + 2 x {hbc * (wby // 2)} blocks)")
    if c["verify"] and c["sharded"] and small:
        # functional runs of the sharded path: every rank checks its whole band against the oracle's
        # result for the whole image
        from oracle.oracle import Oracle
        want = Oracle().do_quantsmooth([t.cpu().numpy() for t in keep], quants, flags, args.niter, threads=0,
                                       hsamp=hsamp, vsamp=vsamp, colorspace=3, image_size=(size, size))
        ok = bool(np.array_equal(band.eng[0].coef.cpu().numpy(), want["coefs"][0][y0:y1]))
        for ci in (1, 2):
            if want["up"]:
                ok &= bool(np.array_equal(band.up[ci - 1].cpu().numpy(), want["coefs"][ci][y0:y1]))
            else:
                ok &= bool(np.array_equal(band.eng[ci].coef.cpu().numpy(), want["coefs"][ci][c0:c1]))
        flag = torch.tensor([int(ok)], dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        res["verify_band_edges_ok"] = bool(flag.item())
    if (c["verify"] or not args.no_cpu_baseline) and rank == 0:
        # top crop: a block's result depends on blocks within about 2 * niter + 2 chroma block rows
        # (luma iterations feed the predictor through the low-res luma plane, then the chroma iterations)
        from oracle.oracle import Oracle
        rows_c = min(hbc, 8)
        crop_c = min(hbc, rows_c + 2 * args.niter + 3)
        crop = [keep[0][: crop_c * 2].cpu().numpy(), keep[1][:crop_c].cpu().numpy(), keep[2][:crop_c].cpu().numpy()]
        kw = dict(hsamp=hsamp, vsamp=vsamp, colorspace=3, image_size=(size, crop_c * 16))
        if c["verify"]:
            from oracle import oracle as om
            truth = om.Reference("none") if om.have_ref("none") else Oracle()
            res["verify_against"] = ("compiled reference (oracle/_ref/libqsref_none.so)" if om.have_ref("none")
                                     else "plain-C port (oracle/libqs_oracle.so)")
            want = truth.do_quantsmooth(crop, quants, flags, args.niter, threads=0, **kw)
            got = [band.eng[0].coef[: rows_c * 2].cpu().numpy()]
            for ci in (1, 2):
                got.append((band.up[ci - 1][: rows_c * 2] if want["up"] else band.eng[ci].coef[:rows_c]).cpu().numpy())
            detail = []
            for ci in range(3):
                w = want["coefs"][ci][: got[ci].shape[0]]
                detail.append(dict(where=f"top, component {ci}", row0=0, row1=int(got[ci].shape[0]),
                                   bad_blocks=int((got[ci] != w).any(axis=2).sum())))
            res["verify_detail"] = detail
            res["verify_rows"] = rows_c
            res["verify_ok"] = all(d["bad_blocks"] == 0 for d in detail)
        if world == 1 and not args.no_cpu_baseline:
            nblk = sum(int(a.shape[0] * a.shape[1]) for a in crop)

            def run_sample(impl, threads, frac):
                impl.do_quantsmooth(crop, quants, flags, args.niter, threads=threads, **kw)
                return nblk
            res["cpu_baseline"] = cpu_baseline(args, run_sample, nblk, "blocks",
                                               lambda f: f"the top {crop_c} chroma block rows of the same image ({nblk} blocks)")
    return res


if __name__ == "__main__":
    main()
