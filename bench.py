#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X jpeg-quantsmooth hot path.

Metric (BASELINE.json): 8x8 blocks/s at q=3 niter=3 on a synthetic 8192x8192 luma
plane, inputs resident in HBM.  A "step" is one complete do_quantsmooth pass over
one plane: niter x {IDCT-to-plane kernel, [halo exchange], recovery kernel}, final
clamp fused into the last recovery launch.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: the plane is split into N contiguous block-row bands (strong scaling, total
work fixed, as BASELINE.json's north_star asks); after each IDCT pass a band swaps
one pixel row with each neighbour over RCCL (torch.distributed send/recv), which is
the only data-path communication the algorithm has.

One JSON line is printed by rank 0.  `roofline` is for the dominant kernel
(qs_smooth_plane_kernel): achieved = algorithmic bytes (256 B per block per launch:
read + write of 64 int16) / mean launch time measured with HIP events on the launch
stream.  The kernel is FP32-VALU-bound, so `roofline_valu` gives the fraction of the
non-FMA FP32 vector peak, which is the roofline that actually binds.
`cpu_baseline` times the reference (oracle/_ref, AVX-512/AVX2 + OpenMP) -- or the
oracle port when the reference build is absent -- on a bounded sample of the same
workload on this box's host cores.
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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=8192, help="luma plane is size x size pixels")
    ap.add_argument("--quality", type=int, default=3, choices=(3, 4), help="jpegqs --quality (3 or 4)")
    ap.add_argument("--niter", type=int, default=3)
    ap.add_argument("--jpeg-quality", type=int, default=50, help="JPEG quality of the synthetic input")
    ap.add_argument("--weak", action="store_true", help="give every rank a full size x size plane")
    ap.add_argument("--overlap", action="store_true",
                    help="N > 1: interior rows on the main stream while halo exchange + edge rows run on a side stream "
                         "(measured slower on MI355X than the default in-order schedule, see DESIGN.md section 8)")
    ap.add_argument("--no-overlap", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="CPU baseline sample is NxN pixels")
    ap.add_argument("--verify", action="store_true", help="check rows against the oracle after the run (N > 1: rows around every band edge)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"), help="torch.distributed backend (gloo: functional test of the sharded path)")
    ap.add_argument("--single-device", action="store_true", help="all ranks use cuda:0 (functional test of the sharded path on a 1-GPU box, with --backend gloo)")
    return ap.parse_args()


def synth_input_gpu(torch, pkg, size, jpeg_quality, dev):
    """Quantised coefficient plane built on the GPU (same formula as synth.py,
    float32 DCT via two small matmuls): int16 [hblk, wblk, 64] + quant table."""
    synth = pkg.synth
    quant = synth.quality_table(synth.STD_LUMA, jpeg_quality)
    g = torch.Generator(device=dev); g.manual_seed(1234)
    x = torch.arange(size, device=dev, dtype=torch.float32)[None, :]
    y = torch.arange(size, device=dev, dtype=torch.float32)[:, None]
    img = 128.0 + 60.0 * torch.sin(x / 17.0) + 50.0 * torch.cos(y / 23.0)
    checker = ((torch.arange(size, device=dev) // 37)[None, :] + (torch.arange(size, device=dev) // 29)[:, None]) & 1
    img = img + 40.0 * (checker.float() - 0.5)
    img = img + torch.randn(size, size, device=dev, generator=g) * 6.0
    y0, x0 = size // 5, size // 3
    img[y0:y0 + size // 7, x0:x0 + size // 4] *= 0.45
    img = img.round().clamp(0, 255) - 128.0
    d = torch.from_numpy(synth._dct_matrix().astype(np.float32)).to(dev)
    blk = img.reshape(size // 8, 8, size // 8, 8).permute(0, 2, 1, 3)
    c = d @ blk @ d.T
    q = torch.from_numpy(quant.astype(np.float32)).to(dev).reshape(8, 8)
    c = torch.round(c / q).to(torch.int16).reshape(size // 8, size // 8, 64).contiguous()
    return c, quant


def cpu_baseline(pkg, args, coef_full, quant, flags):
    """Time the reference (oracle/_ref, best ISA this host runs, OpenMP on all
    cores) -- or the oracle port if the reference build is absent -- on a
    bounded sample: square crops of the workload, doubling until one run takes
    >= 2 s or the whole plane is used; ~10-30 s of CPU work in total."""
    from oracle import oracle as om
    cores = os.cpu_count() or 1
    variant = om.best_ref_variant()
    if variant:
        impl, kind, name = om.Reference(variant), "reference", f"reference {variant}+openmp"
    else:
        impl, kind, name = om.Oracle(), "port", "oracle port (scalar C + openmp)"
    full = coef_full.shape[0]
    n = min(max(args.cpu_sample // 8, 8), full)
    best = None
    t_all = time.time()
    while True:
        crop = np.ascontiguousarray(coef_full[:n, :n])
        runs = []
        for rep in range(3):
            t0 = time.time()
            impl.do_quantsmooth([crop], [quant], flags, args.niter, threads=0)
            runs.append(time.time() - t0)
            if time.time() - t_all > 25:
                break
        rate = n * n / min(runs)
        if best is None or rate > best[0]:
            best = (rate, n, min(runs), len(runs))
        if min(runs) >= 2.0 or n >= full or time.time() - t_all > 15:
            break
        n = min(n * 2, full)
    rate, n, secs, reps = best
    return {"value": rate, "unit": "blocks/s", "cores": cores, "kind": kind,
            "sample": f"{n * 8}x{n * 8} px crop of the workload ({n * n} blocks), q={args.quality} "
                      f"niter={args.niter}, {name}, threads={cores}, best of {reps}",
            "seconds": secs}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run (WORLD_SIZE={world})", file=sys.stderr)
            sys.exit(2)
        args.gpus = world
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_note = None
    if world > 1:
        if args.backend == "nccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)                      # brings RCCL up (or fails) before anything is timed
                torch.cuda.synchronize()
                assert int(probe.item()) == world
            except Exception as e:                          # noqa: BLE001 -- report a number rather than none
                comm_note = f"RCCL unavailable ({type(e).__name__}: {str(e)[:120]}); halo rows staged through the host over gloo"
                print(f"bench.py rank {rank}: {comm_note}", file=sys.stderr, flush=True)
                if dist.is_initialized():
                    dist.destroy_process_group()
                args.backend = "gloo"
        if args.backend != "nccl":
            dist.init_process_group("gloo", rank=rank, world_size=world)

    pkg = jpegqs_pkg.load()
    hip = pkg.HipQS()          # raises if the HIP library is missing: no fallback
    flags = pkg.flags_for_quality(args.quality)
    size = args.size
    hblk_total, wblk = size // 8, size // 8

    # ---- band owned by this rank (jpeg-quantsmooth_amd/bands.py)
    from jpeg_quantsmooth_amd import bands
    if args.weak or world == 1:
        topo = bands.BandTopology(0, 1, 0, hblk_total)       # a whole plane per rank
    else:
        r0, r1 = bands.band_rows(hblk_total, world, rank)
        topo = bands.BandTopology(rank, world, r0, r1)
    r0, r1, hblk = topo.r0, topo.r1, topo.hblk
    total_blocks = (hblk_total * wblk) * (world if args.weak else 1)

    full, quant = synth_input_gpu(torch, pkg, size, args.jpeg_quality, dev)
    pristine = full[r0:r1].contiguous()
    cpu_sample = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_sample = full.cpu().numpy()
    band_for_verify = full[: min(16, hblk_total)].contiguous().cpu().numpy() if (args.verify and rank == 0) else None
    edge_checks = []
    if args.verify and world > 1 and not args.weak:
        # the two block rows on our side of each band edge, checked against the oracle run on a
        # crop around the edge (a block after n iterations depends only on blocks within n of it)
        m = args.niter + 2
        for edge, mine in [e for e in ((r1, (r1 - 2, r1)) if topo.down is not None else None,
                                       (r0, (r0, r0 + 2)) if topo.up is not None else None) if e]:
            lo, hi = max(0, edge - 2 - m), min(hblk_total, edge + 2 + m)
            edge_checks.append((lo, mine, full[lo:hi].contiguous().cpu().numpy()))
    del full

    nsteps = args.steps + args.warmup
    work = [pristine.clone() for _ in range(nsteps)]          # one resident plane per step
    eng = bands.HipBandEngine(hip, torch, work[0], quant, flags, luma=1, device=dev)
    stream = torch.cuda.current_stream()
    ev_pairs = []
    sharded = topo.up is not None or topo.down is not None

    comm = eng.comm_scope() if (sharded and args.overlap) else None
    exch = bands.exchange_halo_dist if args.backend == "nccl" else bands.exchange_halo_dist_hostcopy

    def step(coef, timed):
        eng.rebind(coef)
        if sharded:
            # default: pass A, halo exchange, pass B in stream order (bands.run_band); --overlap:
            # bands.run_band_overlapped.  Per-kernel event timing is an N = 1 matter (roofline is
            # reported there)
            if args.overlap:
                bands.run_band_overlapped(eng, topo, args.niter, lambda: exch(eng, topo, dist), comm=comm)
            else:
                bands.run_band(eng, topo, args.niter, lambda: exch(eng, topo, dist))
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

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(work[i], False)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(work[args.warmup + i], True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    edges_ok = None
    if args.verify and world > 1 and not args.weak:
        from oracle.oracle import Oracle
        got = work[-1].cpu().numpy()                      # the last step's result, this rank's band
        ok = True
        for lo, (a0, a1), crop in edge_checks:
            want = Oracle().do_quantsmooth([crop], [quant], flags, args.niter, threads=0)["coefs"][0]
            ok &= bool(np.array_equal(got[a0 - r0:a1 - r0], want[a0 - lo:a1 - lo]))
        flag = torch.tensor([int(ok)], dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        edges_ok = bool(flag.item())
    assert not eng.bad_coef(), "range check tripped on synthetic input"

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_pairs])) if ev_pairs else float("nan")
    band_blocks = hblk * wblk

    if rank == 0:
        value = total_blocks * args.steps / elapsed
        if not ev_pairs:   # sharded run: no per-kernel events; derive from the step time (comm included)
            kern_ms = elapsed / args.steps / args.niter * 1e3
        achieved_gbs = band_blocks * ALGO_BYTES_PER_BLOCK_ITER / (kern_ms * 1e-3) / 1e9
        achieved_tf = band_blocks * FLOP_PER_BLOCK_ITER[flags & 1] / (kern_ms * 1e-3) / 1e12
        traffic = None
        pmc = ROOT / "profiles" / "pmc_traffic.json"
        if pmc.exists():
            try:
                traffic = json.loads(pmc.read_text()).get(f"q{args.quality}_{size}")
            except Exception:
                traffic = None
        out = {
            "metric": "8x8 blocks/s at q=%d niter=%d" % (args.quality, args.niter),
            "value": value, "unit": "blocks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "mpixels_per_s": value * 64 / 1e6,
            "config": {"workload": f"{size}x{size} luma plane ({hblk_total * wblk} blocks), jpegqs --quality {args.quality} "
                                   f"(flags={flags}) --niter {args.niter}, synthetic JPEG-quality-{args.jpeg_quality} coefficients",
                       "sharding": "none" if world == 1 else f"{world} block-row bands, 1-pixel-row halo over "
                                                               f"{'RCCL' if args.backend == 'nccl' else 'gloo (host-staged)'} per iteration, "
                                                               + ("exchange overlapped with the interior rows" if args.overlap
                                                                  else "exchange between pass A and pass B in stream order"),
                       "blocks_per_gpu": band_blocks, **({"comm_note": comm_note} if comm_note else {})},
            "roofline": {"bound": "hbm", "kernel": "qs_smooth_plane_kernel", "achieved": achieved_gbs,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": band_blocks * ALGO_BYTES_PER_BLOCK_ITER,
                         "note": "kernel is FP32-VALU-bound (~290 flop/B); see roofline_valu"},
            "roofline_valu": {"bound": "fp32-valu (separate mul/add, FMA forbidden by bit-exactness)",
                              "achieved": achieved_tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": achieved_tf / VALU_PEAK_TFLOPS,
                              "flop_per_block_iter": FLOP_PER_BLOCK_ITER[flags & 1]},
        }
        if cpu_sample is not None:
            out["cpu_baseline"] = cpu_baseline(pkg, args, cpu_sample, quant, flags)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        if edges_ok is not None:
            out["verify_band_edges_ok"] = edges_ok
        if band_for_verify is not None:
            from oracle.oracle import Oracle
            n = band_for_verify.shape[0]
            want = Oracle().do_quantsmooth([band_for_verify], [quant], flags, args.niter, threads=0)["coefs"][0]
            got = work[-1][: n].cpu().numpy()
            safe = n - args.niter - 1
            out["verify_rows"] = safe
            out["verify_ok"] = bool(np.array_equal(got[:safe], want[:safe]))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
