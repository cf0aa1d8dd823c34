#!/usr/bin/env python3
"""What ONE rank of an N-GPU strong-scaling run of bench.py does per step, emulated on one GPU: the middle
band (1/N of the rows) of a batch of 12 planes of 8192^2, niter x {pass A, halo rows, pass B}, with the halo
exchange replaced by a device copy of the same size.  Two schedules: one launch per plane and pass
(bands.run_bands_batched) against one launch per pass for the whole batch (plane sets,
bands.run_bands_batched_sets: what bench.py --gpus N uses).  Prints ms per plane and the implied speed-up over
the one-GPU step (kernel time only: no interconnect latency).
    python tools/bench_band_sets.py [--quality 3] [--batch 12]"""
import argparse
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import jpegqs_pkg  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--quality", type=int, default=3)
ap.add_argument("--batch", type=int, default=12)
ap.add_argument("--niter", type=int, default=3)
args = ap.parse_args()
pkg = jpegqs_pkg.load(); hip = pkg.HipQS()
from jpeg_quantsmooth_amd import bands  # noqa: E402
dev = torch.device("cuda:0")
flags = pkg.flags_for_quality(args.quality)
full, quant = bench.synth_input_gpu(torch, pkg, 8192, 50, dev)
hb_total = full.shape[0]


def fake_exchange(engs):
    for e in engs:                                   # same bytes as the real exchange: two rows in, (two rows out)
        h = e.hblk * 8
        e.row(-1).copy_(e.row(0)); e.row(h).copy_(e.row(h - 1))


def run(world, sets, steps=6):
    if world == 1:
        topo = bands.BandTopology(0, 1, 0, hb_total)
    else:
        r = world // 2
        r0, r1 = bands.band_rows(hb_total, world, r)
        topo = bands.BandTopology(r, world, r0, r1)
    src = full[topo.r0:topo.r1].contiguous()
    work = [[src.clone() for _ in range(args.batch)] for _ in range(steps + 1)]
    engs = [bands.HipBandEngine(hip, torch, work[0][b], quant, flags, luma=1, device=dev) for b in range(args.batch)]
    ex = (lambda: fake_exchange(engs)) if world > 1 else (lambda: None)
    def step(planes):
        for e, p in zip(engs, planes):
            e.rebind(p)
        if sets:
            bands.run_bands_batched_sets(hip, engs, topo, args.niter, ex)
        else:
            bands.run_bands_batched(engs, topo, args.niter, ex)
    step(work[0]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(work[1 + i])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps / args.batch * 1e3


base = run(1, False)
print(f"q{args.quality}: one GPU, whole plane: {base:.3f} ms per plane   (as one plane set of {args.batch}: {run(1, True):.3f} ms)")
for world in (2, 4, 8):
    a, b = run(world, False), run(world, True)
    print(f"  1/{world} band: per-plane launches {a:.3f} ms ({base / a:.2f}x)   plane-set launches {b:.3f} ms ({base / b:.2f}x)", flush=True)
