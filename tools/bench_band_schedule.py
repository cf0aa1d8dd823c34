#!/usr/bin/env python3
"""What one rank of an N-GPU run does per step, measured on ONE GPU: a middle band of
the 8192^2 workload (both neighbours present), with the halo "exchange" replaced by a
device-to-device copy of the band's own edge rows (the NCCL latency is not modelled;
kernel scheduling, the side stream and the launch count are exactly bench.py's).

    python tools/bench_band_schedule.py [--size 8192] [--world 8] [--steps 50]
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import jpegqs_pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--quality", type=int, default=3)
    ap.add_argument("--niter", type=int, default=3)
    args = ap.parse_args()
    import torch
    pkg = jpegqs_pkg.load()
    from jpeg_quantsmooth_amd import bands
    import bench
    hip = pkg.HipQS()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    flags = pkg.flags_for_quality(args.quality)
    hb_total = args.size // 8
    rank = args.world // 2 if args.world > 2 else 0
    r0, r1 = bands.band_rows(hb_total, args.world, rank)
    topo = bands.BandTopology(rank, args.world, r0, r1)
    full, quant = bench.synth_input_gpu(torch, pkg, args.size, 50, dev)
    pristine = full[r0:r1].contiguous()
    del full
    n = args.steps + 5
    work = [pristine.clone() for _ in range(n)]
    eng = bands.HipBandEngine(hip, torch, work[0], quant, flags, luma=1, device=dev)

    def fake_exchange():
        if topo.up is not None:
            eng.row(-1).copy_(eng.row(0), non_blocking=True)
        if topo.down is not None:
            eng.row(eng.hblk * 8).copy_(eng.row(eng.hblk * 8 - 1), non_blocking=True)

    out = {"band_rows": [r0, r1], "world": args.world, "size": args.size}
    for name in ("simple", "overlapped"):
        comm = eng.comm_scope()
        def step(c):
            eng.rebind(c)
            if name == "simple":
                bands.run_band(eng, topo, args.niter, fake_exchange)
            else:
                bands.run_band_overlapped(eng, topo, args.niter, fake_exchange, comm=comm)
        for i in range(5):
            step(work[i])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(work[5 + i])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        out[name + "_ms_per_step"] = dt * 1e3
        out[name + "_implied_blocks_per_s_at_world"] = hb_total * (args.size // 8) / dt
        for i in range(n):
            work[i].copy_(pristine)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
