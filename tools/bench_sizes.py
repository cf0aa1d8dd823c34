#!/usr/bin/env python3
"""Pass-B launch time against plane height at 8192 px width (1024 blocks per row):
hblk = 64 rows is one wave per SIMD on the whole chip.  Shows the latency floor
(a lone wave), the occupancy steps and where the kernel reaches its streaming rate.
    python tools/bench_sizes.py [--flags 0] [--rows 64,128,...] [lib ...]"""
import hashlib
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import jpegqs_pkg  # noqa: E402
import bench  # noqa: E402

import argparse  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--rows", default="1,8,16,32,48,64,96,112,128,144,160,192,256,384,512,1024")
ap.add_argument("--smooth", action="store_true", help="the smooth variant of the synthetic image (periods x10, no noise)")
ap.add_argument("--jpeg-quality", type=int, default=50, help="IJG quality the synthetic plane is encoded at (95 / 98: quantiser entries equal to 1, which the kernels skip)")
ap.add_argument("libs", nargs="*")
args = ap.parse_args()
pkg = jpegqs_pkg.load()
flags = args.flags
libs = [Path(p) for p in args.libs] or [pkg.lib_path()]
dev = torch.device("cuda:0")
full, quant = bench.synth_input_gpu(torch, pkg, 8192, args.jpeg_quality, dev, smooth=args.smooth)
print("JPEG quality", args.jpeg_quality, ": AC quantiser entries equal to 1:", int((quant[1:] == 1).sum()), "of 63")
wb = 1024
rows = [int(r) for r in args.rows.replace("+", ",").split(",")]
print("flags", flags, "rows:", rows)
for lib in libs:
    hip = pkg.HipQS(lib)
    d_cst = torch.from_numpy(hip.consts_build(quant, flags)).to(dev)
    out, hashes = [], []
    for hb in rows:
        src = full[:hb].contiguous()
        d_plane = torch.zeros(hip.plane_bytes(wb, hb), dtype=torch.uint8, device=dev)
        d_status = torch.zeros(1, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        times = []
        for rep in range(5):
            c = src.clone()
            hip.idct_plane(d_cst.data_ptr(), c.data_ptr(), d_plane.data_ptr(), wb, hb, 1, 1, 1, d_status.data_ptr(), s)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); hip.smooth_plane(d_cst.data_ptr(), c.data_ptr(), d_plane.data_ptr(), wb, hb, flags, 1, 0, s); e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        out.append(min(times[1:]))
        hashes.append(hashlib.md5(c.cpu().numpy().tobytes()).hexdigest()[:6])
    print(f"{lib.name:36s} " + " ".join(f"{t * 1e3:7.0f}" for t in out) + "  us", flush=True)
    print(f"{'  result md5':36s} " + " ".join(f"{h:>7s}" for h in hashes), flush=True)
    print(f"{'  Gblk-iter/s':36s} " + " ".join(f"{hb * wb / t / 1e6:7.3f}" for hb, t in zip(rows, out)), flush=True)
