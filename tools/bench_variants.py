#!/usr/bin/env python3
"""Time qs_smooth_plane_kernel for every tuning variant in build/variants/
(measurement only; results of all variants must be identical)."""
import hashlib
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import jpegqs_pkg  # noqa: E402
import bench  # noqa: E402

pkg = jpegqs_pkg.load()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
coef, quant = bench.synth_input_gpu(torch, pkg, size, 50, dev)
hb, wb = coef.shape[:2]
libs = [pkg.lib_path()] + sorted((ROOT / "build" / "variants").glob("libjpegqs_hip_*.so"))
for flags in (0, 1):
    ref_hash = None
    for lib in libs:
        hip = pkg.HipQS(lib)
        d_cst = torch.from_numpy(hip.consts_build(quant, flags)).to(dev)
        d_plane = torch.zeros(hip.plane_bytes(wb, hb), dtype=torch.uint8, device=dev)
        d_status = torch.zeros(1, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        times = []
        for rep in range(4):
            c = coef.clone()
            hip.idct_plane(d_cst.data_ptr(), c.data_ptr(), d_plane.data_ptr(), wb, hb, 1, 1, 1, d_status.data_ptr(), s)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); hip.smooth_plane(d_cst.data_ptr(), c.data_ptr(), d_plane.data_ptr(), wb, hb, flags, 1, 0, s); e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        h = hashlib.md5(c.cpu().numpy().tobytes()).hexdigest()[:8]
        if ref_hash is None:
            ref_hash = h
        ms = min(times[1:])
        nblk = hb * wb
        print(f"flags={flags} {lib.name:40s} {ms:8.3f} ms  {nblk / ms / 1e6:7.3f} Gblk-iter/s  hash={h} {'OK' if h == ref_hash else 'DIFF'}", flush=True)
