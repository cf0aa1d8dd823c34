#!/usr/bin/env python3
"""Where does a wave of the recovery kernel spend its shader cycles?  Needs the measurement build
    tools/build_variants.sh "timeline:-DQS_TIMELINE=1"
(every wave of qs_smooth_plane_kernel accumulates s_memtime deltas per phase, csrc/qs_smooth_kernel.inc) and
QS_HIP_DP=0 (one block per lane at every size).  8192 px wide planes of N block rows: 64 rows = one wave per SIMD.
    QS_HIP_DP=0 python tools/timeline.py [--flags 0] [--rows 64+128+192+1024]"""
import argparse
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import jpegqs_pkg  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--rows", default="64+128+192+1024")
ap.add_argument("--smooth", action="store_true")
a = ap.parse_args()
pkg = jpegqs_pkg.load()
hip = pkg.HipQS(ROOT / "build" / "variants" / "libjpegqs_hip_timeline.so")
setbuf = hip.lib.qs_hip_debug_timeline
setbuf.restype = C.c_int; setbuf.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
full, quant = bench.synth_input_gpu(torch, pkg, 8192, 50, dev, smooth=a.smooth)
wb = 1024
# instructions per wave by phase (profiles/r02*: SQ_INSTS_VALU per wave and the ISA listing), q3 / q4
INSTR = {0: dict(refresh=953 * 14, terms=68800, update=4000), 1: dict(refresh=953 * 14, terms=124000, update=4000)}[a.flags & 1]
print(f"# flags {a.flags}; per wave: shader cycles (s_memtime) mean over waves; 'cyc/instr' = cycles of the phase / its VALU instructions")
print(f"# {'rows':>5s} {'waves/SIMD':>10s} {'kernel us':>9s} {'life':>9s} {'stage':>7s} {'refresh':>8s} {'terms':>8s} {'update':>8s} {'rest':>7s} |"
      f" {'refresh%':>8s} {'terms%':>7s} {'update%':>7s} | cyc/instr: refresh terms update")
for hb in [int(r) for r in a.rows.replace("+", ",").split(",")]:
    src = full[:hb].contiguous()
    nw = (hb * wb + 63) // 64
    tl = torch.zeros(((nw + 3) // 4 * 4) * 8, dtype=torch.int64, device=dev)
    assert setbuf(tl.data_ptr()) == 0
    d_cst = torch.from_numpy(hip.consts_build(quant, a.flags)).to(dev)
    d_plane = torch.zeros(hip.plane_bytes(wb, hb), dtype=torch.uint8, device=dev)
    d_status = torch.zeros(1, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    best = None
    for rep in range(3):
        c = src.clone()
        hip.idct_plane(d_cst.data_ptr(), c.data_ptr(), d_plane.data_ptr(), wb, hb, 1, 1, 1, d_status.data_ptr(), s)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); hip.smooth_plane(d_cst.data_ptr(), c.data_ptr(), d_plane.data_ptr(), wb, hb, a.flags, 1, 0, s); e1.record()
        torch.cuda.synchronize()
        best = e0.elapsed_time(e1) if best is None else min(best, e0.elapsed_time(e1))
    t = tl.cpu().numpy().reshape(-1, 8)[:nw].astype(np.float64)
    life, stage, refresh, terms, update = (t[:, k].mean() for k in range(5))
    rest = life - stage - refresh - terms - update
    start = t[:, 5]
    print(f"  {hb:5d} {nw / 1024:10.2f} {best * 1e3:9.0f} {life:9.0f} {stage:7.0f} {refresh:8.0f} {terms:8.0f} {update:8.0f} {rest:7.0f} |"
          f" {100 * refresh / life:8.1f} {100 * terms / life:7.1f} {100 * update / life:7.1f} |"
          f"  {refresh / INSTR['refresh']:6.2f} {terms / INSTR['terms']:5.2f} {update / INSTR['update']:6.2f}"
          f"   (first..last wave start spread {(start.max() - start.min()):.0f} cycles)", flush=True)
