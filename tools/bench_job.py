#!/usr/bin/env python3
"""End-to-end timing of the JOB layer (host arrays in, host arrays out: includes
H2D/D2H over PCIe, device allocation, all launches) for the BASELINE configs."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import jpegqs_pkg
pkg = jpegqs_pkg.load(); hip = pkg.HipQS(); synth = pkg.synth

def run(name, coefs, quants, flags, niter, **kw):
    nblk = sum(c.shape[0] * c.shape[1] for c in coefs)
    ts = []
    for rep in range(4):
        t0 = time.perf_counter(); hip.do_quantsmooth(coefs, quants, flags, niter, **kw); ts.append(time.perf_counter() - t0)
    # the python wrapper copies the inputs (np.copy) before the call: measure that and subtract
    t0 = time.perf_counter(); _ = [c.copy() for c in coefs]; tcopy = time.perf_counter() - t0
    best = min(ts[1:]) - tcopy
    print(f"{name:46s} blocks={nblk:8d} first={ts[0]*1e3:8.2f} ms  steady={best*1e3:8.2f} ms  {nblk/best/1e6:8.2f} Mblocks/s (PCIe-inclusive)", flush=True)

c, q = synth.synth_gray(64, 64, 50); run("C0 64x64 gray q3 n3", [c], [q], 0, 3)
j = synth.synth_ycc(1920, 1080, 2, 2, 50); kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(1920, 1080))
run("C1 1920x1080 4:2:0 q3 n3", j["coefs"], j["quants"], 0, 3, **kw)
run("C1' 1920x1080 4:2:0 q6 n3", j["coefs"], j["quants"], 7, 3, **kw)
c, q = synth.synth_gray(8192, 8192, 50); run("C2 8192x8192 gray q4 n3", [c], [q], 1, 3)
run("C2' 8192x8192 gray q3 n3", [c], [q], 0, 3)
j = synth.synth_ycc(4096, 4096, 2, 2, 50); kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(4096, 4096))
run("C4' 4096x4096 4:2:0 q6 n5", j["coefs"], j["quants"], 7, 5, **kw)
