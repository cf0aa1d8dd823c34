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

import ctypes as C
from jpeg_quantsmooth_amd import hipqs


def run(name, coefs, quants, flags, niter, hsamp=None, vsamp=None, colorspace=None, image_size=None):
    """times the C entry point itself (the Python wrapper's defensive copies are outside the clock)"""
    n = len(coefs)
    nblk = sum(c.shape[0] * c.shape[1] for c in coefs)
    ts = []
    for rep in range(8):
        job = hipqs.Job(); job.ncomp = n
        job.colorspace = colorspace if colorspace is not None else (3 if n == 3 else 1)
        work = [np.ascontiguousarray(c).copy() for c in coefs]
        for ci in range(n):
            job.hblk[ci], job.wblk[ci] = work[ci].shape[:2]
            job.hsamp[ci], job.vsamp[ci] = (hsamp or [1] * n)[ci], (vsamp or [1] * n)[ci]
            job.coef[ci] = work[ci].ctypes.data; job.has_quant[ci] = 1
            for i in range(64): job.quant[ci][i] = int(quants[ci][i])
        job.image_width, job.image_height = image_size or (work[0].shape[1] * 8, work[0].shape[0] * 8)
        t0 = time.perf_counter()
        rc = hip.lib.qs_hip_do_quantsmooth(C.byref(job), flags, niter, 0, C.cast(None, hipqs.PROGRESS_FN), None)
        ts.append(time.perf_counter() - t0)
        assert rc == 0, hip.lib.qs_hip_last_error()
        for j in range(2):
            if job.coef_up[j]: hip.lib.qs_hip_free(job.coef_up[j])
    best = min(ts[1:])
    print(f"{name:46s} blocks={nblk:8d} first={ts[0]*1e3:8.2f} ms  steady={best*1e3:8.2f} ms  {nblk/best/1e6:8.2f} Mblocks/s (PCIe-inclusive)"
          f"   all: {' '.join('%.2f' % (t * 1e3) for t in ts)}", flush=True)

c, q = synth.synth_gray(64, 64, 50); run("C0 64x64 gray q3 n3", [c], [q], 0, 3)
j = synth.synth_ycc(1920, 1080, 2, 2, 50); kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(1920, 1080))
run("C1 1920x1080 4:2:0 q3 n3", j["coefs"], j["quants"], 0, 3, **kw)
run("C1' 1920x1080 4:2:0 q6 n3", j["coefs"], j["quants"], 7, 3, **kw)
c, q = synth.synth_gray(8192, 8192, 50); run("C2 8192x8192 gray q4 n3", [c], [q], 1, 3)
run("C2' 8192x8192 gray q3 n3", [c], [q], 0, 3)
j = synth.synth_ycc(4096, 4096, 2, 2, 50); kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(4096, 4096))
run("C4' 4096x4096 4:2:0 q6 n5", j["coefs"], j["quants"], 7, 5, **kw)
