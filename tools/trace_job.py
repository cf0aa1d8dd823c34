#!/usr/bin/env python3
"""Phase times of single jobs through the C entry point (QS_HIP_TRACE=1 output on stderr) plus the wall
time of each call: where a full-HD frame's milliseconds go.   python tools/trace_job.py [w h]"""
import ctypes as C
import os
import sys
import time
from pathlib import Path

os.environ["QS_HIP_TRACE"] = "1"
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import jpegqs_pkg  # noqa: E402

pkg = jpegqs_pkg.load(); hip = pkg.HipQS(); synth = pkg.synth
from jpeg_quantsmooth_amd import hipqs  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
j = synth.synth_ycc(w, h, 2, 2, 50)
for flags, niter, name in ((0, 3, "q3"), (1, 3, "q4"), (3, 3, "q5"), (7, 3, "q6")):
    for rep in range(4):
        job, work = hip._make_job(j["coefs"], j["quants"], j["hsamp"], j["vsamp"], 3, (w, h))
        sys.stderr.flush()
        t0 = time.perf_counter()
        rc = hip.lib.qs_hip_do_quantsmooth(C.byref(job), flags, niter, 0, C.cast(None, hipqs.PROGRESS_FN), None)
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        for k in range(2):
            if job.coef_up[k]:
                hip.lib.qs_hip_free(job.coef_up[k])
        print(f"{name} rep {rep}: rc={rc} call {dt * 1e3:.3f} ms  (freeing the replacement arrays {1e3 * (time.perf_counter() - t1):.3f} ms)", file=sys.stderr, flush=True)
