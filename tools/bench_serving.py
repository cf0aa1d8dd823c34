#!/usr/bin/env python3
"""Throughput of many small jobs submitted concurrently from host threads (the
job layer is thread-safe: each call leases its own streams and pooled buffers).
1920x1080 4:2:0 --quality 3 --niter 3, PCIe-inclusive."""
import sys, time, threading
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import jpegqs_pkg
from oracle.oracle import Oracle
pkg = jpegqs_pkg.load(); hip = pkg.HipQS(); synth = pkg.synth
j = synth.synth_ycc(1920, 1080, 2, 2, 50)
kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(1920, 1080))
ref = hip.do_quantsmooth(j["coefs"], j["quants"], 0, 3, **kw)
want = Oracle().do_quantsmooth(j["coefs"], j["quants"], 0, 3, threads=0, **kw)
assert all(np.array_equal(a, b) for a, b in zip(ref["coefs"], want["coefs"]))
nblk = sum(c.shape[0] * c.shape[1] for c in j["coefs"])
for nthreads in (1, 2, 4, 8, 16):
    per = 24
    ok = [True] * nthreads
    def worker(t):
        for _ in range(per):
            r = hip.do_quantsmooth(j["coefs"], j["quants"], 0, 3, **kw)
            if not all(np.array_equal(a, b) for a, b in zip(r["coefs"], ref["coefs"])):
                ok[t] = False
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    n = nthreads * per
    print(f"threads={nthreads:2d}: {n / dt:8.1f} images/s  {n * nblk / dt / 1e6:8.2f} Mblocks/s  all bit-exact={all(ok)}", flush=True)
