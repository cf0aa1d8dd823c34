#!/usr/bin/env python3
"""Throughput of many small jobs submitted concurrently from host threads (the
job layer is thread-safe: each call leases its own streams and pooled buffers).
PCIe-inclusive, host arrays in and out.  The threads live in a small C program
(tools/bench_serving.c, built here) that calls qs_hip_do_quantsmooth directly;
this script makes the job, checks one result against the oracle and prints the
C program's JSON lines.

    python tools/bench_serving.py [width height [quality [niter]]]     default 1920 1080 3 3, 4:2:0
    SERVING_CONFIGS=1x32,2x16 ...                                       only these threads x batch combinations
"""
import struct
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import jpegqs_pkg  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

pkg = jpegqs_pkg.load(); hip = pkg.HipQS(); synth = pkg.synth
w = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
quality = int(sys.argv[3]) if len(sys.argv) > 3 else 3
niter = int(sys.argv[4]) if len(sys.argv) > 4 else 3
flags = pkg.flags_for_quality(quality)
j = synth.synth_ycc(w, h, 2, 2, 50)
kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(w, h))
got = hip.do_quantsmooth(j["coefs"], j["quants"], flags, niter, **kw)
want = Oracle().do_quantsmooth(j["coefs"], j["quants"], flags, niter, threads=0, **kw)
assert all(np.array_equal(a, b) for a, b in zip(got["coefs"], want["coefs"])), "GPU result differs from the oracle"

out = Path("/tmp/qs_serving"); out.mkdir(exist_ok=True)
with open(out / "job.bin", "wb") as f:
    f.write(struct.pack("<4i", len(j["coefs"]), 3, w, h))
    for c, q, hs, vs in zip(j["coefs"], j["quants"], j["hsamp"], j["vsamp"]):
        f.write(struct.pack("<4i", c.shape[1], c.shape[0], hs, vs))
        f.write(np.asarray(q, dtype="<u2").tobytes())
    for c in j["coefs"]:
        f.write(np.ascontiguousarray(c, dtype="<i2").tobytes())
exe = out / "bench_serving"
libdir = Path(pkg.lib_path()).parent
subprocess.check_call(["gcc", "-O2", "-o", str(exe), str(ROOT / "tools" / "bench_serving.c"), f"-I{ROOT / 'include'}",
                       f"-L{libdir}", "-ljpegqs_hip", f"-Wl,-rpath,{libdir}", "-lpthread"])
print(f"# {w}x{h} 4:2:0 --quality {quality} --niter {niter}, {sum(c.shape[0] * c.shape[1] for c in j['coefs'])} blocks per image", flush=True)
import os  # noqa: E402
configs = ((1, 1), (2, 1), (4, 1), (8, 1), (16, 1), (1, 4), (1, 8), (1, 16), (1, 32), (2, 8), (2, 16), (4, 8), (4, 16))
if os.environ.get("SERVING_CONFIGS"):          # e.g. "1x32,2x16": threads x batch
    configs = tuple(tuple(int(v) for v in c.split("x")) for c in os.environ["SERVING_CONFIGS"].split(","))
for nthreads, batch in configs:
    per = max(2 * batch, 256 // nthreads)
    r = subprocess.run([str(exe), str(out / "job.bin"), str(flags), str(niter), str(nthreads), str(per), str(batch)],
                       capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr.strip(), flush=True)
