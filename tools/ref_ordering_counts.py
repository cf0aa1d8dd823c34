#!/usr/bin/env python3
"""Information: how many blocks differ between the reference's scalar build
(oracle A, what the GPU path is bit-exact to) and its SIMD builds (other summation orders,
fused multiply-adds) on the same input.  CPU only; needs oracle/_ref (build box or a tree
that carries the prebuilt files).   python tools/ref_ordering_counts.py [w h]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import jpegqs_pkg  # noqa: E402
from oracle import oracle as om  # noqa: E402

pkg = jpegqs_pkg.load()
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
j = pkg.synth.synth_ycc(w, h, 2, 2, quality=50, seed=5)
kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(w, h))
total = sum(int(c.shape[0] * c.shape[1]) for c in j["coefs"])
base = om.Reference("none")
print(f"{w}x{h} 4:2:0, JPEG quality 50, niter 3, {total} blocks; blocks that differ from the scalar (SIMD=none) build:")
for quality, flags in ((3, 0), (4, 1), (5, 3), (6, 7)):
    a = base.do_quantsmooth(j["coefs"], j["quants"], flags, 3, threads=0, **kw)
    o = om.Oracle().do_quantsmooth(j["coefs"], j["quants"], flags, 3, threads=0, **kw)
    row = {"oracle port": sum(int((x != y).any(axis=2).sum()) for x, y in zip(a["coefs"], o["coefs"]))}
    for v in ("sse2", "avx2", "avx512"):
        if not om.have_ref(v) or (v == "avx2" and not om.cpu_has("avx2")) or (v == "avx512" and not om.cpu_has("avx512bw")):
            continue
        b = om.Reference(v).do_quantsmooth(j["coefs"], j["quants"], flags, 3, threads=0, **kw)
        row[v] = sum(int((x != y).any(axis=2).sum()) for x, y in zip(a["coefs"], b["coefs"]))
    print(f"  --quality {quality} (flags {flags}): {row}")
