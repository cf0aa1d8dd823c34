#!/usr/bin/env python3
"""Locate GPU-vs-oracle mismatches on a YCbCr job (debug aid; run on the GPU box).
    python tools/debug_colour.py width height hsamp vsamp quality niter"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import jpegqs_pkg
from oracle.oracle import Oracle
pkg = jpegqs_pkg.load(); hip = pkg.HipQS(); O = Oracle()
w, h, hs, vs, q, niter = (int(a) for a in sys.argv[1:7])
flags = pkg.flags_for_quality(q)
j = pkg.synth.synth_ycc(w, h, hs, vs, 50)
kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(w, h))
a = hip.do_quantsmooth(j["coefs"], j["quants"], flags, niter, **kw)
b = O.do_quantsmooth(j["coefs"], j["quants"], flags, niter, threads=0, **kw)
print(f"{w}x{h} {hs}x{vs} q{q} flags={flags} niter={niter}: ret {a['ret']} {b['ret']} up {a.get('up')} {b.get('up')}")
for ci, (x, y) in enumerate(zip(a["coefs"], b["coefs"])):
    if x.shape != y.shape:
        print(f"  comp {ci}: shape {x.shape} vs {y.shape}"); continue
    bad = np.argwhere(x != y)
    blocks = sorted(set(map(tuple, bad[:, :2])))
    print(f"  comp {ci} {x.shape}: {len(bad)} coef mismatches in {len(blocks)} blocks; rows {sorted(set(b[0] for b in blocks))[:20]} cols {sorted(set(b[1] for b in blocks))[:20]}")
    for by, bx, i in bad[:6]:
        print(f"     block({by},{bx}) coef {i}: gpu {x[by, bx, i]} oracle {y[by, bx, i]}")
