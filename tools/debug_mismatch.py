#!/usr/bin/env python3
"""Locate GPU-vs-oracle mismatches (debug aid; run on the GPU box)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import jpegqs_pkg
from oracle.oracle import Oracle
pkg = jpegqs_pkg.load(); hip = pkg.HipQS(); O = Oracle()
for (w, h, flags) in ((64, 64, 0), (200, 120, 1), (200, 120, 0), (520, 264, 0)):
    coef, quant = pkg.synth.synth_gray(w, h, 50)
    for niter in (1, 2, 3):
        a = hip.do_quantsmooth([coef], [quant], flags, niter)["coefs"][0]
        b = O.do_quantsmooth([coef], [quant], flags, niter, threads=0)["coefs"][0]
        bad = np.argwhere(a != b)
        print(f"{w}x{h} flags={flags} niter={niter}: {len(bad)} coef mismatches in {len(set(map(tuple, bad[:, :2])))} blocks")
        for by, bx, i in bad[:8]:
            print(f"   block({by},{bx}) coef {i}: gpu {a[by, bx, i]} oracle {b[by, bx, i]} in {coef[by, bx, i] * quant[i]}")
