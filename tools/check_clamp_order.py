"""Sensitivity check for the clamp-after-refresh order (debug aid): `cpu` pins the oracle against the
reference on the extreme-block jobs; `gpu` compares the product library (and build/oldclamp/, a build of the
pre-fix host code, if present) with the oracle."""
import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import jpegqs_pkg
from oracle.oracle import Oracle, Reference
pkg = jpegqs_pkg.load(); synth = pkg.synth
import helpers as m
mode = sys.argv[1]
for (w, h), samp in (((200, 136), (2, 2)), ((136, 88), (1, 1)), ((176, 72), (2, 1))):
    j = m.inject_extreme_blocks(synth.synth_ycc(w, h, samp[0], samp[1], quality=60, seed=5))
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(w, h))
    for flags in (3, 7):
        o = Oracle().do_quantsmooth(j["coefs"], j["quants"], flags, 1, threads=0, **kw)
        if mode == "cpu":
            r = Reference("none").do_quantsmooth(j["coefs"], j["quants"], flags, 1, threads=0, **kw)
            print((w, h), samp, flags, "ret", o["ret"], "oracle==ref", all(np.array_equal(a, b) for a, b in zip(o["coefs"], r["coefs"])),
                  "max |coef| out", [int(np.abs(c).max()) for c in o["coefs"]])
        else:
            for lib in (None, "/root/repo/build/oldclamp/libjpegqs_hip_oldclamp.so"):
                g = pkg.HipQS(lib).do_quantsmooth(j["coefs"], j["quants"], flags, 1, **kw)
                print((w, h), samp, flags, "old" if lib else "new", [int((a != b).sum()) for a, b in zip(g["coefs"], o["coefs"])])
