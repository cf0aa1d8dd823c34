#!/usr/bin/env python3
"""Step (b) of tools/first_contact.sh: the PRODUCT's multi-GPU route on real devices, every block checked.

qs_hip_do_quantsmooth_sharded (csrc/qs_shard.cpp: one host process, one block-row band per device, halo rows pulled
with hipMemcpyPeerAsync, cross-device event waits) over devices 0..N-1 for the three big BASELINE configurations --
8192^2 luma q3, 16384^2 luma q3, 8192^2 4:2:0 q6 niter 5 -- each compared BLOCK FOR BLOCK with the compiled,
unmodified reference (oracle/_ref/libqsref_none.so; the plain-C port when it did not travel) and with the one-device
result.  Prints one line per configuration and a final PASS / FAIL; exit code 0 = all equal.

    python tools/first_contact_shard.py [--devices 0,1,...] [--configs 8192q3,16384q3,8192q6] [--small]

--small: 1024^2 / 2048^2 / 1024^2 instead (the `-m gpu` test uses it so that a suite run stays short; the explicit
device list takes the sharded route at any size).
N = 1 (`--devices 0`) is the degenerate form: one band, no exchange -- runs on a one-GPU box.
"""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default=None, help="comma list (default: all visible devices)")
    ap.add_argument("--configs", default="8192q3,16384q3,8192q6")
    ap.add_argument("--small", action="store_true")
    a = ap.parse_args()
    import torch
    import jpegqs_pkg
    import bench
    from oracle import oracle as om
    pkg = jpegqs_pkg.load()
    hip = pkg.HipQS()
    ndev = hip.device_count()
    devices = [int(d) for d in a.devices.split(",")] if a.devices else list(range(ndev))
    if not devices or max(devices) >= ndev:
        print(f"FAIL: devices {devices} but {ndev} visible")
        return 2
    truth = om.Reference("none") if om.have_ref("none") else om.Oracle()
    print(f"first_contact_shard: devices {devices} of {ndev}; checker = "
          f"{'compiled reference (oracle/_ref/libqsref_none.so)' if om.have_ref('none') else 'oracle port'}", flush=True)
    dev0 = torch.device("cuda", devices[0])
    ok_all = True
    for cfg in a.configs.split(","):
        size, q = cfg.split("q")
        size, q = int(size), int(q)
        if a.small:
            size //= 8
        flags = pkg.flags_for_quality(q)
        niter = 5 if q >= 5 else 3
        if q >= 5:
            coefs_d, quants = bench.synth_colour_gpu(torch, pkg, size, 50, dev0)
            coefs = [c.cpu().numpy() for c in coefs_d]
            kw = dict(hsamp=[2, 1, 1], vsamp=[2, 1, 1], colorspace=3, image_size=(size, size))
            del coefs_d
        else:
            c, quant = bench.synth_input_gpu(torch, pkg, size, 50, dev0)
            coefs, quants, kw = [c.cpu().numpy()], [quant], {}
            del c
        torch.cuda.empty_cache()
        t0 = time.time()
        want = truth.do_quantsmooth(coefs, quants, flags, niter, threads=0, **kw)
        t_ref = time.time() - t0
        t0 = time.time()
        got = hip.do_quantsmooth(coefs, quants, flags, niter, devices=devices, **kw)
        t_first = time.time() - t0
        t0 = time.time()
        got2 = hip.do_quantsmooth(coefs, quants, flags, niter, devices=devices, **kw)
        t_second = time.time() - t0
        one = hip.do_quantsmooth(coefs, quants, flags, niter, **kw)
        bad_ref = bad_one = 0
        ok = got["ret"] == want["ret"] == 0 and got2["ret"] == 0
        for ci in range(len(coefs)):
            w, g, g2, o = want["coefs"][ci], got["coefs"][ci], got2["coefs"][ci], one["coefs"][ci]
            if g.shape != w.shape:
                ok = False
                continue
            bad_ref += int((g != w).any(axis=2).sum()) + int((g2 != w).any(axis=2).sum())
            bad_one += int((g != o).any(axis=2).sum())
        ok = ok and bad_ref == 0 and bad_one == 0
        nblk = sum(c.shape[0] * c.shape[1] for c in coefs)
        print(f"  {size}^2 q{q} niter {niter} ({nblk} input blocks) over {len(devices)} device(s): "
              f"{'OK' if ok else 'MISMATCH'} -- blocks differing from the reference {bad_ref}, from the one-device result {bad_one}; "
              f"call {t_first * 1e3:.1f} ms (first) / {t_second * 1e3:.1f} ms (second), reference {t_ref:.1f} s", flush=True)
        ok_all = ok_all and ok
    print("first_contact_shard:", "PASS" if ok_all else "FAIL")
    return 0 if ok_all else 1


if __name__ == "__main__":
    sys.exit(main())
