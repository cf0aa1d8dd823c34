#!/usr/bin/env python3
"""The PRODUCT's own multi-GPU route on the benchmark image, timed at the C entry point:
qs_hip_do_quantsmooth_sharded (csrc/qs_shard.cpp: one host process, one band per device, halo rows
pulled with hipMemcpyPeerAsync) -- host arrays in, host arrays out, PCIe included.  One device:
qs_hip_do_quantsmooth (fused route, the plane cut into pipelined bands).

bench.py runs this as a child process of rank 0 (a crash or hang here cannot take the headline line
with it) and merges the JSON it prints into its own line as `product_route`.

    python tools/bench_product_route.py --devices 0,1,2,3 [--size 8192 --quality 3 --reps 4]
"""
import argparse
import ctypes as C
import json
import os
import re
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default="0")
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--quality", type=int, default=3, choices=(3, 4))
    ap.add_argument("--niter", type=int, default=3)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--jpeg-quality", type=int, default=50)
    ap.add_argument("--print-trace", action="store_true", help="print the QS_HIP_TRACE lines of the last timed call to stderr")
    a = ap.parse_args()
    devices = [int(d) for d in a.devices.replace("+", ",").split(",") if d != ""]   # ("+" works too: tools/session.sh turns commas into spaces)
    os.environ["QS_HIP_TRACE"] = "1"
    import torch
    import jpegqs_pkg
    import bench
    pkg = jpegqs_pkg.load()
    hip = pkg.HipQS()
    from jpeg_quantsmooth_amd.hipqs import PROGRESS_FN
    flags = pkg.flags_for_quality(a.quality)
    dev = torch.device("cuda", devices[0])
    d_coef, quant = bench.synth_input_gpu(torch, pkg, a.size, a.jpeg_quality, dev)
    coef = d_coef.cpu().numpy()
    del d_coef
    torch.cuda.empty_cache()
    nblk = coef.shape[0] * coef.shape[1]
    arr = (C.c_int * len(devices))(*devices)

    def call(sharded):
        job, work = hip._make_job([coef], [quant])
        # the trace lines go to fd 2: collect them through a pipe
        r, w = os.pipe()
        saved = os.dup(2)
        os.dup2(w, 2)
        t0 = time.perf_counter()
        if sharded:
            rc = hip.lib.qs_hip_do_quantsmooth_sharded(C.byref(job), flags, a.niter, arr, len(devices))
        else:
            rc = hip.lib.qs_hip_do_quantsmooth(C.byref(job), flags, a.niter, 0, C.cast(None, PROGRESS_FN), None)
        ms = (time.perf_counter() - t0) * 1e3
        os.dup2(saved, 2); os.close(saved); os.close(w)
        trace = b""
        while True:
            chunk = os.read(r, 65536)
            if not chunk:
                break
            trace += chunk
        os.close(r)
        if rc != 0:
            raise RuntimeError(f"rc {rc}: {hip.lib.qs_hip_last_error().decode(errors='replace')}")
        return ms, work[0], trace.decode(errors="replace")

    sharded = len(devices) > 1
    one_ms, one, _ = call(False)                         # also the warm-up of device 0's pools
    times, traces, got = [], [], None
    warm = 6     # untimed: cold contexts and pools on the other devices; staging blocks and extra queues come up in the background
    for rep in range(a.reps + warm):
        ms, got, tr = call(sharded)
        if rep >= warm:
            times.append(ms); traces.append(tr)
    if a.print_trace and traces:
        print(traces[-1], file=sys.stderr)
    out = {"entry": "qs_hip_do_quantsmooth_sharded" if sharded else "qs_hip_do_quantsmooth",
           "devices": devices, "image": f"{a.size}x{a.size} luma, q={a.quality} niter={a.niter}",
           "ms_per_image": float(np.median(times)), "ms_all": [round(t, 2) for t in times],
           "blocks_per_s": nblk / (float(np.median(times)) * 1e-3),
           "includes": "gather from / scatter to host arrays and PCIe both ways (not the HBM-resident figure `value` reports)",
           "equals_one_device_result": bool(np.array_equal(got, one)),
           "one_device_ms": one_ms if not sharded else None}
    m = None
    for tr in reversed(traces):
        m = re.search(r"sharded\(set\) (\d+) band\(s\)  upload\+stage ([0-9.]+) ms  enqueue ([0-9.]+) ms  drain\+scatter ([0-9.]+) ms"
                      r"(?:  iterations on device: max ([0-9.]+) ms \(per band:([0-9. ]+)\))?", tr)
        if m:
            break
    if m:
        out["phases_ms"] = {"upload_and_stage": float(m.group(2)), "enqueue": float(m.group(3)), "drain_and_scatter": float(m.group(4))}
        if m.group(5):
            out["iterations_on_device_ms"] = {"max": float(m.group(5)), "per_band": [float(v) for v in m.group(6).split()]}
    else:
        for tr in reversed(traces):
            m2 = re.search(r"fused  .*enqueue ([0-9.]+) ms  drain\+download ([0-9.]+) ms", tr)
            if m2:
                out["phases_ms"] = {"enqueue": float(m2.group(1)), "drain_and_download": float(m2.group(2))}
                m3 = re.search(r"device timeline \(upload done, kernels done, download done\):((?: \[[-0-9. ]+\])+)", tr)
                if m3:   # per band, ms after the first band's upload (HIP events on the bands' streams)
                    out["bands_device_ms"] = [dict(zip(("upload_done", "kernels_done", "download_done"), (float(v) for v in b.split())))
                                              for b in re.findall(r"\[([-0-9. ]+)\]", m3.group(1))]
                break
    # a cheap exact check against the CPU oracle: 16 block rows top / middle / bottom
    try:
        from oracle import oracle as om
        truth = om.Reference("none") if om.have_ref("none") else om.Oracle()
        det = om.verify_bands(truth, om.RowSource(coef), quant, flags, a.niter, om.RowSource(got), rows=16)
        out["verify_ok"] = all(d["bad_blocks"] == 0 for d in det)
    except Exception as e:  # noqa: BLE001
        out["verify_ok"] = None
        out["verify_error"] = repr(e)[:200]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
