import ctypes as C, os, sys, time
os.environ["QS_HIP_TRACE"] = "1"
sys.path.insert(0, "/root/repo")
import numpy as np, jpegqs_pkg
pkg = jpegqs_pkg.load(); hip = pkg.HipQS(); synth = pkg.synth
from jpeg_quantsmooth_amd import hipqs
c, q = synth.synth_gray(8192, 8192, 50)
for flags in (0, 1):
    for rep in range(4):
        job, work = hip._make_job([c], [q])
        t0 = time.perf_counter()
        rc = hip.lib.qs_hip_do_quantsmooth(C.byref(job), flags, 3, 0, C.cast(None, hipqs.PROGRESS_FN), None)
        print(f"flags {flags} rep {rep}: {1e3 * (time.perf_counter() - t0):.2f} ms", file=sys.stderr, flush=True)
