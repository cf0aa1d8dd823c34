"""ctypes view of the CPU back end inside libjpegqs.so (csrc/qs_cpu.c) for the parity tests: same calling
convention and result dict as HipQS.do_quantsmooth / the oracles.  Test-side only -- the product reaches this code
through do_quantsmooth() of include/libjpegqs.h when no HIP device is visible."""
import ctypes as C
import os
from pathlib import Path

import numpy as np

import jpegqs_pkg

pkg = jpegqs_pkg.load()
from jpeg_quantsmooth_amd.hipqs import PKG_DIR, PROGRESS_FN, Job, HipQS, _share_hip_runtime_with_torch  # noqa: E402


class CpuBackend:
    def __init__(self):
        _share_hip_runtime_with_torch()
        jpeg = Path("/opt/conda/lib/libjpeg.so.9")
        if jpeg.exists():   # the library's libjpeg-facing half wants libjpeg's symbols (the application's, normally)
            C.CDLL(str(jpeg), mode=os.RTLD_GLOBAL | os.RTLD_LAZY)
        self.lib = C.CDLL(str(PKG_DIR / "libjpegqs.so"), mode=os.RTLD_LAZY)
        self._bind()

    def _bind(self):
        f = self.lib.qs_cpu_do_quantsmooth
        f.restype = C.c_int
        f.argtypes = [C.POINTER(Job), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, PROGRESS_FN, C.c_void_p]
        self.lib.qs_cpu_free.argtypes = [C.c_void_p]
        self.lib.qs_cpu_free.restype = None
        self.lib.qs_cpu_isa.restype = C.c_char_p
        self.lib.qs_cpu_lanes.restype = C.c_int

    def isa(self):
        return self.lib.qs_cpu_isa().decode()

    def lanes(self):
        return self.lib.qs_cpu_lanes()

    def do_quantsmooth(self, coefs, quants, flags, niter, *, hsamp=None, vsamp=None, colorspace=None, image_size=None,
                       progprec=0, progress=None, threads=0, by_rows=False):
        job, work = HipQS._make_job(coefs, quants, hsamp, vsamp, colorspace, image_size)
        rows_arg, keep = None, []
        if by_rows:   # the libjpeg-facing form: one pointer per block row, job->coef ignored
            outer = (C.c_void_p * job.ncomp)()
            for ci in range(job.ncomp):
                a = work[ci]
                inner = (C.c_void_p * a.shape[0])(*[a[y].ctypes.data for y in range(a.shape[0])])
                keep.append(inner)
                outer[ci] = C.addressof(inner)
                job.coef[ci] = None
            rows_arg = C.addressof(outer)
            keep.append(outer)
        cb = PROGRESS_FN(progress) if progress else C.cast(None, PROGRESS_FN)
        ret = self.lib.qs_cpu_do_quantsmooth(C.byref(job), rows_arg, flags, niter, threads, progprec, cb, None)
        if ret < 0:
            raise RuntimeError(f"qs_cpu_do_quantsmooth failed with {ret}")
        up = job.up_wblk > 0
        if up:
            for j in range(2):
                cnt = job.up_wblk * job.up_hblk * 64
                buf = (C.c_int16 * cnt).from_address(job.coef_up[j])
                work[1 + j] = np.frombuffer(buf, dtype=np.int16).reshape(job.up_hblk, job.up_wblk, 64).copy()
                self.lib.qs_cpu_free(job.coef_up[j])
        qout = [np.array(job.quant[ci][:], dtype=np.uint16) if quants[ci] is not None else None for ci in range(job.ncomp)]
        return dict(ret=ret, coefs=work, quants=qout, up=up, hsamp0=job.out_hsamp0, vsamp0=job.out_vsamp0)
