"""ctypes binding of libjpegqs_hip.so (include/jpegqs_hip.h)."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
MAXC = 4


class FLAGS:
    """JPEGQS_* algorithm flags, numerically the reference's (libjpegqs.h:14-32)."""
    DIAGONALS = 1
    JOINT_YUV = 2
    UPSAMPLE_UV = 4
    LOW_QUALITY = 8
    NO_REBALANCE = 16
    NO_REBALANCE_UV = 32
    TRANSCODE = 64
    MASK = 0x7F
    ITER_MAX = 100


def flags_for_quality(quality: int) -> int:
    """the jpegqs CLI's --quality -> flags mapping (reference quantsmooth.c:380-393)."""
    q = int(quality)
    flags = 0
    if q < 3:
        flags |= FLAGS.LOW_QUALITY
        q += 4
    if q >= 4:
        flags |= FLAGS.DIAGONALS
    if q >= 5:
        flags |= FLAGS.JOINT_YUV
    if q >= 6:
        flags |= FLAGS.UPSAMPLE_UV
    return flags


class QsHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libjpegqs_hip error {code}: {msg}")
        self.code = code


class Job(C.Structure):
    _fields_ = [
        ("ncomp", C.c_int32), ("colorspace", C.c_int32),
        ("image_width", C.c_int32), ("image_height", C.c_int32),
        ("wblk", C.c_int32 * MAXC), ("hblk", C.c_int32 * MAXC),
        ("hsamp", C.c_int32 * MAXC), ("vsamp", C.c_int32 * MAXC),
        ("has_quant", C.c_int32 * MAXC),
        ("quant", (C.c_uint16 * 64) * MAXC),
        ("coef", C.c_void_p * MAXC),
        ("coef_up", C.c_void_p * 2),
        ("up_wblk", C.c_int32), ("up_hblk", C.c_int32),
        ("out_hsamp0", C.c_int32), ("out_vsamp0", C.c_int32),
    ]


class PlaneRef(C.Structure):
    """qs_hip_plane_ref: one plane of a plane-set launch (device pointers)"""
    _fields_ = [("d_consts", C.c_void_p), ("d_coef", C.c_void_p), ("d_plane", C.c_void_p), ("d_status", C.c_void_p),
                ("wblk", C.c_int32), ("hblk", C.c_int32), ("luma", C.c_int32), ("band", C.c_int32)]


class PlaneRefs:
    """what HipQS.plane_refs() returns: the qs_hip_plane_ref array of a plane-set launch plus the PARALLEL array of second
    planes qs_hip_smooth_planes_next takes (None when no plane has one)"""
    def __init__(self, arr, nxt):
        self.arr, self.next = arr, nxt

    def __len__(self):
        return len(self.arr)

    def __getitem__(self, i):
        return self.arr[i]


MAX_PLANES = 56
PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)

# every symbol include/jpegqs_hip.h declares: (restype, argtypes)
ABI = {
    "qs_hip_do_quantsmooth": (C.c_int, [C.POINTER(Job), C.c_int, C.c_int, C.c_int, PROGRESS_FN, C.c_void_p]),
    "qs_hip_do_quantsmooth_batch": (C.c_int, [C.POINTER(C.POINTER(Job)), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "qs_hip_do_quantsmooth_rows": (C.c_int, [C.POINTER(Job), C.POINTER(C.POINTER(C.c_void_p)), C.c_int, C.c_int, C.c_int,
                                          PROGRESS_FN, C.c_void_p]),
    "qs_hip_set_devices": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "qs_hip_set_shard_schedule": (C.c_int, [C.c_int]),
    "qs_hip_do_quantsmooth_band": (C.c_int, [C.POINTER(Job), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "qs_hip_do_quantsmooth_sharded": (C.c_int, [C.POINTER(Job), C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "qs_hip_band_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "qs_hip_colour_band_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4),
    "qs_hip_band_halo_rows": (C.c_int, [C.c_int, C.c_int] + [C.POINTER(C.c_size_t)] * 5),
    "qs_hip_prewarm": (C.c_int, [C.POINTER(Job), C.c_int, C.c_int]),
    "qs_hip_progress_calls": (C.c_int, [C.POINTER(Job), C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]),
    "qs_hip_free": (None, [C.c_void_p]),
    "qs_hip_release_cache": (None, []),
    "qs_hip_device_count": (C.c_int, []),
    "qs_hip_last_error": (C.c_char_p, []),
    "qs_hip_consts_bytes": (C.c_size_t, []),
    "qs_hip_plane_pitch": (C.c_size_t, [C.c_int]),
    "qs_hip_plane_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "qs_hip_plane_row_offset": (C.c_size_t, [C.c_int, C.c_int]),
    "qs_hip_consts_build": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint16), C.c_int]),
    "qs_hip_idct_plane": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "qs_hip_smooth_plane": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p]),
    "qs_hip_smooth_plane_next": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "qs_hip_smooth_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "qs_hip_idct_planes": (C.c_int, [C.POINTER(PlaneRef), C.c_int, C.c_int, C.c_void_p]),
    "qs_hip_smooth_planes": (C.c_int, [C.POINTER(PlaneRef), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "qs_hip_smooth_planes_next": (C.c_int, [C.POINTER(PlaneRef), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "qs_hip_abi_version": (C.c_int, []),
    "qs_hip_joint_plane": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_void_p]),
    "qs_hip_lowq_plane": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "qs_hip_downsample_plane": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_void_p]),
    "qs_hip_upsample_pitch": (C.c_size_t, [C.c_int, C.c_int]),
    "qs_hip_upsample_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "qs_hip_upsample_plane": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "qs_hip_upsample_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                       C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "qs_hip_fdct_plane": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "qs_hip_clamp_plane": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "qs_hip_dequant_plane": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
}


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so (SONAME libamdhip64.so.7, but requested by torch under the
    unversioned name), so loading our library first would pull in /opt/rocm's
    copy and torch would later load a SECOND runtime that sees no GPUs.  If a
    torch install is present, map its copy first; our NEEDED libamdhip64.so.7
    then resolves to it by SONAME and torch finds the same file by inode."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if cand.exists():
        try:
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib_path() -> Path:
    """the in-tree library; QS_HIP_LIB points measurement runs at an A/B build (tools/build_variants.sh)"""
    import os
    return Path(os.environ["QS_HIP_LIB"]) if os.environ.get("QS_HIP_LIB") else PKG_DIR / "libjpegqs_hip.so"


def load_library(path: Path | None = None) -> C.CDLL:
    """dlopen the HIP library; raises (no fallback) when it has not been built."""
    p = Path(path) if path else lib_path()
    if not p.exists():
        raise FileNotFoundError(
            f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {PKG_DIR / 'csrc'}`; there is no CPU fallback")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(str(p))
    for name, (res, args) in ABI.items():
        f = getattr(lib, name)  # AttributeError if the ABI is incomplete
        f.restype = res
        f.argtypes = args
    return lib


class HipQS:
    """Thin object wrapper: job layer on numpy arrays, plane layer on device pointers."""

    def __init__(self, path: Path | None = None):
        self.lib = load_library(path)

    # -- helpers ---------------------------------------------------------------
    def _check(self, rc: int) -> int:
        if rc < 0:
            raise QsHipError(rc, self.lib.qs_hip_last_error().decode(errors="replace"))
        return rc

    def device_count(self) -> int:
        return self.lib.qs_hip_device_count()

    def consts_bytes(self) -> int:
        return self.lib.qs_hip_consts_bytes()

    def plane_pitch(self, wblk: int) -> int:
        return self.lib.qs_hip_plane_pitch(wblk)

    def plane_bytes(self, wblk: int, hblk: int) -> int:
        return self.lib.qs_hip_plane_bytes(wblk, hblk)

    def plane_row_offset(self, wblk: int, y: int) -> int:
        return self.lib.qs_hip_plane_row_offset(wblk, y)

    def consts_build(self, quant, flags: int) -> np.ndarray:
        """host-side constant block (uint8 array) for one component"""
        q = np.ascontiguousarray(quant, dtype=np.uint16)
        out = np.zeros(self.consts_bytes(), dtype=np.uint8)
        self._check(self.lib.qs_hip_consts_build(out.ctypes.data, q.ctypes.data_as(C.POINTER(C.c_uint16)), flags))
        return out

    # -- band arithmetic (one definition, shared with csrc/qs_shard.cpp) ----------
    def band_rows(self, hblk: int, nbands: int, band: int, align: int = 1):
        r0, r1 = C.c_int(0), C.c_int(0)
        self._check(self.lib.qs_hip_band_rows(hblk, nbands, band, align, C.byref(r0), C.byref(r1)))
        return r0.value, r1.value

    def colour_band_rows(self, hblk_luma: int, hblk_chroma: int, v_samp: int, nbands: int, band: int):
        v = [C.c_int(0) for _ in range(4)]
        self._check(self.lib.qs_hip_colour_band_rows(hblk_luma, hblk_chroma, v_samp, nbands, band, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def band_halo_rows(self, wblk: int, hblk: int):
        """-> (send_top, send_bot, recv_top, recv_bot, nbytes): byte offsets inside a band's pixel plane"""
        v = [C.c_size_t(0) for _ in range(5)]
        self._check(self.lib.qs_hip_band_halo_rows(wblk, hblk, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    # -- job layer -------------------------------------------------------------
    @staticmethod
    def _make_job(coefs, quants, hsamp=None, vsamp=None, colorspace=None, image_size=None):
        n = len(coefs)
        job = Job()
        job.ncomp = n
        job.colorspace = colorspace if colorspace is not None else (3 if n == 3 else 1)
        hsamp = hsamp or [1] * n
        vsamp = vsamp or [1] * n
        work = []
        for ci in range(n):
            a = np.ascontiguousarray(coefs[ci], dtype=np.int16).copy()
            assert a.ndim == 3 and a.shape[2] == 64
            work.append(a)
            job.hblk[ci], job.wblk[ci] = a.shape[0], a.shape[1]
            job.hsamp[ci], job.vsamp[ci] = hsamp[ci], vsamp[ci]
            job.coef[ci] = a.ctypes.data
            if quants[ci] is not None:
                job.has_quant[ci] = 1
                for i in range(64):
                    job.quant[ci][i] = int(quants[ci][i])
        if image_size is None:
            mh, mv = max(hsamp), max(vsamp)
            image_size = (work[0].shape[1] * 8 * mh // hsamp[0], work[0].shape[0] * 8 * mv // vsamp[0])
        job.image_width, job.image_height = image_size
        return job, work

    def _job_result(self, job, work, quants, ret):
        up = job.up_wblk > 0
        if up:
            for j in range(2):
                cnt = job.up_wblk * job.up_hblk * 64
                buf = (C.c_int16 * cnt).from_address(job.coef_up[j])
                work[1 + j] = np.frombuffer(buf, dtype=np.int16).reshape(job.up_hblk, job.up_wblk, 64).copy()
                self.lib.qs_hip_free(job.coef_up[j])
        qout = [np.array(job.quant[ci][:], dtype=np.uint16) if quants[ci] is not None else None
                for ci in range(job.ncomp)]
        return dict(ret=ret, coefs=work, quants=qout, up=up,
                    hsamp0=job.out_hsamp0, vsamp0=job.out_vsamp0)

    def do_quantsmooth(self, coefs, quants, flags, niter, *, hsamp=None, vsamp=None,
                       colorspace=None, image_size=None, progprec=0, progress=None, threads=None, devices=None):
        """Whole do_quantsmooth() on copies of the inputs; same calling convention
        and result dict as the test oracles (`threads` is accepted and ignored:
        the GPU has no use for jpegqs_control_t.threads).  devices=[...]: cut the job
        over these HIP devices (qs_hip_do_quantsmooth_sharded; an ordinal may repeat)."""
        job, work = self._make_job(coefs, quants, hsamp, vsamp, colorspace, image_size)
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            ret = self._check(self.lib.qs_hip_do_quantsmooth_sharded(C.byref(job), flags, niter, arr, len(devices)))
            return self._job_result(job, work, quants, ret)
        cb = PROGRESS_FN(progress) if progress else C.cast(None, PROGRESS_FN)
        ret = self._check(self.lib.qs_hip_do_quantsmooth(C.byref(job), flags, niter, progprec, cb, None))
        return self._job_result(job, work, quants, ret)

    def progress_calls(self, coefs, quants, niter, progprec=0, **kw):
        """qs_hip_progress_calls: [(cur, max), ...] the progress callback of this job will see (no device needed)"""
        job, _work = self._make_job(coefs, quants, kw.get("hsamp"), kw.get("vsamp"), kw.get("colorspace"), kw.get("image_size"))
        out = (C.c_int * 4096)()
        mx = C.c_int(0)
        n = self._check(self.lib.qs_hip_progress_calls(C.byref(job), niter, progprec, out, 4096, C.byref(mx)))
        return [(int(out[k]), int(mx.value)) for k in range(min(n, 4096))]

    def set_devices(self, devices):
        """qs_hip_set_devices: the device list large jobs are spread over ([] = default)"""
        arr = (C.c_int * max(1, len(devices)))(*devices)
        self._check(self.lib.qs_hip_set_devices(arr, len(devices)))

    def do_quantsmooth_band(self, coefs, quants, flags, niter, rank, nranks, comm=None, **kw):
        """qs_hip_do_quantsmooth_band: this rank's band of a job whose halo rows travel through RCCL (`comm`: ncclComm_t as int)"""
        job, work = self._make_job(coefs, quants, kw.get("hsamp"), kw.get("vsamp"), kw.get("colorspace"), kw.get("image_size"))
        ret = self._check(self.lib.qs_hip_do_quantsmooth_band(C.byref(job), flags, niter, rank, nranks, comm))
        return self._job_result(job, work, quants, ret)

    def set_shard_schedule(self, schedule: int):
        """qs_hip_set_shard_schedule: 0 = one halo row per iteration, 1 = deep halo (no exchange), -1 = default"""
        self._check(self.lib.qs_hip_set_shard_schedule(schedule))

    def do_quantsmooth_batch(self, jobs, flags, niter):
        """qs_hip_do_quantsmooth_batch: `jobs` = list of dicts with the keyword
        arguments of do_quantsmooth (coefs, quants and optionally hsamp, vsamp,
        colorspace, image_size) -> list of result dicts in the same order
        (ret < 0: that job failed with this error code)"""
        made = [self._make_job(j["coefs"], j["quants"], j.get("hsamp"), j.get("vsamp"),
                               j.get("colorspace"), j.get("image_size")) for j in jobs]
        ptrs = (C.POINTER(Job) * len(made))(*[C.pointer(m[0]) for m in made])
        results = (C.c_int * max(1, len(made)))()
        self._check(self.lib.qs_hip_do_quantsmooth_batch(ptrs, len(made), flags, niter, results))
        return [self._job_result(m[0], m[1], j["quants"], int(results[i])) for i, (m, j) in enumerate(zip(made, jobs))]

    # -- plane layer (device pointers as ints, stream as int or None) ----------
    def idct_plane(self, d_consts, d_coef, d_plane, wblk, hblk, first, rep_top, rep_bot, d_status, stream=None):
        self._check(self.lib.qs_hip_idct_plane(d_consts, d_coef, d_plane, wblk, hblk, int(first),
                                               int(rep_top), int(rep_bot), d_status, stream))

    def smooth_plane(self, d_consts, d_coef, d_plane, wblk, hblk, flags, luma=1, final_clamp=0, stream=None):
        self._check(self.lib.qs_hip_smooth_plane(d_consts, d_coef, d_plane, wblk, hblk, flags,
                                                 int(luma), int(final_clamp), stream))

    def smooth_plane_next(self, d_consts, d_coef, d_plane, d_plane_next, wblk, hblk, flags, luma=1, final_clamp=0,
                          rep_top=1, rep_bot=1, stream=None):
        """pass B that also writes the NEXT iteration's pixel plane (fused pass A) into d_plane_next"""
        self._check(self.lib.qs_hip_smooth_plane_next(d_consts, d_coef, d_plane, d_plane_next, wblk, hblk, flags,
                                                      int(luma), int(final_clamp), int(rep_top), int(rep_bot), stream))

    @staticmethod
    def plane_refs(planes):
        """[(d_consts, d_coef, d_plane, d_status, wblk, hblk, luma[, band[, d_plane_next]])] -> PlaneRefs for the
        *_planes calls; band: bit 0 / bit 1 = the top / bottom apron row is a halo row (a band of a sharded plane);
        d_plane_next: the plane the smoothing launch writes the next iteration's pixels into (None: none) -- these go
        into the parallel array of qs_hip_smooth_planes_next, the struct itself has no such field"""
        arr = (PlaneRef * len(planes))()
        nxt = (C.c_void_p * len(planes))()
        any_next = False
        for i, (r, p) in enumerate(zip(arr, planes)):
            cst, coef, plane, status, wb, hb, luma = p[:7]
            r.d_consts, r.d_coef, r.d_plane, r.d_status = cst, coef, plane, status
            r.wblk, r.hblk, r.luma = wb, hb, int(luma)
            r.band = int(p[7]) if len(p) > 7 else 0
            nxt[i] = p[8] if len(p) > 8 else None
            any_next = any_next or bool(nxt[i])
        return PlaneRefs(arr, nxt if any_next else None)

    def idct_planes(self, refs, first, stream=None):
        self._check(self.lib.qs_hip_idct_planes(refs.arr, len(refs), int(first), stream))

    def smooth_planes(self, refs, flags, final_clamp=0, stream=None):
        if refs.next is not None:
            self._check(self.lib.qs_hip_smooth_planes_next(refs.arr, refs.next, len(refs), flags, int(final_clamp), stream))
        else:
            self._check(self.lib.qs_hip_smooth_planes(refs.arr, len(refs), flags, int(final_clamp), stream))

    def smooth_rows(self, d_consts, d_coef, d_plane, wblk, hblk, row0, row1, flags, luma=1, final_clamp=0, stream=None):
        self._check(self.lib.qs_hip_smooth_rows(d_consts, d_coef, d_plane, wblk, hblk, row0, row1, flags,
                                                int(luma), int(final_clamp), stream))

    def joint_plane(self, d_consts, d_coef, d_plane, d_lowres, wblk, hblk, rebalance=0, final_clamp=0, stream=None):
        self._check(self.lib.qs_hip_joint_plane(d_consts, d_coef, d_plane, d_lowres, wblk, hblk,
                                                int(rebalance), int(final_clamp), stream))

    def lowq_plane(self, d_consts, d_coef, d_plane, wblk, hblk, rebalance=1, final_clamp=0, stream=None):
        self._check(self.lib.qs_hip_lowq_plane(d_consts, d_coef, d_plane, wblk, hblk, int(rebalance), int(final_clamp), stream))

    def downsample_plane(self, d_luma, ywblk, yhblk, d_lowres, lwblk, lhblk, ws, hs, stream=None):
        self._check(self.lib.qs_hip_downsample_plane(d_luma, ywblk, yhblk, d_lowres, lwblk, lhblk, ws, hs, stream))

    def upsample_pitch(self, image_width, ws):
        return self.lib.qs_hip_upsample_pitch(image_width, ws)

    def upsample_rows(self, d_chroma, d_lowres, cwblk, d_luma, ywblk, yhblk, d_pixels, pitch, w1, h1, first_rows, ws, hs, stream=None):
        self._check(self.lib.qs_hip_upsample_rows(d_chroma, d_lowres, cwblk, d_luma, ywblk, yhblk, d_pixels, pitch,
                                                  w1, h1, first_rows, ws, hs, stream))

    def fdct_plane(self, d_pixels, pitch, d_coef, wblk, hblk, stream=None):
        self._check(self.lib.qs_hip_fdct_plane(d_pixels, pitch, d_coef, wblk, hblk, stream))

    def clamp_plane(self, d_coef, wblk, hblk, stream=None):
        self._check(self.lib.qs_hip_clamp_plane(d_coef, wblk, hblk, stream))

    def dequant_plane(self, d_consts, d_coef, wblk, hblk, stream=None):
        self._check(self.lib.qs_hip_dequant_plane(d_consts, d_coef, wblk, hblk, stream))
