"""ctypes bindings for the CPU oracles -- TEST INFRASTRUCTURE ONLY.

Two libraries share one flat job ABI:
  * oracle/libqs_oracle.so        -- our plain-C restatement (qso_* symbols)
  * oracle/_ref/libqsref_<v>.so   -- the unmodified reference behind
                                     ref_harness.c (qsref_* symbols),
                                     v in {none, sse2, avx2, avx512}

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product path never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
MAXC = 4


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


PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)


def build_oracle() -> Path:
    """compile oracle/libqs_oracle.so (gcc); cheap, idempotent."""
    subprocess.run(["make", "-s", "-C", str(HERE), "oracle"], check=True)
    return HERE / "libqs_oracle.so"


def build_ref() -> bool:
    """compile oracle/_ref/*.so when /root/reference is mounted (build box only)."""
    if not Path("/root/reference/quantsmooth.h").exists():
        return False
    subprocess.run(["make", "-s", "-C", str(HERE), "ref"], check=True)
    return True


def ref_path(variant: str = "none") -> Path:
    return HERE / "_ref" / f"libqsref_{variant}.so"


def have_ref(variant: str = "none") -> bool:
    return ref_path(variant).exists()


def cpu_has(flag: str) -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return flag in line.split()
    except OSError:
        pass
    return False


def best_ref_variant() -> str | None:
    """fastest reference build this host can execute (for cpu_baseline)."""
    if have_ref("avx512") and all(cpu_has(f) for f in ("avx512f", "avx512bw", "avx512dq", "fma")):
        return "avx512"
    if have_ref("avx2") and cpu_has("avx2") and cpu_has("fma"):
        return "avx2"
    if have_ref("sse2"):
        return "sse2"
    if have_ref("none"):
        return "none"
    return None


class _Lib:
    """common driver for the flat job ABI; `prefix` is 'qso' or 'qsref'."""

    def __init__(self, path: Path, prefix: str):
        self.path = Path(path)
        self.lib = C.CDLL(str(path))
        self.prefix = prefix
        self._run = getattr(self.lib, f"{prefix}_do_quantsmooth")
        self._run.restype = C.c_int
        self._run.argtypes = [C.POINTER(Job), C.c_int, C.c_int, C.c_int, C.c_int, PROGRESS_FN, C.c_void_p]
        self._free = getattr(self.lib, f"{prefix}_free")
        self._free.argtypes = [C.c_void_p]
        self._free.restype = None

    def fn(self, name, restype=None, argtypes=None):
        f = getattr(self.lib, f"{self.prefix}_{name}")
        f.restype = restype
        if argtypes is not None:
            f.argtypes = argtypes
        return f

    def do_quantsmooth(self, coefs, quants, flags, niter, *, hsamp=None, vsamp=None,
                       colorspace=None, image_size=None, threads=1, progprec=0, progress=None):
        """Run the whole plane driver on copies of the inputs.

        coefs: list of int16 arrays [hblk, wblk, 64] (quantised); quants: list of
        uint16[64] or None (component without a table).
        -> dict(ret, coefs, quants, up (bool), hsamp0, vsamp0)
        """
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
        cb = PROGRESS_FN(progress) if progress else C.cast(None, PROGRESS_FN)
        ret = self._run(C.byref(job), flags, niter, threads, progprec, cb, None)
        up = job.up_wblk > 0
        if up:
            for j in range(2):
                cnt = job.up_wblk * job.up_hblk * 64
                buf = (C.c_int16 * cnt).from_address(job.coef_up[j])
                work[1 + j] = np.frombuffer(buf, dtype=np.int16).reshape(job.up_hblk, job.up_wblk, 64).copy()
                self._free(job.coef_up[j])
        qout = [np.array(job.quant[ci][:], dtype=np.uint16) if quants[ci] is not None else None
                for ci in range(n)]
        return dict(ret=ret, coefs=work, quants=qout, up=up,
                    hsamp0=job.out_hsamp0, vsamp0=job.out_vsamp0)


class Oracle(_Lib):
    """our restatement (libqs_oracle.so)"""

    def __init__(self, path: Path | None = None):
        path = path or (HERE / "libqs_oracle.so")
        if not Path(path).exists():
            build_oracle()
        super().__init__(path, "qso")
        u8p, i16p, u16p, f32p = (C.POINTER(t) for t in (C.c_uint8, C.c_int16, C.c_uint16, C.c_float))
        self._idct = self.fn("idct_islow", None, [i16p, u8p, C.c_int])
        self._idctf = self.fn("idct_float", None, [f32p, f32p])
        self._fdctf = self.fn("fdct_float", None, [f32p, f32p])
        self._tables = self.fn("tables", C.c_int, [C.c_int, f32p])
        self._tsize = self.fn("table_size", C.c_int, [C.c_int])
        self._block = self.fn("block", None, [i16p, u16p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int])
        self._interval = self.fn("interval", None, [C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 3)
        self._recip = self.fn("interval_recip", C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int)])
        self._prep = self.fn("quant_prep", None, [u16p, u16p, C.POINTER(C.c_int), C.POINTER(C.c_int)])

    def idct_islow(self, coef):
        c = np.ascontiguousarray(coef, dtype=np.int16)
        out = np.zeros(64, dtype=np.uint8)
        self._idct(c.ctypes.data_as(C.POINTER(C.c_int16)), out.ctypes.data_as(C.POINTER(C.c_uint8)), 8)
        return out

    def idct_float(self, x):
        a = np.ascontiguousarray(x, dtype=np.float32); o = np.zeros(64, np.float32)
        self._idctf(a.ctypes.data_as(C.POINTER(C.c_float)), o.ctypes.data_as(C.POINTER(C.c_float)))
        return o

    def fdct_float(self, x):
        a = np.ascontiguousarray(x, dtype=np.float32); o = np.zeros(64, np.float32)
        self._fdctf(a.ctypes.data_as(C.POINTER(C.c_float)), o.ctypes.data_as(C.POINTER(C.c_float)))
        return o

    def tables(self, flags):
        size = self._tsize(flags)
        out = np.zeros((64, size), dtype=np.float32)
        self._tables(flags, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def quant_prep(self, q):
        q = np.ascontiguousarray(q, dtype=np.uint16); eff = np.zeros(64, np.uint16)
        a, b = C.c_int(0), C.c_int(0)
        self._prep(q.ctypes.data_as(C.POINTER(C.c_uint16)), eff.ctypes.data_as(C.POINTER(C.c_uint16)),
                   C.byref(a), C.byref(b))
        return eff, bool(a.value), bool(b.value)

    def interval(self, coef, div):
        o, lo, hi = C.c_int(0), C.c_int(0), C.c_int(0)
        self._interval(coef, div, C.byref(o), C.byref(lo), C.byref(hi))
        return o.value, lo.value, hi.value

    def block(self, coef, eff_quant, plane, bx, by, flags, luma=1, plane2=None):
        """plane: uint8 [h+2, w+2] with apron; processes block (bx, by) in place on a copy"""
        c = np.ascontiguousarray(coef, dtype=np.int16).copy()
        q = np.ascontiguousarray(eff_quant, dtype=np.uint16)
        assert plane.flags.c_contiguous
        stride = plane.shape[1]
        off = (by * 8 + 1) * stride + bx * 8 + 1
        p2 = None
        if plane2 is not None:
            assert plane2.shape == plane.shape and plane2.flags.c_contiguous
            p2 = plane2.ctypes.data + off
        self._block(c.ctypes.data_as(C.POINTER(C.c_int16)), q.ctypes.data_as(C.POINTER(C.c_uint16)),
                    plane.ctypes.data + off, p2, stride, flags, luma)
        return c


class Reference(_Lib):
    """the compiled, unmodified reference behind ref_harness.c"""

    def __init__(self, variant: str = "none"):
        super().__init__(ref_path(variant), "qsref")
        self.variant = variant
        u8p, i16p, u16p, f32p = (C.POINTER(t) for t in (C.c_uint8, C.c_int16, C.c_uint16, C.c_float))
        self._idct = self.fn("idct_islow", None, [i16p, u8p, C.c_int])
        self._idctf = self.fn("idct_float", None, [f32p, f32p])
        self._fdctf = self.fn("fdct_float", None, [f32p, f32p])
        self._tables = self.fn("tables", C.c_int, [C.c_int, f32p])
        self._block = self.fn("block", None, [i16p, u16p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int])
        v = self.fn("variant", C.c_char_p, [])
        self.compiled_variant = v().decode()

    def idct_islow(self, coef):
        c = np.ascontiguousarray(coef, dtype=np.int16)
        out = np.zeros(64, dtype=np.uint8)
        self._idct(c.ctypes.data_as(C.POINTER(C.c_int16)), out.ctypes.data_as(C.POINTER(C.c_uint8)), 8)
        return out

    def idct_float(self, x):
        a = np.ascontiguousarray(x, dtype=np.float32); o = np.zeros(64, np.float32)
        self._idctf(a.ctypes.data_as(C.POINTER(C.c_float)), o.ctypes.data_as(C.POINTER(C.c_float)))
        return o

    def fdct_float(self, x):
        a = np.ascontiguousarray(x, dtype=np.float32); o = np.zeros(64, np.float32)
        self._fdctf(a.ctypes.data_as(C.POINTER(C.c_float)), o.ctypes.data_as(C.POINTER(C.c_float)))
        return o

    def tables(self, flags):
        size = 272 if flags & 1 else 160
        out = np.zeros((64, size), dtype=np.float32)
        got = self._tables(flags, out.ctypes.data_as(C.POINTER(C.c_float)))
        assert got == size
        return out

    @staticmethod
    def quantval192(q):
        """the 192-entry quantval[] the reference's driver builds
        (reference quantsmooth.h:2506-2539): [q(0->1), x1, x2]"""
        out = np.zeros(192, dtype=np.uint16)
        for i in range(64):
            v = int(q[i]) or 1
            n = v.bit_length() - 1
            x1 = ((0x10000 << n) + v - 1) // v
            if n:
                x1 |= x1 >> 16
            x2 = (-0x8000) >> n
            out[i] = v; out[64 + i] = x1 & 0xFFFF; out[128 + i] = x2 & 0xFFFF
        return out

    def block(self, coef, eff_quant, plane, bx, by, flags, luma=1, plane2=None):
        c = np.ascontiguousarray(coef, dtype=np.int16).copy()
        q = self.quantval192(eff_quant)
        stride = plane.shape[1]
        off = (by * 8 + 1) * stride + bx * 8 + 1
        p2 = None
        if plane2 is not None:
            p2 = plane2.ctypes.data + off
        self._block(c.ctypes.data_as(C.POINTER(C.c_int16)), q.ctypes.data_as(C.POINTER(C.c_uint16)),
                    plane.ctypes.data + off, p2, stride, flags, luma)
        return c


def verify_bands(oracle, coef_in, quant, flags, niter, got, rows=16, luma=True):
    """Exact check of a LARGE single-component plane against the oracle without running
    the oracle on all of it: `rows` block rows at the top, in the middle and at the bottom
    (together they cover the image's top/bottom edges and, over their whole width, its
    left/right edge columns).  A block's result after n iterations depends only on blocks
    within n block rows of it, so the oracle runs on a crop with niter + 1 margin rows.

    coef_in(r0, r1) / got(r0, r1) -> int16 [r1 - r0, wblk, 64] numpy arrays (input and
    result rows); hblk = coef_in.hblk.  -> list of dict(where, row0, row1, bad_blocks)."""
    hblk = coef_in.hblk
    rows = min(rows, hblk)
    m = niter + 1
    spans = [("top", 0, rows)]
    if hblk > 3 * rows:
        mid = (hblk - rows) // 2
        spans.append(("middle", mid, mid + rows))
    if hblk > rows:
        spans.append(("bottom", hblk - rows, hblk))
    out = []
    for name, a, b in spans:
        lo, hi = max(0, a - m), min(hblk, b + m)
        crop = np.ascontiguousarray(coef_in(lo, hi))
        want = oracle.do_quantsmooth([crop], [quant], flags, niter, threads=0,
                                     colorspace=1 if luma else 3)["coefs"][0][a - lo:b - lo]
        have = np.asarray(got(a, b))
        bad = int((have != want).any(axis=2).sum())
        out.append(dict(where=name, row0=a, row1=b, bad_blocks=bad))
    return out


class RowSource:
    """adapter for verify_bands: rows of a numpy array or of a torch tensor (device or host)"""

    def __init__(self, arr):
        self.arr = arr
        self.hblk = int(arr.shape[0])

    def __call__(self, r0, r1):
        a = self.arr[r0:r1]
        return a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
