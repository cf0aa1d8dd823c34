"""CPU stand-in for HipBandEngine, built on the TEST ORACLE (tests only): lets
the band/halo driver in jpeg-quantsmooth_amd/bands.py run under the gloo
backend without a GPU.  Uses the product's plane geometry so the halo rows
that travel are the same bytes the GPU path would send."""
import ctypes as C

import numpy as np
import torch

APRON_X = 16  # QS_APRON_X in csrc/qs_device.h


class OracleBandEngine:
    def __init__(self, oracle, hip, coef, quant, flags, luma=1, plane=None):
        self.o = oracle
        self.coef = np.ascontiguousarray(coef, dtype=np.int16)   # updated in place
        self.hblk, self.wblk = self.coef.shape[:2]
        self.quant = np.ascontiguousarray(quant, dtype=np.uint16)
        self.flags, self.luma = flags, luma
        self.pitch = hip.plane_pitch(self.wblk)
        self._row_off = lambda y: hip.plane_row_offset(self.wblk, y)
        self.plane = plane if plane is not None else torch.zeros(hip.plane_bytes(self.wblk, self.hblk), dtype=torch.uint8)
        self._bad = C.c_int(0)
        lib = oracle.lib
        self._idct = lib.qso_band_idct
        self._idct.restype = None
        self._idct.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.POINTER(C.c_int)]
        self._smooth = lib.qso_band_smooth
        self._smooth.restype = None
        self._smooth.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int]

    def idct(self, first, rep_top, rep_bot):
        self._idct(self.coef.ctypes.data, self.wblk, self.hblk, self.quant.ctypes.data, int(first),
                   self.plane.data_ptr(), self.pitch, APRON_X, int(rep_top), int(rep_bot), C.byref(self._bad))

    def smooth(self, final_clamp):
        self.smooth_rows(0, self.hblk, final_clamp)

    def smooth_rows(self, row0, row1, final_clamp):
        f = self.o.lib.qso_band_smooth_rows
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        f(self.coef.ctypes.data, self.wblk, self.hblk, self.quant.ctypes.data,
          self.plane.data_ptr(), self.pitch, APRON_X, self.flags, self.luma, int(final_clamp), row0, row1)

    def smooth_next(self, final_clamp, write_next, rep_top=1, rep_bot=1):
        """the fused form of the product engine, restated with two oracle calls: pass B, then pass A of the next
        iteration into the SECOND plane (unclamped coefficients: the clamp comes last), and the planes swap"""
        self.smooth(False if write_next else final_clamp)
        if write_next:
            if getattr(self, "plane2", None) is None:
                self.plane2 = torch.zeros_like(self.plane)
            self.plane, self.plane2 = self.plane2, self.plane
            self.idct(False, rep_top, rep_bot)
            assert not final_clamp, "the tests' fused loop clamps on the last iteration only, which writes no next plane"

    def row(self, y):
        o = self._row_off(y)
        return self.plane[o:o + self.pitch]

    def bad_coef(self):
        return bool(self._bad.value)
