"""Row-band sharding of one component plane across GPUs (one process per GPU).

The recovery path is a Jacobi iteration: pass B of iteration n reads the pixel
plane written by pass A of the same iteration (all blocks' coefficients before
the iteration) plus its own coefficients, and writes only its own block
(reference quantsmooth.h:2589-2640).  Blocks only look one pixel across their
border (reference :1396-1401), so a plane splits into contiguous block-row
bands that own their coefficients for the whole run and swap ONE PIXEL ROW with
each neighbour after every pass A -- the only data-path communication the
algorithm has.  No reduction exists anywhere.

This module holds the band arithmetic and the per-step loop; the compute
backend is an object with the small `BandEngine` interface below.  The product
backend is `HipBandEngine` (kernels through the flat C ABI, buffers are torch
device tensors so that torch.distributed / RCCL can move the halo rows).  Tests
plug a CPU engine (built on the test oracle) into the same loop to exercise
the N > 1 logic with the gloo backend.
"""
from __future__ import annotations

from dataclasses import dataclass


def band_rows(hblk: int, world: int, rank: int, align: int = 1):
    """block rows [r0, r1) owned by `rank`; band edges fall on multiples of
    `align` block rows (2 for the luma plane of a 4:2:0 image so that chroma
    bands line up with luma bands)."""
    units = -(-hblk // align)
    r0 = min(hblk, (units * rank // world) * align)
    r1 = min(hblk, (units * (rank + 1) // world) * align)
    return r0, r1


@dataclass
class BandTopology:
    rank: int
    world: int
    r0: int
    r1: int

    @property
    def hblk(self):
        return self.r1 - self.r0

    @property
    def up(self):          # rank owning the rows above, or None at the image top
        return self.rank - 1 if self.rank > 0 else None

    @property
    def down(self):
        return self.rank + 1 if self.rank < self.world - 1 else None

    @property
    def rep_top(self):     # image edge: apron row is a replica, not a halo
        return int(self.up is None)

    @property
    def rep_bot(self):
        return int(self.down is None)


class BandEngine:
    """interface a compute backend offers for ONE band of ONE component"""

    hblk: int

    def idct(self, first: bool, rep_top: int, rep_bot: int) -> None: ...
    def smooth(self, final_clamp: bool) -> None: ...
    def smooth_rows(self, row0: int, row1: int, final_clamp: bool) -> None: ...
    def row(self, y: int):
        """1-D uint8 tensor aliasing pixel row y of the band's plane, apron
        columns included; y = -1 and y = hblk*8 are the apron (halo) rows"""
    def bad_coef(self) -> bool: ...


def exchange_halo_dist(engine: BandEngine, topo: BandTopology, dist) -> None:
    """swap edge pixel rows with the neighbouring ranks (RCCL / gloo p2p)"""
    ops = []
    h = engine.hblk * 8
    if topo.up is not None:
        ops.append(dist.P2POp(dist.isend, engine.row(0), topo.up))
        ops.append(dist.P2POp(dist.irecv, engine.row(-1), topo.up))
    if topo.down is not None:
        ops.append(dist.P2POp(dist.isend, engine.row(h - 1), topo.down))
        ops.append(dist.P2POp(dist.irecv, engine.row(h), topo.down))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def exchange_halo_local(engines) -> None:
    """the same exchange between N logical bands living in one process
    (device-to-device copies): used to test the band logic on a single GPU"""
    for upper, lower in zip(engines[:-1], engines[1:]):
        h = upper.hblk * 8
        lower.row(-1).copy_(upper.row(h - 1))
        upper.row(h).copy_(lower.row(0))


def run_band(engine: BandEngine, topo: BandTopology, niter: int, exchange) -> None:
    """one complete smoothing of the band: niter x {pass A, halo, pass B}
    (simple schedule: the halo exchange sits between the two passes)"""
    for it in range(niter):
        engine.idct(it == 0, topo.rep_top, topo.rep_bot)
        exchange()
        engine.smooth(it == niter - 1)


def run_band_overlapped(engine: BandEngine, topo: BandTopology, niter: int, exchange, comm=None) -> None:
    """Same result, communication hidden: only the first and last block row of a
    band read the halo rows, so pass B starts on the interior rows right after
    pass A while the exchange is in flight, and finishes with the edge rows once
    it has landed.  `comm` = (begin, end) callables that fork the exchange onto a
    side stream and join it again (HipBandEngine.comm_scope); without it the
    exchange is simply issued first (CPU engines)."""
    hb = engine.hblk
    lo = 1 if topo.up is not None else 0                    # edge rows that must wait
    hi = hb - 1 if topo.down is not None else hb
    if hi < lo:                                             # one-row band with two neighbours
        lo, hi = 0, 0
    for it in range(niter):
        last = it == niter - 1
        engine.idct(it == 0, topo.rep_top, topo.rep_bot)
        if comm:
            comm[0]()
        exchange()
        if comm:
            comm[1](False)                                  # stay forked: do not join yet
        engine.smooth_rows(lo, hi, last)                    # interior: no halo dependence
        if comm:
            comm[1](True)                                   # join: halo rows are in place
        if lo > 0:
            engine.smooth_rows(0, lo, last)
        if hi < hb:
            engine.smooth_rows(hi, hb, last)


class HipBandEngine(BandEngine):
    """band backend on the MI355X kernels; all buffers are torch device tensors"""

    def __init__(self, hip, torch, coef, quant, flags, luma=1, device=None, stream=None):
        self.hip, self.torch = hip, torch
        self.coef = coef                                   # int16 [hblk, wblk, 64] on device
        self.hblk, self.wblk = int(coef.shape[0]), int(coef.shape[1])
        self.flags, self.luma = flags, luma
        dev = device if device is not None else coef.device
        self.cst = torch.from_numpy(hip.consts_build(quant, flags)).to(dev)
        self.plane = torch.zeros(hip.plane_bytes(self.wblk, self.hblk), dtype=torch.uint8, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.pitch = hip.plane_pitch(self.wblk)
        self._stream = stream

    def _s(self):
        s = self._stream if self._stream is not None else self.torch.cuda.current_stream()
        return s.cuda_stream

    def rebind(self, coef):
        """point the engine at another resident coefficient band of the same shape"""
        assert tuple(coef.shape) == (self.hblk, self.wblk, 64)
        self.coef = coef

    def idct(self, first, rep_top, rep_bot):
        self.hip.idct_plane(self.cst.data_ptr(), self.coef.data_ptr(), self.plane.data_ptr(),
                            self.wblk, self.hblk, first, rep_top, rep_bot, self.status.data_ptr(), self._s())

    def smooth(self, final_clamp):
        self.hip.smooth_plane(self.cst.data_ptr(), self.coef.data_ptr(), self.plane.data_ptr(),
                              self.wblk, self.hblk, self.flags, self.luma, final_clamp, self._s())

    def smooth_rows(self, row0, row1, final_clamp):
        if row1 > row0:
            self.hip.smooth_rows(self.cst.data_ptr(), self.coef.data_ptr(), self.plane.data_ptr(),
                                 self.wblk, self.hblk, row0, row1, self.flags, self.luma, final_clamp, self._s())

    def row(self, y):
        o = self.hip.plane_row_offset(self.wblk, y)
        return self.plane[o:o + self.pitch]

    def bad_coef(self):
        return bool(int(self.status.item()))

    def comm_scope(self):
        """(begin, end) for run_band_overlapped: run the halo exchange on a side
        stream that waits for pass A, and make the compute stream wait for it
        only before the edge rows"""
        torch = self.torch
        if not hasattr(self, "_comm_stream"):
            self._comm_stream = torch.cuda.Stream(device=self.plane.device)
            self._ctx = None
        main = self._stream if self._stream is not None else torch.cuda.current_stream()

        def begin():
            self._comm_stream.wait_stream(main)            # after pass A
            self._ctx = torch.cuda.stream(self._comm_stream)
            self._ctx.__enter__()                          # collectives issued now sync with the side stream

        def end(join):
            if self._ctx is not None:
                self._ctx.__exit__(None, None, None)
                self._ctx = None
            if join:
                main.wait_stream(self._comm_stream)
        return begin, end
