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

import os
from dataclasses import dataclass


_LIB = None


def _lib():
    """the product library: the band arithmetic below is ITS definition (csrc/qs_planes.cpp:
    qs_hip_band_rows / qs_hip_colour_band_rows / qs_hip_band_halo_rows), shared with the in-process
    multi-GPU route csrc/qs_shard.cpp -- this driver only adds the transport (torch.distributed).
    None when the library cannot be loaded (a CPU-only box without the HIP runtime): the two pure
    integer functions then fall back to the Python restatement below, which tests/test_bands.py
    pins to the C definition value for value wherever the library does load."""
    global _LIB
    if _LIB is None:
        try:
            from .hipqs import HipQS
            _LIB = HipQS()
        except (OSError, RuntimeError, ImportError):
            _LIB = False
    return _LIB or None


def _band_rows_py(hblk: int, world: int, rank: int, align: int = 1):
    """qs_hip_band_rows restated (integer arithmetic only; no compute)"""
    if hblk < 0 or world < 1 or not 0 <= rank < world or align < 1:
        raise ValueError("band_rows: bad argument")
    units = (hblk + align - 1) // align
    a, b = units * rank // world * align, units * (rank + 1) // world * align
    return min(a, hblk), min(b, hblk)


def _colour_band_rows_py(hblk_y: int, hblk_c: int, vs: int, world: int, rank: int):
    """qs_hip_colour_band_rows restated"""
    if vs < 1:
        raise ValueError("colour_band_rows: bad argument")
    c0, c1 = _band_rows_py(hblk_c, world, rank, 1)
    y0 = min(c0 * vs, hblk_y)
    y1 = hblk_y if rank == world - 1 else min(c1 * vs, hblk_y)
    return y0, y1, c0, c1


def band_rows(hblk: int, world: int, rank: int, align: int = 1):
    """block rows [r0, r1) owned by `rank`; band edges fall on multiples of
    `align` block rows (2 for the luma plane of a 4:2:0 image so that chroma
    bands line up with luma bands).  C: qs_hip_band_rows."""
    lib = _lib()
    return lib.band_rows(hblk, world, rank, align) if lib else _band_rows_py(hblk, world, rank, align)


def deep_band_rows(hblk: int, world: int, rank: int, niter: int, align: int = 1):
    """The COMMUNICATION-AVOIDING schedule (csrc/qs_shard.cpp: qs_hip_set_shard_schedule(1)): -> (r0, r1, e0, e1), the rows
    [r0, r1) the rank owns and the rows [e0, e1) it holds and runs -- `niter` more block rows on every cut side (none at
    the image edges).  Treating the cuts as image edges makes an error that travels one block row per iteration (a block
    reads one pixel row across its border, reference quantsmooth.h:1396-1401) and therefore stops short of [r0, r1):
    all iterations run without a single exchange.  Cost: 2 * niter extra block rows per inner band."""
    r0, r1 = band_rows(hblk, world, rank, align)
    return r0, r1, max(0, r0 - niter), min(hblk, r1 + niter)


def run_band_deep(engine, niter: int) -> None:
    """one complete smoothing of a deep-halo band (an engine built on rows [e0, e1) of deep_band_rows): both plane edges
    replicate like image edges, nothing is exchanged; the caller keeps rows [r0 - e0, r1 - e0) of the result"""
    if hasattr(engine, "smooth_next"):
        engine.idct(True, 1, 1)
        for it in range(niter):
            engine.smooth_next(it == niter - 1, it < niter - 1, 1, 1)
        return
    for it in range(niter):
        engine.idct(it == 0, 1, 1)
        engine.smooth(it == niter - 1)


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


def exchange_halo_dist_hostcopy(engine: BandEngine, topo: BandTopology, dist) -> None:
    """the same exchange staged through host memory: for back ends without
    device-buffer p2p (gloo), i.e. functional tests of the multi-process path
    on a box where RCCL cannot run (one GPU)"""
    import torch
    h = engine.hblk * 8
    ops, recvs = [], []
    for nbr, src_y, dst_y in ((topo.up, 0, -1), (topo.down, h - 1, h)):
        if nbr is None:
            continue
        out = engine.row(src_y).to("cpu")               # stream-ordered after pass A (blocking copy)
        buf = torch.empty_like(out)
        ops.append(dist.P2POp(dist.isend, out, nbr))
        ops.append(dist.P2POp(dist.irecv, buf, nbr))
        recvs.append((dst_y, buf))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for dst_y, buf in recvs:
        engine.row(dst_y).copy_(buf)


def exchange_halo_dist_many(engines, topo: BandTopology, dist) -> None:
    """the halo rows of SEVERAL independent planes (a batch: one engine per plane, all cut into
    the same bands) in ONE batched send/recv: the exchange is latency-bound (8 KB rows), so a
    step over a batch of planes pays for it once per iteration instead of once per plane"""
    ops = []
    for engine in engines:
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


def exchange_halo_dist_many_hostcopy(engines, topo: BandTopology, dist) -> None:
    """exchange_halo_dist_many staged through host memory (gloo: functional tests on one GPU)"""
    import torch
    ops, recvs = [], []
    for engine in engines:
        h = engine.hblk * 8
        for nbr, src_y, dst_y in ((topo.up, 0, -1), (topo.down, h - 1, h)):
            if nbr is None:
                continue
            out = engine.row(src_y).to("cpu")
            buf = torch.empty_like(out)
            ops.append(dist.P2POp(dist.isend, out, nbr))
            ops.append(dist.P2POp(dist.irecv, buf, nbr))
            recvs.append((engine, dst_y, buf))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for engine, dst_y, buf in recvs:
        engine.row(dst_y).copy_(buf)


def exchange_halo_packed(hip, planes2d, wblk: int, hblk: int, topo: BandTopology, dist, hostcopy: bool = False) -> None:
    """The halo exchange of a whole batch with ONE send and ONE receive per neighbour: `planes2d` is the
    [batch, plane_bytes] tensor whose rows are the engines' planes (same band geometry), so pixel row y of
    every plane is the strided view planes2d[:, off(y) : off(y) + pitch]; it is packed into a contiguous
    [batch, pitch] buffer by one copy kernel, sent, and the received buffer lands in the apron rows by one
    strided copy.  hostcopy: stage through host memory (gloo)."""
    send_top, send_bot, recv_top, recv_bot, pitch = hip.band_halo_rows(wblk, hblk)   # the C definition
    h = hblk * 8
    offs = {0: send_top, h - 1: send_bot, -1: recv_top, h: recv_bot}

    def view(y):
        o = offs[y]
        return planes2d[:, o:o + pitch]
    ops, recvs = [], []
    for nbr, src_y, dst_y in ((topo.up, 0, -1), (topo.down, h - 1, h)):
        if nbr is None:
            continue
        out = view(src_y).contiguous()
        if hostcopy:
            out = out.to("cpu")
        buf = out.new_empty(out.shape)
        ops.append(dist.P2POp(dist.isend, out, nbr))
        ops.append(dist.P2POp(dist.irecv, buf, nbr))
        recvs.append((dst_y, buf))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for dst_y, buf in recvs:
        view(dst_y).copy_(buf)


def run_bands_batched(engines, topo: BandTopology, niter: int, exchange_many) -> None:
    """one complete smoothing of the same band of several independent planes, iteration by
    iteration for all of them: niter x {pass A of every plane, ONE halo exchange, pass B of every plane}"""
    for it in range(niter):
        for e in engines:
            e.idct(it == 0, topo.rep_top, topo.rep_bot)
        exchange_many()
        for e in engines:
            e.smooth(it == niter - 1)


def run_bands_batched_sets(hip, engines, topo: BandTopology, niter: int, exchange_many, stream=None, mark=None, fused=True) -> None:
    """run_bands_batched with ONE launch per pass for all planes of the batch (plane sets,
    qs_hip_idct_planes / qs_hip_smooth_planes): a 1/8 band of an 8192^2 plane is 2048 waves, two per
    SIMD -- launched alone it runs at 72 % of the rate the same kernel reaches once the chip is full
    (LABNOTES.md 4.2b table); twelve bands in one launch fill it.  HipBandEngine objects only.

    fused (default): pass A runs ONCE; every pass B but the last writes the next iteration's pixel planes itself
    (the parallel d_plane_next[] array of qs_hip_smooth_planes_next) into each engine's second plane, and the engines' `plane` / `plane2` swap --
    `engine.plane` (and `engine.row()`) is always the plane the coming pass B reads, which is what
    `exchange_many()` must exchange the halo rows of.  Per iteration: [halo rows], ONE launch."""
    band = (1 if topo.up is not None else 0) | (2 if topo.down is not None else 0)
    flags = engines[0].flags
    s = stream if stream is not None else engines[0]._s()
    if fused:
        for e in engines:
            e.ensure_plane2()
    for it in range(niter):
        if it == 0 or not fused:
            refs = hip.plane_refs([(e.cst.data_ptr(), e.coef.data_ptr(), e.plane.data_ptr(), e.status.data_ptr(),
                                    e.wblk, e.hblk, e.luma, band) for e in engines])
            hip.idct_planes(refs, it == 0, s)
        exchange_many()
        nxt = fused and it < niter - 1
        refs = hip.plane_refs([(e.cst.data_ptr(), e.coef.data_ptr(), e.plane.data_ptr(), e.status.data_ptr(),
                                e.wblk, e.hblk, e.luma, band, e.plane2.data_ptr() if nxt else None) for e in engines])
        if mark:
            mark(0)                                  # (bench.py: HIP events around the pass-B launch)
        hip.smooth_planes(refs, flags, it == niter - 1, s)
        if mark:
            mark(1)
        if nxt:
            for e in engines:
                e.plane, e.plane2 = e.plane2, e.plane


def run_band_edge_first(hip, e, topo: BandTopology, niter: int, exchange, main, side, torch) -> None:
    """ONE plane (a single image's band), latency-hiding schedule on the fused kernels (round 6).  Only the first and the last
    block row of a band read the halo rows, and only they PRODUCE the rows the neighbours need.  Per iteration:
        side stream:  pass B of block rows {0, hblk-1} as ONE two-plane set launch (32 groups at 8192 px: the small-plane
                      kernel, ~80 us), then the halo exchange for the NEXT iteration's planes;
        main stream:  pass B of the interior rows [1, hblk-1) (reads no halo row) -- the exchange hides behind it.
    The three launch regions are VIEWS of the band's coefficient array and pixel planes (plane pointer + 8 * r * pitch: a
    view's apron rows are the neighbouring region's real pixel rows, which is exactly what a block reads across its border),
    so nothing is copied and the kernels are the unchanged plane-set kernels.  Iteration n + 1 starts when both kernels of
    iteration n are done (events both ways); the exchange is ordered on the side stream behind the edge kernel that produced
    its rows.  `exchange()` is called with the side stream current and must move the halo rows of `e.plane` (the plane the
    coming pass B reads).  Bit-exact with run_bands_batched_sets; falls back to it for bands of fewer than 3 block rows."""
    if e.hblk < 3:
        return run_bands_batched_sets(hip, [e], topo, niter, exchange, stream=main.cuda_stream)
    e.ensure_plane2()
    band_top = 1 if topo.up is not None else 0          # the band's own top / bottom apron row is a halo row
    band_bot = 2 if topo.down is not None else 0
    pitch, wb, hb = e.pitch, e.wblk, e.hblk
    row_bytes = wb * 128

    def views(cur, nxt):
        """(edge set, interior set) for planes cur -> nxt (nxt None on the last iteration)"""
        c0, p0 = e.coef.data_ptr(), cur.data_ptr()
        n0 = nxt.data_ptr() if nxt is not None else None
        at = lambda base, r: base + r * 8 * pitch if base is not None else None
        top = (e.cst.data_ptr(), c0, p0, e.status.data_ptr(), wb, 1, e.luma, band_top | 2, n0)
        bot = (e.cst.data_ptr(), c0 + (hb - 1) * row_bytes, at(p0, hb - 1), e.status.data_ptr(), wb, 1, e.luma, 1 | band_bot, at(n0, hb - 1))
        mid = (e.cst.data_ptr(), c0 + row_bytes, at(p0, 1), e.status.data_ptr(), wb, hb - 2, e.luma, 3, at(n0, 1))
        return hip.plane_refs([top, bot]), hip.plane_refs([mid])
    ev_main, ev_edge = torch.cuda.Event(), torch.cuda.Event()
    whole = hip.plane_refs([(e.cst.data_ptr(), e.coef.data_ptr(), e.plane.data_ptr(), e.status.data_ptr(), wb, hb, e.luma, band_top | band_bot)])
    hip.idct_planes(whole, True, main.cuda_stream)
    ev_main.record(main)
    side.wait_event(ev_main)
    with torch.cuda.stream(side):
        exchange()                                       # halo rows of the first planes
    for it in range(niter):
        last = it == niter - 1
        edge, mid = views(e.plane, None if last else e.plane2)
        # edge rows: behind the previous interior launch (their neighbours' pixels) and, in stream order, the exchange
        if it:
            side.wait_event(ev_main)
        hip.smooth_planes(edge, e.flags, last, side.cuda_stream)
        ev_edge_now = torch.cuda.Event(); ev_edge_now.record(side)
        # interior rows: behind the previous EDGE launch (rows 0 and hblk-1 are their neighbours)
        if it:
            main.wait_event(ev_edge)
        hip.smooth_planes(mid, e.flags, last, main.cuda_stream)
        ev_main = torch.cuda.Event(); ev_main.record(main)
        ev_edge = ev_edge_now
        if not last:
            e.plane, e.plane2 = e.plane2, e.plane
            with torch.cuda.stream(side):
                exchange()                               # rows of the new current plane, produced by the edge launch just queued
    main.wait_event(ev_edge)


def run_bands_batched_fused(engines, topo: BandTopology, niter: int, exchange_many) -> None:
    """The fused schedule for ANY engine that offers `smooth_next(final_clamp, write_next, rep_top, rep_bot)` (pass B that
    also writes the next iteration's pixel plane into the engine's second plane and swaps the two): pass A once, then
    niter x {ONE halo exchange for the planes the coming pass B reads, pass B of every plane}.  One launch per plane and
    pass (run_bands_batched_sets is the one-launch-per-pass form for HipBandEngine); the CPU engine of the tests runs the
    same loop under gloo."""
    for e in engines:
        e.idct(True, topo.rep_top, topo.rep_bot)
    for it in range(niter):
        exchange_many()
        for e in engines:
            e.smooth_next(it == niter - 1, it < niter - 1, topo.rep_top, topo.rep_bot)


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
    """Same result, communication hidden.  Only the first and last block row of a
    band read the halo rows, so after pass A two things proceed side by side:
      * main stream: pass B on the interior rows;
      * side stream: the halo exchange, then pass B on the two edge rows (a
        one-row launch is latency-bound -- a single wave needs ~0.2 ms for its
        63 coefficient steps -- so it must not be serialised behind the interior).
    `comm` = HipBandEngine.comm_scope(): (fork, side, join) callables; without it
    (CPU engines) the same steps simply run in program order."""
    hb = engine.hblk
    lo = 1 if topo.up is not None else 0                    # edge rows that must wait
    hi = hb - 1 if topo.down is not None else hb
    if hi < lo:                                             # one-row band with two neighbours
        lo, hi = 0, 0
    for it in range(niter):
        last = it == niter - 1
        engine.idct(it == 0, topo.rep_top, topo.rep_bot)
        if comm:
            comm[0]()                                       # mark "pass A done" on the main stream
        engine.smooth_rows(lo, hi, last)                    # interior: no halo dependence

        def edges():
            exchange()
            if lo > 0:
                engine.smooth_rows(0, lo, last)
            if hi < hb:
                engine.smooth_rows(hi, hb, last)
        if comm:
            comm[1](edges)                                  # on the side stream, after the mark
            comm[2]()                                       # main waits for the side stream
        else:
            edges()


class HipBandEngine(BandEngine):
    """band backend on the MI355X kernels; all buffers are torch device tensors"""

    def __init__(self, hip, torch, coef, quant, flags, luma=1, device=None, stream=None, plane=None, plane2=None):
        self.hip, self.torch = hip, torch
        self.coef = coef                                   # int16 [hblk, wblk, 64] on device
        self.hblk, self.wblk = int(coef.shape[0]), int(coef.shape[1])
        self.flags, self.luma = flags, luma
        dev = device if device is not None else coef.device
        self.cst = torch.from_numpy(hip.consts_build(quant, flags)).to(dev)
        # plane: a caller-owned uint8 tensor of plane_bytes (e.g. one row of a [batch, plane_bytes] tensor, so
        # that the halo rows of a whole batch can be packed with one strided copy: exchange_halo_packed)
        self.plane = plane if plane is not None else \
            torch.zeros(hip.plane_bytes(self.wblk, self.hblk), dtype=torch.uint8, device=dev)
        # the second plane of the fused schedule (run_bands_batched_sets): pass B writes the next iteration's pixels
        # there; allocated on first use unless the caller owns it (as for `plane`)
        self.plane2 = plane2
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.pitch = hip.plane_pitch(self.wblk)
        self._stream = stream

    def ensure_plane2(self):
        if self.plane2 is None:
            self.plane2 = self.torch.zeros_like(self.plane)

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

    def smooth_next(self, final_clamp, write_next, rep_top=1, rep_bot=1):
        """pass B; write_next: it also writes the next iteration's pixel plane (fused pass A) into plane2, and the two swap"""
        if not write_next:
            return self.smooth(final_clamp)
        self.ensure_plane2()
        self.hip.smooth_plane_next(self.cst.data_ptr(), self.coef.data_ptr(), self.plane.data_ptr(), self.plane2.data_ptr(),
                                   self.wblk, self.hblk, self.flags, self.luma, final_clamp, rep_top, rep_bot, self._s())
        self.plane, self.plane2 = self.plane2, self.plane

    def smooth_rows(self, row0, row1, final_clamp):
        if row1 > row0:
            self.hip.smooth_rows(self.cst.data_ptr(), self.coef.data_ptr(), self.plane.data_ptr(),
                                 self.wblk, self.hblk, row0, row1, self.flags, self.luma, final_clamp, self._s())

    def row(self, y):
        """the four rows of the halo exchange come from the C definition (qs_hip_band_halo_rows); any
        other row (tests) from the plane geometry"""
        h = self.hblk * 8
        if y in (0, h - 1, -1, h):
            st, sb, rt, rb, n = self.hip.band_halo_rows(self.wblk, self.hblk)
            o = {0: st, h - 1: sb, -1: rt, h: rb}[y]          # (hblk * 8 - 1 == 0 cannot happen: a band has >= 8 pixel rows)
            return self.plane[o:o + n]
        o = self.hip.plane_row_offset(self.wblk, y)
        return self.plane[o:o + self.pitch]

    def bad_coef(self):
        return bool(int(self.status.item()))

    def comm_scope(self):
        """(fork, side, join) for run_band_overlapped"""
        torch = self.torch
        if not hasattr(self, "_comm_stream"):
            self._comm_stream = torch.cuda.Stream(device=self.plane.device)
            self._mark = torch.cuda.Event()
        main = self._stream if self._stream is not None else torch.cuda.current_stream()

        def fork():
            self._mark.record(main)                        # pass A is complete at this point of the main stream

        def side(fn):
            self._comm_stream.wait_event(self._mark)
            with torch.cuda.stream(self._comm_stream):     # kernels and collectives issued by fn go to the side stream
                fn()

        def join():
            main.wait_stream(self._comm_stream)
        return fork, side, join


# ---------------------------------------------------------------------------
# YCbCr jobs with JOINT_YUV / UPSAMPLE_UV, sharded in row bands
# (BASELINE config 4: 8192x8192 4:2:0, q=6, cross-component dependence)
#
# Chroma depends on the FINAL luma only (through the low-res luma plane L and,
# for the upsample, the full-res luma plane; reference quantsmooth.h:2753-2815),
# Cb and Cr are independent of each other.  Bands are cut on chroma block rows;
# the luma band is the `vs`-times taller range of the same image rows, so the
# downsample and the upsample of a band touch only that band's luma rows.  What
# crosses a band edge: one pixel row of the component's own plane per
# iteration (as in the luma-only case), one row of L once, and one row of the
# refreshed chroma plane once (the 3x3 regression windows of the predictor and
# of the upsample).

class PlaneRows:
    """row accessor over any plane-geometry tensor (for the halo exchange)"""

    def __init__(self, hip, tensor, wblk, hblk):
        self.hip, self.t, self.wblk, self.hblk = hip, tensor, wblk, hblk
        self.pitch = hip.plane_pitch(wblk)

    def row(self, y):
        h = self.hblk * 8
        if y in (0, h - 1, -1, h):
            st, sb, rt, rb, n = self.hip.band_halo_rows(self.wblk, self.hblk)
            o = {0: st, h - 1: sb, -1: rt, h: rb}[y]
            return self.t[o:o + n]
        o = self.hip.plane_row_offset(self.wblk, y)
        return self.t[o:o + self.pitch]


class ColourBand:
    """one rank's band of a 3-component YCbCr image, all buffers on the device"""

    def __init__(self, hip, torch, coefs, quants, hsamp, vsamp, image_size, flags, niter, topo, device):
        self.hip, self.torch, self.flags, self.niter, self.topo = hip, torch, flags, max(0, min(int(niter), 100)), topo
        self.ws, self.hs = hsamp[0], vsamp[0]
        self.image_w, self.image_h = image_size
        self.dev = device
        self.eng = [HipBandEngine(hip, torch, c, q, flags & 0x31, luma=int(ci == 0), device=device)
                    for ci, (c, q) in enumerate(zip(coefs, quants))]
        self.joint = bool(flags & 2)
        self.upsample = bool(flags & 4) and (self.ws, self.hs) != (1, 1)
        self.lowq = bool(flags & 8)
        ce = self.eng[1]
        self.L = None if (self.ws, self.hs) == (1, 1) else \
            torch.zeros(hip.plane_bytes(ce.wblk, ce.hblk), dtype=torch.uint8, device=device)
        self.up = [None, None]

    # ---- helpers ---------------------------------------------------------
    def _s(self):
        return self.torch.cuda.current_stream().cuda_stream

    def _rebalance(self, ci):
        luma = ci == 0
        return int(not (self.flags & 16) and (luma or not (self.flags & 32)))

    def lowres_plane(self):
        return self.eng[0].plane if self.L is None else self.L

    def planes_for_halo(self, which):
        e = self.eng[which] if isinstance(which, int) else None
        if which == "L":
            ce = self.eng[1]
            return PlaneRows(self.hip, self.L, ce.wblk, ce.hblk)
        return PlaneRows(self.hip, e.plane, e.wblk, e.hblk)

    # ---- phases (the driver interleaves them with halo exchanges) ----------
    def pass_a(self, ci, first):
        self.eng[ci].idct(first, self.topo.rep_top, self.topo.rep_bot)

    def pass_b(self, ci, last):
        e = self.eng[ci]
        s = self._s()
        joint = ci > 0 and self.joint
        if self.lowq:
            if joint:
                self.hip.joint_plane(e.cst.data_ptr(), e.coef.data_ptr(), e.plane.data_ptr(),
                                     self.lowres_plane().data_ptr(), e.wblk, e.hblk, self._rebalance(ci), last, s)
            else:
                self.hip.lowq_plane(e.cst.data_ptr(), e.coef.data_ptr(), e.plane.data_ptr(), e.wblk, e.hblk,
                                    self._rebalance(ci), last, s)
            return
        if joint:
            self.hip.joint_plane(e.cst.data_ptr(), e.coef.data_ptr(), e.plane.data_ptr(),
                                 self.lowres_plane().data_ptr(), e.wblk, e.hblk, 0, 0, s)
        e.smooth(last)

    def pass_b_next(self, ci, final_clamp, write_next):
        """pass B of the recovery route (not LOW_QUALITY) that also writes the next pass A's output (fused, see
        HipBandEngine.smooth_next): the JOINT_YUV step first, on the plane this iteration reads"""
        e = self.eng[ci]
        if ci > 0 and self.joint:
            self.hip.joint_plane(e.cst.data_ptr(), e.coef.data_ptr(), e.plane.data_ptr(),
                                 self.lowres_plane().data_ptr(), e.wblk, e.hblk, 0, 0, self._s())
        e.smooth_next(final_clamp, write_next, self.topo.rep_top, self.topo.rep_bot)

    def pass_b_next_chroma(self, final_clamp, write_next):
        """pass_b_next for Cb AND Cr at once: the two chroma planes are independent of each other (each depends on the final
        luma only), so after their JOINT_YUV steps they go through the recovery kernel as ONE two-plane set launch
        (qs_hip_smooth_planes_next) -- a 262 k-block chroma plane alone is 1.33 rounds of the chip's resident waves, two
        together 2.67: less of the launch is spent in a partly filled last round.  Same arithmetic per block."""
        hip, s = self.hip, self._s()
        band = (0 if self.topo.rep_top else 1) | (0 if self.topo.rep_bot else 2)
        planes = []
        for ci in (1, 2):
            e = self.eng[ci]
            if self.joint:
                hip.joint_plane(e.cst.data_ptr(), e.coef.data_ptr(), e.plane.data_ptr(),
                                self.lowres_plane().data_ptr(), e.wblk, e.hblk, 0, 0, s)
            if write_next:
                e.ensure_plane2()
            planes.append((e.cst.data_ptr(), e.coef.data_ptr(), e.plane.data_ptr(), e.status.data_ptr(), e.wblk, e.hblk, e.luma,
                           band, e.plane2.data_ptr() if write_next else None))
        hip.smooth_planes(hip.plane_refs(planes), self.eng[1].flags, final_clamp, s)
        if write_next:
            for ci in (1, 2):
                e = self.eng[ci]
                e.plane, e.plane2 = e.plane2, e.plane

    def clamp(self, ci):
        e = self.eng[ci]
        self.hip.clamp_plane(e.coef.data_ptr(), e.wblk, e.hblk, self._s())

    def downsample(self):
        y, c = self.eng[0], self.eng[1]
        self.hip.downsample_plane(y.plane.data_ptr(), y.wblk, y.hblk, self.L.data_ptr(), c.wblk, c.hblk,
                                  self.ws, self.hs, self._s())

    def upsample_chroma(self, ci, chroma_row0):
        """chroma band -> coefficients at luma resolution (this band's luma rows)"""
        torch, hip = self.torch, self.hip
        y, c = self.eng[0], self.eng[ci]
        w1 = (self.image_w + self.ws - 1) // self.ws
        h1_img = (self.image_h + self.hs - 1) // self.hs
        px0 = chroma_row0 * 8                                   # first low-res pixel row of the band (image coords)
        h1 = max(0, min(h1_img - px0, c.hblk * 8))              # valid low-res rows inside the band
        first_rows = max(0, min(8 - px0, h1))                   # rows of the image's first strip
        pitch = hip.upsample_pitch(self.image_w, self.ws)
        px = torch.zeros(pitch * (y.hblk * 8 + 8 * self.hs) + 64, dtype=torch.uint8, device=self.dev)
        out = torch.zeros((y.hblk, y.wblk, 64), dtype=torch.int16, device=self.dev)
        hip.upsample_rows(c.plane.data_ptr(), self.lowres_plane().data_ptr(), c.wblk, y.plane.data_ptr(),
                          y.wblk, y.hblk, px.data_ptr(), pitch, w1, h1, first_rows, self.ws, self.hs, self._s())
        hip.fdct_plane(px.data_ptr(), pitch, out.data_ptr(), y.wblk, y.hblk, self._s())
        self.up[ci - 1] = out


def run_colour_bands(bands, exchange) -> None:
    """Drive one or more ColourBand objects through the whole job in lockstep.
    `bands`: list of the bands living in this process (one per rank in a real
    run, N logical bands in the single-GPU test); `exchange(rows_list)` swaps the
    edge rows of the given PlaneRows (one per band) with the neighbours."""
    b0 = bands[0]
    niter, need_lowres = b0.niter, True
    extra_y = 1
    if niter > 0 and not b0.lowq:
        return _run_colour_bands_fused(bands, exchange)
    # ---- luma: iterations + the extra refresh that feeds the chroma passes
    for it in range(niter + extra_y):
        for b in bands:
            b.pass_a(0, it == 0)
        last_refresh = it == niter
        if not last_refresh or b0.L is None:
            exchange([b.planes_for_halo(0) for b in bands])    # 1x1 luma: L is this plane, its aprons are read
        if last_refresh:
            break
        for b in bands:
            b.pass_b(0, False)
    # the +-1023 clamp comes after the refresh pass: the planes the chroma passes read are
    # the IDCT of the unclamped luma (reference :2668-2689 sits after the loop)
    for b in bands:
        b.clamp(0)
    if b0.L is not None:
        for b in bands:
            b.downsample()
        exchange([b.planes_for_halo("L") for b in bands])
    # ---- chroma
    for ci in (1, 2):
        extra = 1 if b0.upsample else 0
        for it in range(niter + extra):
            for b in bands:
                b.pass_a(ci, it == 0)
            exchange([b.planes_for_halo(ci) for b in bands])
            if it == niter:
                break
            for b in bands:
                b.pass_b(ci, it == niter - 1 and not extra)
        if extra:                                              # as for luma: clamp after the refresh
            for b in bands:
                b.clamp(ci)
        if b0.upsample:
            for b in bands:
                b.upsample_chroma(ci, b.chroma_row0)


def _run_colour_bands_fused(bands, exchange) -> None:
    """run_colour_bands with pass A fused into pass B (niter >= 1, recovery route): pass A once per component; every pass B
    writes the plane the NEXT stage reads -- the next iteration's, or the refresh the chroma stages / the upsampling read
    -- and the +-1023 clamp rides on the last one (the fused IDCT sees the unclamped coefficients, reference :2668-2689).
    Halo rows are exchanged on the engines' current planes, at the same points of the schedule as in the unfused loop."""
    b0 = bands[0]
    niter = b0.niter
    # ---- luma: niter iterations; the last one writes the refresh that feeds the chroma passes
    for b in bands:
        b.pass_a(0, True)
    for it in range(niter):
        exchange([b.planes_for_halo(0) for b in bands])
        for b in bands:
            b.pass_b_next(0, it == niter - 1, True)
    if b0.L is None:
        exchange([b.planes_for_halo(0) for b in bands])        # 1x1 luma: L is this (refreshed) plane, its aprons are read
    else:
        for b in bands:
            b.downsample()
        exchange([b.planes_for_halo("L") for b in bands])
    # ---- chroma: Cb and Cr advance together (independent of each other; one two-plane launch per iteration and band)
    extra = bool(b0.upsample)
    # (HipBandEngine; other engines: plane by plane.  QS_BANDS_CHROMA_PAIR=0: plane by plane, for A/B runs)
    pair = hasattr(b0.eng[1], "plane2") and hasattr(b0.hip, "smooth_planes") and os.environ.get("QS_BANDS_CHROMA_PAIR", "1") != "0"
    for ci in (1, 2):
        for b in bands:
            b.pass_a(ci, True)
    for it in range(niter):
        for ci in (1, 2):
            exchange([b.planes_for_halo(ci) for b in bands])
        for b in bands:
            if pair:
                b.pass_b_next_chroma(it == niter - 1, it < niter - 1 or extra)
            else:
                for ci in (1, 2):
                    b.pass_b_next(ci, it == niter - 1, it < niter - 1 or extra)
    if extra:
        for ci in (1, 2):
            exchange([b.planes_for_halo(ci) for b in bands])    # the refreshed planes' halo: the upsampling's 3x3 windows
            for b in bands:
                b.upsample_chroma(ci, b.chroma_row0)


def exchange_rows_local(rows_list) -> None:
    for upper, lower in zip(rows_list[:-1], rows_list[1:]):
        h = upper.hblk * 8
        lower.row(-1).copy_(upper.row(h - 1))
        upper.row(h).copy_(lower.row(0))


def exchange_rows_dist(rows: PlaneRows, topo: BandTopology, dist) -> None:
    ops = []
    h = rows.hblk * 8
    if topo.up is not None:
        ops.append(dist.P2POp(dist.isend, rows.row(0), topo.up))
        ops.append(dist.P2POp(dist.irecv, rows.row(-1), topo.up))
    if topo.down is not None:
        ops.append(dist.P2POp(dist.isend, rows.row(h - 1), topo.down))
        ops.append(dist.P2POp(dist.irecv, rows.row(h), topo.down))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def exchange_rows_dist_hostcopy(rows: PlaneRows, topo: BandTopology, dist) -> None:
    """exchange_rows_dist staged through host memory (gloo has no device-buffer p2p): functional
    tests of the multi-process colour path on a box where RCCL cannot run (one GPU)"""
    import torch
    h = rows.hblk * 8
    ops, recvs = [], []
    for nbr, src_y, dst_y in ((topo.up, 0, -1), (topo.down, h - 1, h)):
        if nbr is None:
            continue
        out = rows.row(src_y).to("cpu")                 # stream-ordered after the producing kernel (blocking copy)
        buf = torch.empty_like(out)
        ops.append(dist.P2POp(dist.isend, out, nbr))
        ops.append(dist.P2POp(dist.irecv, buf, nbr))
        recvs.append((dst_y, buf))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for dst_y, buf in recvs:
        rows.row(dst_y).copy_(buf)


def colour_band_split(hblk_y, hblk_c, vs, world):
    """[(luma r0, r1, chroma r0, r1)] per rank; cut on chroma block rows.  C: qs_hip_colour_band_rows."""
    lib = _lib()
    return [lib.colour_band_rows(hblk_y, hblk_c, vs, world, r) if lib else _colour_band_rows_py(hblk_y, hblk_c, vs, world, r)
            for r in range(world)]


def run_colour_band_dist(band: ColourBand, dist) -> None:
    """one rank's share of a sharded YCbCr job (RCCL / gloo halo exchange)"""
    run_colour_bands([band], lambda rows_list: exchange_rows_dist(rows_list[0], band.topo, dist))
