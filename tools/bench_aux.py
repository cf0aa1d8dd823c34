#!/usr/bin/env python3
"""Stand-alone launch times of the cross-component kernels (JOINT_YUV predictor, LOW_QUALITY filter, luma
downsample, upsample, re-FDCT) on the planes of an 8192x8192 4:2:0 image (chroma 512x512 blocks), against the
bytes each moves.   python tools/bench_aux.py"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import jpegqs_pkg  # noqa: E402
import bench  # noqa: E402

pkg = jpegqs_pkg.load(); hip = pkg.HipQS()
dev = torch.device("cuda:0")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
coefs, quants = bench.synth_colour_gpu(torch, pkg, size, 50, dev)
s = torch.cuda.current_stream().cuda_stream
Y, C = coefs[0], coefs[1]
yh, yw = Y.shape[:2]; ch, cw = C.shape[:2]
cstY = torch.from_numpy(hip.consts_build(quants[0], 7)).to(dev)
cstC = torch.from_numpy(hip.consts_build(quants[1], 7)).to(dev)
pY = torch.zeros(hip.plane_bytes(yw, yh), dtype=torch.uint8, device=dev)
pC = torch.zeros(hip.plane_bytes(cw, ch), dtype=torch.uint8, device=dev)
pL = torch.zeros(hip.plane_bytes(cw, ch), dtype=torch.uint8, device=dev)
st = torch.zeros(2, dtype=torch.int32, device=dev)
hip.idct_plane(cstY.data_ptr(), Y.data_ptr(), pY.data_ptr(), yw, yh, 1, 1, 1, st.data_ptr(), s)
hip.idct_plane(cstC.data_ptr(), C.data_ptr(), pC.data_ptr(), cw, ch, 1, 1, 1, st[1:].data_ptr(), s)
pitch = hip.upsample_pitch(size, 2)
px = torch.zeros(pitch * (yh * 8 + 16) + 64, dtype=torch.uint8, device=dev)
up = torch.zeros((yh, yw, 64), dtype=torch.int16, device=dev)


def timeit(name, fn, blocks, bytes_per_block):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = min(ts)
    print(f"{name:34s} {t * 1e3:8.1f} us   {blocks / t / 1e6:7.3f} G blocks/s   ~{blocks * bytes_per_block / t / 1e6:7.1f} GB/s", flush=True)


nb_c, nb_y = ch * cw, yh * yw
timeit("downsample (Y -> L)", lambda: hip.downsample_plane(pY.data_ptr(), yw, yh, pL.data_ptr(), cw, ch, 2, 2, s), nb_c, 64 * 4 + 64)
work = C.clone()
timeit("joint (chroma plane)", lambda: hip.joint_plane(cstC.data_ptr(), work.data_ptr(), pC.data_ptr(), pL.data_ptr(), cw, ch, 0, 0, s), nb_c, 256 + 200)
timeit("lowq (chroma plane)", lambda: hip.lowq_plane(cstC.data_ptr(), work.data_ptr(), pC.data_ptr(), cw, ch, 1, 0, s), nb_c, 256 + 100)
timeit("idct pass A (chroma plane)", lambda: hip.idct_plane(cstC.data_ptr(), work.data_ptr(), pC.data_ptr(), cw, ch, 0, 1, 1, st[1:].data_ptr(), s), nb_c, 192)
timeit("upsample (chroma -> luma size)", lambda: hip.upsample_rows(pC.data_ptr(), pL.data_ptr(), cw, pY.data_ptr(), yw, yh, px.data_ptr(), pitch,
                                                                  size // 2, size // 2, 8, 2, 2, s), nb_y, 64 + 64 + 32)
timeit("re-FDCT (luma-size plane)", lambda: hip.fdct_plane(px.data_ptr(), pitch, up.data_ptr(), yw, yh, s), nb_y, 64 + 128)
timeit("clamp (luma-size plane)", lambda: hip.clamp_plane(up.data_ptr(), yw, yh, s), nb_y, 256)
