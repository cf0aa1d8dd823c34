#!/usr/bin/env python3
"""Debug aid (GPU box): check kernel A's plane and kernel B with/without rebalance."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import jpegqs_pkg
from oracle.oracle import Oracle
pkg = jpegqs_pkg.load(); hip = pkg.HipQS(); O = Oracle()
dev = torch.device("cuda:0")
coef, quant = pkg.synth.synth_gray(64, 64, 50)
hb, wb = coef.shape[:2]
d_coef = torch.from_numpy(coef.copy()).to(dev)
d_cst = torch.from_numpy(hip.consts_build(quant, 0)).to(dev)
d_plane = torch.zeros(hip.plane_bytes(wb, hb), dtype=torch.uint8, device=dev)
d_status = torch.zeros(1, dtype=torch.int32, device=dev)
hip.idct_plane(d_cst.data_ptr(), d_coef.data_ptr(), d_plane.data_ptr(), wb, hb, 1, 1, 1, d_status.data_ptr(), None)
torch.cuda.synchronize()
deq = d_coef.cpu().numpy()
want = (coef.astype(np.int32) * quant.astype(np.int32)).astype(np.int16)
print("dequant ok:", np.array_equal(deq, want), "status", int(d_status.item()))
pitch = hip.plane_pitch(wb)
pl = d_plane.cpu().numpy()[: pitch * (hb * 8 + 2)].reshape(hb * 8 + 2, pitch)
img = np.zeros((hb * 8, wb * 8), np.uint8)
for by in range(hb):
    for bx in range(wb):
        img[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] = O.idct_islow(want[by, bx]).reshape(8, 8)
got = pl[1:1 + hb * 8, 16:16 + wb * 8]
print("plane interior ok:", np.array_equal(got, img), "mismatch px", int((got != img).sum()))
pad = np.pad(img, 1, mode="edge")
gotp = pl[0:hb * 8 + 2, 15:15 + wb * 8 + 2]
print("plane with apron ok:", np.array_equal(gotp, pad), int((gotp != pad).sum()))
if not np.array_equal(got, img):
    bad = np.argwhere(got != img)[:10]
    for y, x in bad: print("  px", y, x, got[y, x], img[y, x])
for flags in (16, 0):
    c = torch.from_numpy(want.copy()).to(dev)
    hip.smooth_plane(d_cst.data_ptr(), c.data_ptr(), d_plane.data_ptr(), wb, hb, flags, 1, 0, None)
    torch.cuda.synchronize()
    a = c.cpu().numpy()
    # oracle: one iteration, same flags, no final clamp difference expected
    b = O.do_quantsmooth([coef], [quant], flags, 1)["coefs"][0]
    bad = np.argwhere(a != b)
    print(f"smooth flags={flags}: {len(bad)} mismatches; first:", [(int(y), int(x), int(i), int(a[y, x, i]), int(b[y, x, i])) for y, x, i in bad[:12]])
    # which zigzag positions mismatch most
    if len(bad):
        zz = pkg.hipqs  # noqa
        from collections import Counter
        print("   by coef index:", sorted(Counter(int(i) for _, _, i in bad).items())[:64])
