#!/usr/bin/env python3
"""How often could the refresh IDCT at an anti-diagonal start be skipped (reference quantsmooth.h:1407-1409:
"if need_refresh") -- per block, and per WAVE of 64 consecutive blocks (what the GPU kernel can exploit,
LABNOTES.md 4.2c)?  CPU only: builds an instrumented copy of the test oracle under /tmp (a per-block bit mask of
the anti-diagonals whose refresh was needed) and runs one iteration on several inputs.
    python tools/refresh_skip_rates.py > profiles/r03_info/refresh_skip_rates.txt"""
import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import jpegqs_pkg  # noqa: E402
from oracle import oracle as om  # noqa: E402

src = (ROOT / "oracle" / "qs_oracle.c").read_text()
src = src.replace("long long qso_stat_groups, qso_stat_refreshes;",
                  "long long qso_stat_groups, qso_stat_refreshes; unsigned short *qso_mask; long qso_cur; int qso_gidx;")
src = src.replace("if (starts_antidiagonal(k)) { qso_stat_groups++; if (stale) qso_stat_refreshes++; }",
                  "if (k == 63) qso_gidx = 0; if (starts_antidiagonal(k)) { qso_stat_groups++; if (stale) { qso_stat_refreshes++; "
                  "if (qso_mask) qso_mask[qso_cur] |= 1u << qso_gidx; } qso_gidx++; }")
src = src.replace("\t\t\t\t\tqso_block(coefs + ((size_t)by * wb + bx) * 64, q,",
                  "\t\t\t\t\tqso_cur = (long)by * wb + bx;\n\t\t\t\t\tqso_block(coefs + ((size_t)by * wb + bx) * 64, q,")
assert "qso_mask" in src and "qso_cur = " in src
Path("/tmp/qso_mask.c").write_text(src)
subprocess.run(["cp", str(ROOT / "oracle" / "qs_oracle.h"), "/tmp/"], check=True)
subprocess.run(["gcc", "-O2", "-DQSO_STATS", "-ffp-contract=off", "-fPIC", "-shared", "-o", "/tmp/libqso_mask.so", "/tmp/qso_mask.c", "-lm"], check=True)
O = om.Oracle(path="/tmp/libqso_mask.so")
maskp = C.c_void_p.in_dll(O.lib, "qso_mask")
S = jpegqs_pkg.load().synth


def run(name, coef, quant, flags=0):
    nb = coef.shape[0] * coef.shape[1]
    m = np.zeros(nb, np.uint16)
    maskp.value = m.ctypes.data
    O.do_quantsmooth([coef], [quant], flags, 1, threads=1)
    maskp.value = None
    bits = np.array([(m >> g) & 1 for g in range(14)])
    w = bits[:, :nb // 64 * 64].reshape(14, -1, 64).max(axis=2)
    print(f"{name:58s} refreshes needed: per block {bits.mean():.3f}   per wave of 64 blocks {w.mean():.3f}")
    print("      per anti-diagonal, block:", " ".join(f"{v:.2f}" for v in bits.mean(axis=1)))
    print("      per anti-diagonal, wave :", " ".join(f"{v:.2f}" for v in w.mean(axis=1)))


def pix(sigma, k=1, w=2048, h=512, seed=1234):
    rng = np.random.default_rng(seed)
    x = np.arange(w, dtype=np.float32)[None, :]; y = np.arange(h, dtype=np.float32)[:, None]
    img = 128.0 + 60.0 * np.sin(x / (17.0 * k)) + 50.0 * np.cos(y / (23.0 * k))
    checker = (((np.arange(w) // (37 * k))[None, :] + (np.arange(h) // (29 * k))[:, None]) & 1).astype(np.float32)
    img = img + 40.0 * (checker - 0.5)
    if sigma:
        img = img + rng.normal(0.0, sigma, size=(h, w)).astype(np.float32)
    img[h // 5:h // 5 + h // 7, w // 3:w // 3 + w // 4] *= 0.45
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


print("# one iteration, --quality 3, 2048 x 512 pixels; 1.000 = every refresh needed (nothing to skip)")
for q in (50, 85):
    qt = S.quality_table(S.STD_LUMA, q)
    for sigma in (6, 2, 0):
        run(f"SURVEY 8d formula, noise sigma {sigma}, JPEG q{q}" + ("  (= the headline input)" if sigma == 6 and q == 50 else ""),
            S.quantise_plane(pix(sigma), qt), qt)
    run(f"periods x10, no noise, JPEG q{q}  (= bench.py --input smooth)", S.quantise_plane(pix(0, 10), qt), qt)
