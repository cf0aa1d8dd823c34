#!/usr/bin/env python3
"""End to end through the drop-in CLI: a real libjpeg-encoded JPEG in, a JPEG out, `--info 8` timing of
do_quantsmooth itself (reference quantsmooth.h:2820-2825) for our `jpegqs` and for the reference's own CLI
(oracle/_ref/jpegqs_ref_none = its scalar build with OpenMP), and a byte comparison of the two outputs.
    python tools/bench_cli.py [size=8192]"""
import re
import subprocess
import sys
import time
from pathlib import Path

import numpy as np
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import jpegqs_pkg  # noqa: E402

pkg = jpegqs_pkg.load()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
out = Path("/tmp/qs_cli"); out.mkdir(exist_ok=True)
ours = ROOT / "jpeg-quantsmooth_amd" / "jpegqs"
ref = ROOT / "oracle" / "_ref" / "jpegqs_ref_none"
cases = [("gray", size, size, 3), ("gray", size, size, 4), ("rgb420", 1920, 1080, 3), ("rgb420", 1920, 1080, 6), ("rgb420", size // 2, size // 2, 6)]
made = {}
for kind, w, h, q in cases:
    key = (kind, w, h)
    if key not in made:
        y = pkg.synth.synth_pixels(w, h)
        if kind == "gray":
            img = Image.fromarray(y, "L")
        else:
            img = Image.merge("YCbCr", [Image.fromarray(pkg.synth.synth_pixels(w, h, variant=v), "L") for v in range(3)]).convert("RGB")
        src = out / f"{kind}_{w}x{h}.jpg"
        img.save(src, quality=50, subsampling=2 if kind != "gray" else -1)
        made[key] = src
    src = made[key]
    res = {}
    for name, exe in (("gpu", ours), ("reference scalar+openmp", ref)):
        if not exe.exists():
            continue
        dst = out / f"out_{name.split()[0]}.jpg"
        t0 = time.time()
        r = subprocess.run([str(exe), "-q", str(q), "-i", "8", str(src), str(dst)], capture_output=True, text=True)
        wall = time.time() - t0
        m = re.search(r"quantsmooth: ([0-9.]+)ms", r.stderr)
        res[name] = (float(m.group(1)) if m else float("nan"), wall, dst.read_bytes() if dst.exists() else b"", r.returncode)
    line = f"{kind} {w}x{h} --quality {q}: "
    line += "  ".join(f"{n}: do_quantsmooth {v[0]:.1f} ms, whole CLI {v[1]:.2f} s (rc {v[3]})" for n, v in res.items())
    if len(res) == 2:
        a, b = res["gpu"][2], res["reference scalar+openmp"][2]
        line += f"   outputs {'IDENTICAL' if a == b and a else 'DIFFER'} ({len(a)} bytes)"
    print(line, flush=True)
