#!/usr/bin/env python3
"""End to end through the drop-in CLI: a real libjpeg-encoded JPEG in, a JPEG out, `--info 8` timing of
do_quantsmooth itself (reference quantsmooth.h:2820-2825) and the wall time of the whole process, for
  * our `jpegqs` (every run is a FRESH process: HIP start-up, code-object load and cold pools included),
  * the reference's own CLI built as its Makefile does with SIMD=avx512 / SIMD=avx2 (oracle/_ref/jpegqs_ref_avx512,
    _avx2; OpenMP, all usable cores) -- the fast CPU path a user would otherwise run,
  * the reference's scalar build (oracle/_ref/jpegqs_ref_none, OpenMP) -- the PARITY gate: our output must be
    byte-identical to it (the SIMD builds sum in other orders; their size is printed for information).
    python tools/bench_cli.py [size=8192] [repeats=3]"""
import os
import re
import subprocess
import sys
import time
from pathlib import Path

from PIL import Image

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import jpegqs_pkg  # noqa: E402

Image.MAX_IMAGE_PIXELS = None
pkg = jpegqs_pkg.load()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 3
out = Path("/tmp/qs_cli"); out.mkdir(exist_ok=True)
ours = ROOT / "jpeg-quantsmooth_amd" / "jpegqs"
refdir = ROOT / "oracle" / "_ref"
flags = open("/proc/cpuinfo").read()
exes = [("gpu", ours)]
if "avx512bw" in flags:
    exes.append(("ref avx512+openmp", refdir / "jpegqs_ref_avx512"))
if "avx2" in flags:
    exes.append(("ref avx2+openmp", refdir / "jpegqs_ref_avx2"))
exes.append(("ref scalar+openmp", refdir / "jpegqs_ref_none"))
cases = [("gray", size, size, 3), ("gray", size, size, 4), ("rgb420", 1920, 1080, 3), ("rgb420", 1920, 1080, 6), ("rgb420", size // 2, size // 2, 6)]
usable = len(os.sched_getaffinity(0))
try:
    quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
    if quota.isdigit() and int(period) > 0:
        usable = max(1, min(usable, int(quota) // int(period)))
except (OSError, ValueError):
    pass
made = {}
print(f"# reference CLIs run with --threads {usable}")
print(f"# usable cpus: {len(os.sched_getaffinity(0))}, cgroup cpu.max: {Path('/sys/fs/cgroup/cpu.max').read_text().strip() if Path('/sys/fs/cgroup/cpu.max').exists() else '?'}")
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
    for name, exe in exes:
        if not exe.exists():
            continue
        dst = out / f"out_{name.replace(' ', '_').replace('+', '_')}.jpg"
        qs, walls = [], []
        for rep in range(repeats if name != "ref scalar+openmp" else 1):
            t0 = time.time()
            # the reference sizes its OpenMP team by the logical CPU count (256 on this box) although the cgroup grants
            # far fewer cores: give it the usable count, as a user of that box would
            extra = ["-t", str(usable)] if name.startswith("ref") else []
            r = subprocess.run([str(exe), "-q", str(q), "-i", "8", *extra, str(src), str(dst)], capture_output=True, text=True)
            walls.append(time.time() - t0)
            m = re.search(r"quantsmooth: ([0-9.]+)ms", r.stderr)
            qs.append(float(m.group(1)) if m else float("nan"))
        res[name] = (qs, walls, dst.read_bytes() if dst.exists() else b"", r.returncode)
    print(f"{kind} {w}x{h} --quality {q}:")
    for n, (qs, walls, data, rc) in res.items():
        print(f"    {n:20s} do_quantsmooth {' / '.join(f'{v:.1f}' for v in qs)} ms   whole process {' / '.join(f'{v:.2f}' for v in walls)} s   (rc {rc}, {len(data)} bytes)")
    if "gpu" in res and "ref scalar+openmp" in res:
        a, b = res["gpu"][2], res["ref scalar+openmp"][2]
        print(f"    gpu output vs scalar reference output: {'IDENTICAL' if a == b and a else 'DIFFER'}", flush=True)
