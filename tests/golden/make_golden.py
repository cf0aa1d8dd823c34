"""Generate the golden vectors under tests/golden/ from the COMPILED, UNMODIFIED
reference (oracle/_ref/libqsref_none.so = `make SIMD=none` arithmetic, built by
oracle/Makefile from /root/reference).  Run in the build container only:

    python tests/golden/make_golden.py

Each .npz holds the inputs (quantised coefficient planes, quant tables, job
geometry, flags, niter) and the reference's outputs, so the vectors can be
replayed anywhere (/root/reference is not needed to USE them).
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import jpegqs_pkg  # noqa: E402
from oracle.oracle import Reference, build_ref  # noqa: E402

OUT = Path(__file__).resolve().parent


def save(name, ref, coefs, quants, flags, niter, **kw):
    res = ref.do_quantsmooth(coefs, quants, flags, niter, **kw)
    d = dict(flags=flags, niter=niter, ncomp=len(coefs), ret=res["ret"], up=int(res["up"]),
             hsamp=np.array(kw.get("hsamp") or [1] * len(coefs)),
             vsamp=np.array(kw.get("vsamp") or [1] * len(coefs)),
             colorspace=kw.get("colorspace") or (3 if len(coefs) == 3 else 1),
             image_size=np.array(kw.get("image_size") or (coefs[0].shape[1] * 8, coefs[0].shape[0] * 8)),
             hsamp0=res["hsamp0"], vsamp0=res["vsamp0"])
    for ci in range(len(coefs)):
        d[f"in{ci}"] = coefs[ci]; d[f"q{ci}"] = quants[ci]
        d[f"out{ci}"] = res["coefs"][ci]; d[f"qout{ci}"] = res["quants"][ci]
    np.savez_compressed(OUT / f"{name}.npz", **d)
    print(name, "ret", res["ret"], "up", res["up"])


def main():
    assert build_ref(), "/root/reference must be mounted to regenerate golden vectors"
    ref = Reference("none")
    synth = jpegqs_pkg.load().synth
    # BASELINE.json configs[0]: 64x64 grayscale, q=3 niter=3 -- per-iteration dumps
    coef, quant = synth.synth_gray(64, 64, 50)
    for n in (1, 2, 3):
        save(f"gray64_q3_n{n}", ref, [coef], [quant], 0, n)
    save("gray64_q4_n3", ref, [coef], [quant], 1, 3)
    save("gray64_q0_n3", ref, [coef], [quant], 9, 3)
    save("gray64_q3_norebalance_n2", ref, [coef], [quant], 16, 2)
    coef, quant = synth.synth_gray(200, 120, 25)
    save("gray200x120_q4_n3", ref, [coef], [quant], 1, 3)
    # quant table with a zero entry and an all-ones table (iterations skipped)
    coef, quant = synth.synth_gray(64, 64, 50)
    qz = quant.copy(); qz[5] = 0
    save("gray64_zeroquant_q3_n2", ref, [coef], [qz], 0, 2)
    save("gray64_onesquant_q3_n2", ref, [coef], [np.ones(64, np.uint16)], 0, 2)
    qb = quant.copy(); qb[63] = 0x800
    save("gray64_bigquant_q3_n2", ref, [coef], [qb], 0, 2)
    cb = coef.copy(); cb[3, 4, 0] = 300  # 300 * 16 = 4800 > 0x7ff -> bad_coef
    save("gray64_badcoef_q3_n2", ref, [cb], [quant], 0, 2)
    # colour: odd size 4:2:0 (edge replication, partial MCUs), all quality levels
    j = synth.synth_ycc(141, 93, 2, 2, quality=35)
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(141, 93))
    for q, fl in ((3, 0), (4, 1), (5, 3), (6, 7), (2, 15)):
        save(f"ycc420_141x93_q{q}_n2", ref, j["coefs"], j["quants"], fl, 2, **kw)
    j = synth.synth_ycc(72, 40, 1, 1, quality=60)
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(72, 40))
    save("ycc444_72x40_q6_n2", ref, j["coefs"], j["quants"], 7, 2, **kw)
    j = synth.synth_ycc(96, 64, 2, 1, quality=45)
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(96, 64))
    save("ycc422_96x64_q6_n1", ref, j["coefs"], j["quants"], 7, 1, **kw)
    # blocks whose coefficients exceed +-1023 until the final clamp: the refresh-only pass A that
    # feeds JOINT_YUV / UPSAMPLE_UV runs BEFORE that clamp (reference :2668-2689 sits after the loop)
    sys.path.insert(0, str(ROOT / "tests"))
    from helpers import inject_extreme_blocks
    j = inject_extreme_blocks(synth.synth_ycc(104, 72, 2, 2, quality=60, seed=5))
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(104, 72))
    save("ycc420_104x72_extreme_q6_n2", ref, j["coefs"], j["quants"], 7, 2, **kw)
    save("ycc420_104x72_extreme_q5_n1", ref, j["coefs"], j["quants"], 3, 1, **kw)


def make_highq_golden():
    """High-quality JPEGs (IJG quality 92 / 95 / 98): many quantiser entries are 1 there (0 / 8 / 28 of luma's 63 AC
    entries) -- the coefficients the recovery kernels skip (QS_REC_Q1, csrc/qs_device.h), because the reference's
    interval for them is a single point.  Outputs of the compiled reference, like everything else here."""
    assert build_ref(), "/root/reference must be mounted to regenerate golden vectors"
    ref = Reference("none")
    synth = jpegqs_pkg.load().synth
    for jq in (92, 95, 98):
        coef, quant = synth.synth_gray(120, 88, jq, seed=jq)
        print("quality", jq, "luma entries equal to 1:", int((quant[1:] == 1).sum()))
        save(f"gray120x88_jq{jq}_q3_n3", ref, [coef], [quant], 0, 3)
        save(f"gray120x88_jq{jq}_q4_n2", ref, [coef], [quant], 1, 2)
    coef, quant = synth.synth_gray(120, 88, 98, seed=98)
    save("gray120x88_jq98_q3_norebalance_n2", ref, [coef], [quant], 16, 2)
    for jq in (95, 98):
        j = synth.synth_ycc(104, 72, 2, 2, quality=jq, seed=jq)
        kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(104, 72))
        save(f"ycc420_104x72_jq{jq}_q6_n2", ref, j["coefs"], j["quants"], 7, 2, **kw)
        save(f"ycc420_104x72_jq{jq}_q3_n2", ref, j["coefs"], j["quants"], 0, 2, **kw)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "highq":
        make_highq_golden()
    else:
        main()


CLI_OPTION_CASES = [("f33", ["-f", "33", "-n", "2"]), ("f20_n1", ["--flags", "20", "--niter", "1"]),
                    ("c0", ["-q", "3", "-n", "3", "-c", "0"]), ("c1", ["-q", "3", "-n", "3", "--copy", "1"]),
                    ("q5_n0", ["-q", "5", "-n", "0"]), ("q6_n1_o", ["-q", "6", "-n", "1", "-o"])]


def make_cli_golden():
    """JPEG files in -> reference CLI (scalar build) -> JPEG files out, committed
    so the end-to-end CLI test can run where /root/reference is absent."""
    import subprocess
    from PIL import Image
    synth = jpegqs_pkg.load().synth
    cli = ROOT / "oracle" / "_ref" / "jpegqs_ref_none"
    out = OUT / "cli"
    out.mkdir(exist_ok=True)
    gray = synth.synth_pixels(64, 64, seed=3)
    Image.fromarray(gray, "L").save(out / "gray64.jpg", quality=50)
    rgb = np.stack([synth.synth_pixels(141, 93, seed=5, variant=v) for v in range(3)], axis=-1)
    Image.fromarray(rgb, "RGB").save(out / "rgb141x93_420.jpg", quality=35, subsampling=2)
    Image.fromarray(rgb, "RGB").save(out / "rgb141x93_444.jpg", quality=60, subsampling=0)
    for src in ("gray64", "rgb141x93_420", "rgb141x93_444"):
        for q in (2, 3, 4, 5, 6):
            dst = out / f"{src}.q{q}.ref.jpg"
            subprocess.run([str(cli), "-q", str(q), "-n", "3", "-i", "0", "-t", "1",
                            str(out / f"{src}.jpg"), str(dst)], check=True)
    # more container variety: 4 components (Adobe CMYK), progressive scans, 4:2:2, restart markers
    cmyk = np.stack([synth.synth_pixels(96, 64, seed=7, variant=v % 3) for v in range(4)], axis=-1)
    Image.fromarray(cmyk, "CMYK").save(out / "cmyk96x64.jpg", quality=55)
    rgb2 = np.stack([synth.synth_pixels(120, 88, seed=9, variant=v) for v in range(3)], axis=-1)
    Image.fromarray(rgb2, "RGB").save(out / "rgb120x88_prog.jpg", quality=45, subsampling=2, progressive=True)
    Image.fromarray(rgb2, "RGB").save(out / "rgb120x88_422_rst.jpg", quality=70, subsampling=1, restart_marker_blocks=4)
    for src, qs in (("cmyk96x64", (3, 4, 6)), ("rgb120x88_prog", (3, 6)), ("rgb120x88_422_rst", (2, 5, 6))):
        for q in qs:
            subprocess.run([str(cli), "-q", str(q), "-n", "3", "-i", "0", "-t", "1",
                            str(out / f"{src}.jpg"), str(out / f"{src}.q{q}.ref.jpg")], check=True)
    # option coverage: --flags override (NO_REBALANCE_UV + DIAGONALS), --copy 0/1, --optimize with progressive input
    for tag, args in CLI_OPTION_CASES:
        subprocess.run([str(cli), *args, "-i", "0", "-t", "1", str(out / "rgb141x93_420.jpg"),
                        str(out / f"rgb141x93_420.{tag}.ref.jpg")], check=True)
    print(sorted(p.name for p in out.iterdir()))


DECODE_CASES = [("gray64", 4, 2), ("rgb141x93_420", 3, 3), ("rgb128x96_420", 3, 2), ("rgb128x96_420", 5, 2),
                ("rgb141x93_444", 6, 3), ("gray64", 6, 2), ("rgb141x93_444", 5, 2)]
DECODE_ABORT_CASES = [("rgb141x93_420", 6, 3), ("rgb128x96_420", 6, 1)]


def make_decode_golden():
    """decode-mode API (jpegqs_start_decompress ... jpeg_read_scanlines): raw pixels delivered by the
    reference compiled into oracle/decode_demo.c (oracle/_ref/decode_ref_none).  For UPSAMPLE_UV on a
    subsampled image the reference's own decode mode dies inside libjpeg 9d's jinit_upsampler
    ("Fractional sampling not implemented yet"): that text and exit status are recorded too, the
    product has to behave the same way."""
    import subprocess
    out = OUT / "cli"
    demo = ROOT / "oracle" / "_ref" / "decode_ref_none"
    for src, q, n in DECODE_CASES:
        r = subprocess.run([str(demo), str(q), str(n), str(out / f"{src}.jpg")], capture_output=True, check=True)
        dst = out / f"{src}.q{q}.dec.ref.raw"
        if dst.exists():
            assert dst.read_bytes() == r.stdout, f"{dst.name} changed"
        dst.write_bytes(r.stdout)
    for src, q, n in DECODE_ABORT_CASES:
        r = subprocess.run([str(demo), str(q), str(n), str(out / f"{src}.jpg")], capture_output=True)
        assert r.returncode != 0
        (out / f"{src}.q{q}.dec.abort.txt").write_text(f"{r.returncode}\n{r.stderr.decode()}")


if __name__ == "__main__":
    make_cli_golden()
    make_decode_golden()
