"""CPU tests: the oracle restatement is pinned against (a) the golden vectors
generated from the compiled reference and (b), where oracle/_ref is available,
the reference itself on fresh seeded inputs and per-function KATs."""
import numpy as np
import pytest

from helpers import assert_same_result, golden_names, load_golden


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_golden(oracle, name):
    job, want = load_golden(name)
    got = oracle.do_quantsmooth(job["coefs"], job["quants"], job["flags"], job["niter"], **job["kw"])
    assert_same_result(got, want, name)


def test_golden_cover_every_flag():
    seen = 0
    for n in golden_names():
        seen |= load_golden(n)[0]["flags"]
    assert seen & 1 and seen & 2 and seen & 4 and seen & 8 and seen & 16


def test_tables_match_reference(oracle, reference):
    for flags in (0, 1):
        a, b = oracle.tables(flags), reference.tables(flags)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_idct_islow_kat(oracle, reference):
    rng = np.random.default_rng(7)
    cases = [np.zeros(64, np.int16), np.full(64, 2047, np.int16), np.full(64, -2048, np.int16)]
    dc = np.zeros(64, np.int16); dc[0] = 1000; cases.append(dc)          # zero-AC shortcut
    row = np.zeros(64, np.int16); row[:8] = rng.integers(-500, 500, 8); cases.append(row)
    col = np.zeros(64, np.int16); col[::8] = rng.integers(-500, 500, 8); cases.append(col)
    for _ in range(300):
        cases.append((rng.integers(-2048, 2048, 64) * (rng.random(64) < rng.random())).astype(np.int16))
    for _ in range(100):   # out-of-range blocks: int32 wrap-around must agree too
        cases.append(rng.integers(-32768, 32768, 64).astype(np.int16))
    for c in cases:
        assert np.array_equal(oracle.idct_islow(c), reference.idct_islow(c))


def test_float_dct_kat(oracle, reference):
    rng = np.random.default_rng(8)
    for _ in range(200):
        x = rng.normal(0, 60, 64).astype(np.float32)
        assert np.array_equal(oracle.fdct_float(x).view(np.uint32), reference.fdct_float(x).view(np.uint32))
        assert np.array_equal(oracle.idct_float(x).view(np.uint32), reference.idct_float(x).view(np.uint32))


def test_interval_forms_agree(oracle):
    """exact-division interval == the reciprocal-table form the reference runs
    (reference quantsmooth.h:332-341), sampled densely incl. every power of two"""
    import ctypes as C
    divs = sorted(set(list(range(1, 300)) + [2 ** n for n in range(11)] + [2 ** n - 1 for n in range(2, 12)]
                      + [2 ** n + 1 for n in range(1, 11)] + [0x7ff, 1000, 1531]))
    o = C.c_int(0)
    for div in divs:
        if div > 0x7ff:
            continue
        for coef in list(range(-3200, 3201, 7)) + [-0x4000, 0x3fff, -1, 0, 1, div // 2, -(div // 2), div, -div]:
            orig, lo, hi = oracle.interval(coef, div)
            oracle._recip(coef, div, C.byref(o))
            assert o.value == orig, (coef, div, o.value, orig)
            assert lo <= coef <= hi


def test_block_nan_path(oracle, reference, synth):
    """all neighbour differences >= range => a3 == 0 => NaN => INT_MIN => wrap =>
    clamp: checkerboard block, q = 2"""
    q = np.full(64, 2, np.uint16)
    plane = np.zeros((8 + 2, 8 + 2), np.uint8)
    yy, xx = np.mgrid[0:10, 0:10]
    plane[:] = np.where((yy + xx) & 1, 255, 0)
    coef = np.zeros(64, np.int16); coef[63] = 600; coef[0] = 100
    for flags in (0, 1):
        a = oracle.block(coef, q, plane, 0, 0, flags)
        b = reference.block(coef, q, plane, 0, 0, flags)
        assert np.array_equal(a, b)


def test_block_random_kat(oracle, reference):
    rng = np.random.default_rng(11)
    for trial in range(150):
        scale = [1, 4, 16, 60][trial % 4]
        q = np.clip(rng.integers(1, 4 * scale + 1, 64), 1, 255).astype(np.uint16)
        plane = np.clip(rng.normal(128, [3, 20, 80][trial % 3], (26, 26)), 0, 255).astype(np.uint8)
        plane2 = np.clip(plane.astype(int) + rng.integers(-20, 20, plane.shape), 0, 255).astype(np.uint8)
        coef = (rng.integers(-40, 40, 64) * q).astype(np.int16)
        coef = (coef // np.maximum(q, 1) * q).astype(np.int16)
        for flags in (0, 1, 8, 16, 1 | 32):
            for luma in (0, 1):
                a = oracle.block(coef, q, plane, 1, 1, flags, luma)
                b = reference.block(coef, q, plane, 1, 1, flags, luma)
                assert np.array_equal(a, b), (trial, flags, luma)
        for flags in (2, 3, 2 | 8):
            a = oracle.block(coef, q, plane, 1, 1, flags, 0, plane2)
            b = reference.block(coef, q, plane, 1, 1, flags, 0, plane2)
            assert np.array_equal(a, b), (trial, flags)


@pytest.mark.parametrize("size,samp", [((64, 64), (2, 2)), ((333, 517), (2, 2)), ((321, 100), (2, 2)),
                                       ((129, 65), (1, 1)), ((100, 60), (2, 1)), ((90, 70), (1, 2)), ((8, 8), (2, 2))])
def test_oracle_vs_reference_colour(oracle, reference, synth, size, samp):
    w, h = size
    j = synth.synth_ycc(w, h, samp[0], samp[1], quality=40, seed=99)
    kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(w, h))
    for flags in (0, 1, 3, 7, 5, 10, 15, 7 | 16, 7 | 32):
        for niter in (0, 2):
            a = oracle.do_quantsmooth(j["coefs"], j["quants"], flags, niter, **kw)
            b = reference.do_quantsmooth(j["coefs"], j["quants"], flags, niter, **kw)
            assert_same_result(a, b, f"{size} {samp} flags={flags} niter={niter}")


def test_oracle_vs_reference_gray(oracle, reference, synth):
    for (w, h, qual) in ((64, 64, 50), (200, 120, 20), (24, 88, 92)):
        coef, quant = synth.synth_gray(w, h, qual, seed=5)
        for flags in (0, 1, 8, 9, 16, 17):
            a = oracle.do_quantsmooth([coef], [quant], flags, 3)
            b = reference.do_quantsmooth([coef], [quant], flags, 3)
            assert_same_result(a, b, f"{w}x{h} q{qual} flags={flags}")


def test_progress_and_cancel(oracle, reference, synth):
    """progress is reported between iterations and a non-zero return cancels
    (reference quantsmooth.h:2474-2482, 2656-2664)"""
    coef, quant = synth.synth_gray(64, 64, 50)
    for cancel_at in (None, 0, 1):
        logs = []
        for impl in (oracle, reference):
            calls = []

            def cb(_u, cur, mx, calls=calls):
                calls.append((cur, mx))
                return 1 if cancel_at is not None and len(calls) - 1 == cancel_at else 0
            res = impl.do_quantsmooth([coef], [quant], 0, 4, progprec=0, progress=cb)
            logs.append((calls, res))
        assert logs[0][0] == logs[1][0]
        assert_same_result(logs[0][1], logs[1][1], f"cancel_at={cancel_at}")
        if cancel_at is None:
            assert logs[0][0][-1] == (20, 20)


def test_multithreaded_oracle_is_deterministic(oracle, synth):
    coef, quant = synth.synth_gray(256, 128, 50)
    a = oracle.do_quantsmooth([coef], [quant], 1, 2, threads=1)
    b = oracle.do_quantsmooth([coef], [quant], 1, 2, threads=4)
    assert_same_result(a, b)


def test_fuzz_corpus_is_what_the_compiled_reference_produces(reference, tmp_path):
    """tests/golden/fuzz_s2.jsonl (400 trials / 959 jobs, every flags value, all layouts; replayed by the GPU suite in
    every kernel form and route) holds hashes of expected outputs.  Regenerated here from the COMPILED, UNMODIFIED
    reference (tools/fuzz_gpu.py gen with FUZZ_TRUTH=ref): the file must come out byte for byte -- the GPU suite's fuzz
    replays are then comparisons with oracle/_ref itself, not with the port."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    out = tmp_path / "fuzz_ref.jsonl"
    r = subprocess.run([sys.executable, str(root / "tools" / "fuzz_gpu.py"), "gen", str(out), "400", "2"],
                       capture_output=True, text=True, timeout=1500, cwd=str(root), env=dict(os.environ, FUZZ_TRUTH="ref"))
    assert r.returncode == 0 and "truth = Reference" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
    assert out.read_bytes() == (root / "tests" / "golden" / "fuzz_s2.jsonl").read_bytes()
