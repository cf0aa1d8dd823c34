"""shared helpers for the parity tests"""
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def golden_names():
    return sorted(p.stem for p in GOLDEN.glob("*.npz"))


def load_golden(name):
    d = np.load(GOLDEN / f"{name}.npz")
    n = int(d["ncomp"])
    job = dict(
        coefs=[d[f"in{ci}"] for ci in range(n)],
        quants=[d[f"q{ci}"] for ci in range(n)],
        flags=int(d["flags"]), niter=int(d["niter"]),
        kw=dict(hsamp=[int(v) for v in d["hsamp"]], vsamp=[int(v) for v in d["vsamp"]],
                colorspace=int(d["colorspace"]), image_size=tuple(int(v) for v in d["image_size"])),
    )
    want = dict(ret=int(d["ret"]), up=bool(int(d["up"])), hsamp0=int(d["hsamp0"]), vsamp0=int(d["vsamp0"]),
                coefs=[d[f"out{ci}"] for ci in range(n)], quants=[d[f"qout{ci}"] for ci in range(n)])
    return job, want


def assert_same_result(got, want, what=""):
    assert got["ret"] == want["ret"], f"{what}: return value {got['ret']} != {want['ret']}"
    assert bool(got["up"]) == bool(want["up"]), f"{what}: upsample flag"
    assert (got["hsamp0"], got["vsamp0"]) == (want["hsamp0"], want["vsamp0"]), f"{what}: sampling factors"
    for ci, (a, b) in enumerate(zip(got["coefs"], want["coefs"])):
        assert a.shape == b.shape, f"{what}: component {ci} shape {a.shape} != {b.shape}"
        nbad = int((a != b).sum())
        if nbad:
            blocks = np.argwhere((a != b).any(axis=2))
            raise AssertionError(f"{what}: component {ci}: {nbad} coefficients differ in "
                                 f"{len(blocks)} blocks, first at (by,bx)={tuple(blocks[0])}")
    for ci, (a, b) in enumerate(zip(got["quants"], want["quants"])):
        if a is None or b is None:
            assert a is None and b is None
        else:
            assert np.array_equal(a, b), f"{what}: component {ci} quant table"


# flag bits the GPU job layer does not implement (none left)
GPU_FLAG_MASK_UNSUPPORTED = 0


def inject_extreme_blocks(j, seed=7):
    """blocks whose dequantised DC lies beyond +-1023 with strong low AC terms: their IDCT
    depends on whether the +-1023 clamp has happened yet"""
    rng = np.random.default_rng(seed)
    coefs = [c.copy() for c in j["coefs"]]
    for ci, (c, q) in enumerate(zip(coefs, j["quants"])):
        hb, wb = c.shape[:2]
        for _ in range(max(2, hb * wb // 6)):
            by, bx = rng.integers(0, hb), rng.integers(0, wb)
            sgn = 1 if rng.integers(0, 2) else -1
            c[by, bx, 0] = sgn * (int(rng.integers(1200, 2000)) // int(q[0]))
            for k in (1, 8, 9, 2, 16):
                c[by, bx, k] = -sgn * (int(rng.integers(300, 1500)) // int(q[k])) * (1 if rng.integers(0, 2) else -1)
    return dict(j, coefs=coefs)
