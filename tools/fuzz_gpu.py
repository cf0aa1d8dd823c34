#!/usr/bin/env python3
"""Randomised GPU-vs-oracle campaign: larger images than the test suite affords, every
flag combination, extreme blocks, sparse planes, single calls and batches.

The scalar oracle is the slow side, so the campaign is split: `gen` runs ONLY the oracle
(anywhere, no GPU) and writes one JSON line per job with the hashes of the expected
outputs; `run` regenerates the same seeded inputs on the GPU box, runs the product and
compares hashes -- hundreds of jobs in a minute of GPU time.

    python tools/fuzz_gpu.py gen <file.jsonl> <trials> [seed=1]     # CPU only
    python tools/fuzz_gpu.py run <file.jsonl>                       # GPU box
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import jpegqs_pkg  # noqa: E402
from helpers import inject_extreme_blocks  # noqa: E402

import hashlib  # noqa: E402
import json  # noqa: E402

mode, path = sys.argv[1], Path(sys.argv[2])
pkg = jpegqs_pkg.load()
synth = pkg.synth
LAYOUTS = [(1, 1), (2, 2), (2, 1), (1, 2), (4, 1), (2, 2), (2, 2)]


def make_job(rng, trial, big):
    hi = 1400 if big else 260
    w, h = int(rng.integers(8, hi)), int(rng.integers(8, hi * 3 // 4))
    qual = int(rng.choice([1, 5, 20, 35, 50, 65, 80, 95, 100]))
    if rng.random() < 0.3:
        coef, quant = synth.synth_gray(w, h, qual, seed=trial)
        if rng.random() < 0.3:
            coef = (coef * (rng.random(coef.shape[:2]) < 0.3)[:, :, None]).astype(np.int16)
        j = dict(coefs=[coef], quants=[quant])
        desc = f"gray {w}x{h} q{qual}"
    else:
        hs, vs = LAYOUTS[int(rng.integers(0, len(LAYOUTS)))]
        y = synth.synth_ycc(w, h, hs, vs, quality=qual, seed=trial)
        if rng.random() < 0.5 and qual >= 20:
            y = inject_extreme_blocks(y, seed=trial)
        j = dict(coefs=y["coefs"], quants=y["quants"], hsamp=y["hsamp"], vsamp=y["vsamp"], colorspace=3, image_size=(w, h))
        desc = f"ycc {w}x{h} {hs}x{vs} q{qual}"
    return j, desc


def digest(res):
    h = lambda x: hashlib.sha1(np.ascontiguousarray(x).tobytes()).hexdigest()[:16]
    return dict(ret=int(res["ret"]), up=bool(res["up"]), samp=[int(res["hsamp0"]), int(res["vsamp0"])],
                coefs=[[list(c.shape), h(c)] for c in res["coefs"]],
                quants=[None if q is None else h(q) for q in res["quants"]])


def trial_jobs(seed0, trial):
    """the jobs of one trial: ([(job, desc)], flags, niter, is_batch) -- a pure function of the seeds"""
    rng = np.random.default_rng([seed0, trial])
    flags = int(rng.integers(0, 128))
    niter = int(rng.choice([0, 1, 2, 3, 3, 3, 5]))
    if trial % 4 == 0:                                   # a batch of small / medium jobs
        n = int(rng.integers(2, 12))
        return [make_job(rng, trial * 100 + k, big=False) for k in range(n)], flags, niter, True
    return [make_job(rng, trial * 100, big=trial % 4 == 1)], flags, niter, False


def kwargs(j):
    return {k: j[k] for k in ("hsamp", "vsamp", "colorspace", "image_size") if k in j}


if mode == "gen":
    from oracle.oracle import Oracle, Reference, have_ref   # the replay side (`run`) never touches the oracle
    # FUZZ_TRUTH=ref: the expected hashes come from the COMPILED, UNMODIFIED reference (oracle/_ref/libqsref_none.so) instead
    # of the plain-C port -- slower, and how the committed corpus was re-derived in round 6 (identical file: the port is
    # pinned to the reference on CPU, this shows it on the corpus itself)
    import os
    oracle = Reference("none") if os.environ.get("FUZZ_TRUTH") == "ref" and have_ref("none") else Oracle()
    print("gen: truth =", type(oracle).__name__)
    ntrials = int(sys.argv[3]); seed0 = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    t0 = time.time()
    with open(path, "w") as f:
        for trial in range(1, ntrials + 1):
            made, flags, niter, is_batch = trial_jobs(seed0, trial)
            exp = [digest(oracle.do_quantsmooth(j["coefs"], j["quants"], flags, niter, threads=0, **kwargs(j))) for j, _ in made]
            f.write(json.dumps(dict(seed0=seed0, trial=trial, flags=flags, niter=niter, batch=is_batch,
                                    desc=[d for _, d in made], expect=exp)) + "\n")
            f.flush()
    print(f"gen: {ntrials} trials in {time.time() - t0:.0f} s -> {path}")
    sys.exit(0)

hip = pkg.HipQS()
fails = jobs_done = blocks = trials = 0
t0 = time.time()
for line in open(path):
    rec = json.loads(line)
    made, flags, niter, is_batch = trial_jobs(rec["seed0"], rec["trial"])
    assert (flags, niter, is_batch, [d for _, d in made]) == (rec["flags"], rec["niter"], rec["batch"], rec["desc"]), "generator drift"
    if is_batch:
        got = hip.do_quantsmooth_batch([m[0] for m in made], flags, niter)
    else:
        j = made[0][0]
        got = [hip.do_quantsmooth(j["coefs"], j["quants"], flags, niter, **kwargs(j))]
    trials += 1
    for (j, desc), g, want in zip(made, got, rec["expect"]):
        jobs_done += 1
        blocks += sum(c.shape[0] * c.shape[1] for c in j["coefs"])
        if digest(g) != want:
            fails += 1
            print(f"FAIL seed=({rec['seed0']},{rec['trial']}) {'batch ' if is_batch else ''}{desc} flags={flags} niter={niter}", flush=True)
print(f"fuzz: {trials} trials, {jobs_done} jobs, {blocks} blocks, {fails} failures, {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if fails else 0)
