#!/usr/bin/env python3
"""Build-time guard for the recovery kernels' hand-placed scalar loads (csrc/Makefile runs it
on the device ISA of qs_kernels.hip).

The kernels stream weights and per-coefficient records with s_load instructions issued from
inline asm and wait for them with an explicit `s_waitcnt lgkmcnt(0)` several hundred
instructions later.  The compiler does not know that the destination SGPRs are still being
written: it may copy, spill or reuse them right behind the asm statement (it did, round 2: a
record fetched a coefficient ahead was moved to other registers at a control-flow join before
the data had landed -- wrong results only while the scalar cache was cold).  This script
makes that a build error instead of a timing-dependent bug: between an inline-asm s_load and
the next `s_waitcnt lgkmcnt(0)` NO instruction may name one of the load's destination
registers, and no basic-block boundary may be crossed (the self-contained `1f` skips of the
optional terms excepted).

usage: check_inflight.py <file.s> [kernel-name-substring ...]      exit status 1 on a violation
"""
import re
import sys

SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check(path, wanted):
    errors, checked = [], 0
    func, in_asm, inflight, issued_at = None, False, set(), None
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].rstrip() if ";;#" not in raw else raw.strip()
        if "#ASMSTART" in raw:
            in_asm = True
            continue
        if "#ASMEND" in raw:
            in_asm = False
            continue
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            func = m.group(1) if any(w in m.group(1) for w in wanted) else None
            inflight = set()
            continue
        if func is None:
            continue
        text = line.strip()
        if not text:
            continue
        if re.match(r"^\.LBB\w+:", text):                       # basic-block boundary
            if inflight:
                errors.append(f"{path}:{ln}: {func}: label {text} while s{sorted(inflight)[0]}.. (loaded at line {issued_at}) is in flight")
            continue
        if re.match(r"^\d+:$", text) or text.startswith("."):   # local label of an optional term / directive
            continue
        op = text.split()[0]
        if op == "s_waitcnt":
            if "lgkmcnt(0)" in text:
                inflight = set()
            continue
        if op.startswith("s_load_") and in_asm:
            ops = text[len(op):].split(",")
            dst, srcs = sregs(ops[0]), sregs(",".join(ops[1:]))
            if srcs & inflight:
                errors.append(f"{path}:{ln}: {func}: `{text}` reads registers that are in flight (loaded at line {issued_at})")
            inflight |= dst
            issued_at = ln
            checked += 1
            continue
        if not inflight:
            continue
        if op in ("s_branch", "s_setpc_b64", "s_endpgm") or (op.startswith("s_cbranch") and not re.search(r"\b\d+f\b", text)):
            errors.append(f"{path}:{ln}: {func}: `{text}` leaves the block while registers loaded at line {issued_at} are in flight")
            continue
        if sregs(text) & inflight:
            errors.append(f"{path}:{ln}: {func}: `{text}` touches s{sorted(sregs(text) & inflight)[0]} before the load of line {issued_at} has been waited for")
    return errors, checked


if __name__ == "__main__":
    wanted = sys.argv[2:] or ["qs_smooth"]
    errs, n = check(sys.argv[1], wanted)
    for e in errs[:20]:
        print("check_inflight: " + e, file=sys.stderr)
    if errs:
        print(f"check_inflight: {len(errs)} violation(s)", file=sys.stderr)
        sys.exit(1)
    print(f"check_inflight: {n} hand-placed scalar loads checked, none read or moved before its wait")
