#!/usr/bin/env python3
"""Build step for qs_kernels.hip (csrc/Makefile): remove the `s_nop 0` hipcc puts between two
consecutive inline-asm statements of the recovery kernels.

Why it is there: on gfx940+ a VALU instruction that writes only part of its destination
(dst_sel / op_sel) must be one wait state away from a reader of that register.  The compiler cannot
see inside an asm statement, so whenever an asm statement reads or writes a register the PREVIOUS asm
statement defined it assumes the worst and separates the two with `s_nop 0`
(GCNHazardRecognizer::checkInlineAsmHazards).  Every term of the recovery loop is such a statement
(nine plain 32-bit VOP2/VOP3 instructions accumulating into num/den), so a wave executes ~8,600 of
these no-ops per block-iteration: 2 % of a full-size launch, 4-6 % of a band-sized one
(profiles/r04m_nonop).  hipcc has no switch for it, hence this pass over the device assembly.

Why removing it is safe, and what this script checks before it does: a no-op is dropped only when
  * the line before it closes an asm statement that is NOT empty and in which EVERY instruction is on
    the list below -- full-width 32-bit VALU writes and scalar instructions, none of which can be the
    producer of the hazard.  That covers statements with a branch inside as well (the small-plane
    kernel's optional terms: `s_cmp / s_cbranch_scc1 1f / nine VALU / 1:`): whichever path a wave
    takes, the last instruction it executed before the no-op is on the list.  Anything the compiler
    emitted itself lies at least a whole statement further back, outside the one-wait-state window;
  * the line after it opens the next asm statement.
Everything else -- no-ops next to compiler-generated code, after empty (register-pin) statements,
after statements with any other instruction in them, longer waits -- stays.  The result is what the
assembler would have produced had the two statements been written as one.

usage: strip_asm_nops.py <in.s> <out.s> [--min-removed N]
prints the counts.  Without --min-removed a source with nothing to strip is not an error (the A/B variants of
tools/build_variants.sh); csrc/Makefile passes the floor for the shipped translation unit, so that a toolchain
whose assembly printer no longer marks the statements (`;;#ASMSTART` / `;;#ASMEND`) fails the build instead of
silently shipping a 2-6 % slower kernel.
"""
import sys

# the instructions an asm statement may consist of for the no-op behind it to go: full-width f32 VALU writes, scalar
# memory / wait instructions, and the scalar compare-and-branch of the optional terms
FULL_WIDTH = {"v_add_f32", "v_sub_f32", "v_mul_f32", "v_add_f32_e32", "v_sub_f32_e32", "v_mul_f32_e32",
              "v_add_f32_e64", "v_sub_f32_e64", "v_mul_f32_e64",
              "s_waitcnt", "s_load_dword", "s_load_dwordx2", "s_load_dwordx4", "s_load_dwordx8", "s_load_dwordx16",
              "s_cmp_lg_u32", "s_cmp_eq_u32", "s_cbranch_scc0", "s_cbranch_scc1"}


def main(src_path, dst_path, min_removed=0):
    src = open(src_path).read().split("\n")
    out, removed, kept = [], 0, 0
    ops, in_asm = [], False   # mnemonics of the asm statement being read / just closed
    i = 0
    while i < len(src):
        line = src[i]
        text = line.split(";")[0].strip() if ";;#" not in line else ""
        if "#ASMSTART" in line:
            in_asm, ops = True, []
        elif "#ASMEND" in line:
            in_asm = False
            nxt = src[i + 1].strip() if i + 1 < len(src) else ""
            nxt2 = src[i + 2] if i + 2 < len(src) else ""
            if nxt == "s_nop 0":
                if ops and all(o in FULL_WIDTH for o in ops) and "#ASMSTART" in nxt2:
                    out.append(line)
                    removed += 1
                    i += 2
                    continue
                kept += 1
        elif in_asm and text and not text.endswith(":") and not text.startswith("."):
            ops.append(text.split()[0])
        out.append(line)
        i += 1
    open(dst_path, "w").write("\n".join(out))
    print(f"strip_asm_nops: {removed} no-ops between two asm statements removed, {kept} after an asm statement kept")
    if removed < min_removed:
        print(f"strip_asm_nops: ERROR: expected at least {min_removed} removable no-ops in {src_path} -- the assembly printer of "
              f"this toolchain no longer marks inline-asm statements the way this step reads them (;;#ASMSTART / ;;#ASMEND), "
              f"or hipcc stopped separating them with `s_nop 0` (then lower the floor in csrc/Makefile)", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    floor = 0
    argv = sys.argv[1:]
    if "--min-removed" in argv:
        k = argv.index("--min-removed")
        floor = int(argv[k + 1])
        del argv[k:k + 2]
    sys.exit(main(argv[0], argv[1], floor))
