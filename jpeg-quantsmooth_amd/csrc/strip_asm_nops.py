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
  * the line before it closes an asm statement that is NOT empty, and whose last instruction is on
    the list below -- full-width 32-bit VALU writes or scalar instructions, none of which can be the
    producer of the hazard; anything the compiler emitted itself lies at least a whole statement
    further back, outside the one-wait-state window;
  * the line after it opens the next asm statement.
Everything else -- no-ops next to compiler-generated code, after empty (register-pin) statements,
longer waits -- stays.  The result is what the assembler would have produced had the two statements
been written as one.

usage: strip_asm_nops.py <in.s> <out.s>     prints the counts (a compiler that stops inserting the no-ops is not an error)
"""
import sys

# last instruction of an asm statement behind which the no-op may go
FULL_WIDTH = {"v_add_f32", "v_sub_f32", "v_mul_f32", "v_add_f32_e32", "v_sub_f32_e32", "v_mul_f32_e32",
              "v_add_f32_e64", "v_sub_f32_e64", "v_mul_f32_e64",
              "s_waitcnt", "s_load_dword", "s_load_dwordx2", "s_load_dwordx4", "s_load_dwordx8", "s_load_dwordx16"}


def main(src_path, dst_path):
    src = open(src_path).read().split("\n")
    out, removed, kept = [], 0, 0
    last_in_asm, in_asm = None, False   # mnemonic of the last instruction of the asm statement just closed
    i = 0
    while i < len(src):
        line = src[i]
        text = line.split(";")[0].strip() if ";;#" not in line else ""
        if "#ASMSTART" in line:
            in_asm, last_in_asm = True, None
        elif "#ASMEND" in line:
            in_asm = False
            nxt = src[i + 1].strip() if i + 1 < len(src) else ""
            nxt2 = src[i + 2] if i + 2 < len(src) else ""
            if nxt == "s_nop 0":
                if last_in_asm in FULL_WIDTH and "#ASMSTART" in nxt2:
                    out.append(line)
                    removed += 1
                    i += 2
                    continue
                kept += 1
        elif in_asm and text and not text.endswith(":") and not text.startswith("."):
            last_in_asm = text.split()[0]
        out.append(line)
        i += 1
    open(dst_path, "w").write("\n".join(out))
    print(f"strip_asm_nops: {removed} no-ops between two asm statements removed, {kept} after an asm statement kept")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
