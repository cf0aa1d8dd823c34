#!/bin/bash
# Build step of the recovery kernels' translation unit (csrc/Makefile, tools/build_variants.sh):
#   build_stripped.sh <src.hip> <out.o> <min-removed> [hipcc flags ...]
# hipcc separates consecutive inline-asm statements with `s_nop 0` and has no switch for it; the object is therefore
# built through its device assembly with hipcc's own steps (`hipcc -###`): device -S | ISA guards | strip_asm_nops.py |
# assemble | lld | bundle | host compile against the prepared fat binary.
#   * the LLVM tools are the ones hipcc itself would run (`hipcc --print-prog-name=clang`), not a hard-coded prefix;
#   * every intermediate lives in a private mktemp directory that is removed on every exit path;
#   * a failed GUARD (v_ashr_pk_u8_i32 selected, a scalar load read before its wait, fewer removable no-ops than
#     <min-removed>) fails the build;
#   * missing LLVM tools or a bundle the toolchain refuses do NOT: the object is then built by the plain one-step
#     `hipcc -c` (correct, 2-6 % slower kernels) with a warning -- unless QS_REQUIRE_STRIP=1.
set -u
SRC=$1; OUT=$2; FLOOR=$3; shift 3
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
ARCH=${ARCH:-gfx950}
HERE=$(cd "$(dirname "$0")" && pwd)
TMP=$(mktemp -d "${TMPDIR:-/tmp}/qs_strip.XXXXXX") || exit 1
trap 'rm -rf "$TMP"' EXIT
B=$TMP/k

# plain <why> <hipcc flags ...>
plain() {
  local why=$1; shift                # (the remaining arguments are the hipcc flags; the message must not reach hipcc)
  echo "build_stripped.sh: WARNING: $why -- building $OUT with the plain one-step hipcc -c (no-ops between asm statements stay: 2-6 % slower kernels)" >&2
  [ "${QS_REQUIRE_STRIP:-0}" = 1 ] && { echo "build_stripped.sh: QS_REQUIRE_STRIP=1: giving up" >&2; exit 1; }
  exec "$HIPCC" "$@" -c "$SRC" -o "$OUT"
}

"$HIPCC" "$@" -S --cuda-device-only "$SRC" -o "$B.isa.s" 2> "$B.err" || { cat "$B.err" >&2; exit 1; }
grep -v "argument unused during compilation" "$B.err" >&2
# toolchain-hazard guard (idct_pass2_row): this instruction must not be selected
if grep -q v_ashr_pk_u8_i32 "$B.isa.s"; then echo "ERROR: v_ashr_pk_u8_i32 selected" >&2; exit 1; fi
if ! python3 "$HERE/check_inflight.py" "$B.isa.s"; then
  echo "ERROR: a hand-placed scalar load is read or moved before its wait" >&2; exit 1; fi
python3 "$HERE/strip_asm_nops.py" "$B.isa.s" "$B.dev.s" --min-removed "$FLOOR" || exit 1
if ! python3 "$HERE/check_inflight.py" "$B.dev.s" > "$B.chk" 2>&1; then
  cat "$B.chk" >&2; echo "ERROR: the stripped assembly fails the in-flight check" >&2; exit 1; fi

CLANG=${LLVMBIN:+$LLVMBIN/clang}; CLANG=${CLANG:-$("$HIPCC" --print-prog-name=clang 2>/dev/null)}
BIN=$(dirname "${CLANG:-/nonexistent/clang}")
for t in clang lld clang-offload-bundler; do
  [ -x "$BIN/$t" ] || plain "LLVM tool '$t' not found next to hipcc's clang ($BIN)" "$@"
done
"$BIN/clang" -x assembler -target amdgcn-amd-amdhsa -mcpu="$ARCH" -c "$B.dev.s" -o "$B.dev.o" || plain "the device assembly did not assemble" "$@"
"$BIN/lld" -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$B.co" "$B.dev.o" || plain "lld refused the device object" "$@"
"$BIN/clang-offload-bundler" -type=o -bundle-align=4096 \
    -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--"$ARCH" \
    -input=/dev/null -input="$B.co" -output="$B.hipfb" || plain "clang-offload-bundler refused the code object" "$@"
"$HIPCC" "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$B.hipfb" -c "$SRC" -o "$OUT" \
    || plain "the host compile against the prepared fat binary failed" "$@"
