#!/bin/bash
# hipcc_agpr.sh SRC.hip OUT.o "KERNEL_REGEX=N[,KERNEL_REGEX=N...]" [hipcc flags...]
#
# Compiles one HIP translation unit like `hipcc -c`, except that the device functions whose (mangled) name matches KERNEL_REGEX get
# the LLVM function attribute "amdgpu-agpr-alloc"="N": exactly N accumulation registers (AGPRs) next to (budget - N) architectural
# VGPRs.  hipcc has no source-level spelling for this; by default a gfx950 kernel that touches AGPRs at all (inline asm with "a"
# constraints) has its register budget split in HALVES (128 + 128 at two waves per SIMD), which turns a 223-VGPR kernel into 130
# spills.  The steps are the ones `hipcc -###` prints for a normal compile, with one text edit of the device IR in between:
#   device IR -> (attribute) -> gfx950 object -> code object -> offload bundle -> host object that embeds it
set -euo pipefail
SRC=$1; OUT=$2; SPEC=$3; shift 3
LLVM=${ROCM_PATH:-/opt/rocm}/lib/llvm/bin
TMP=$(mktemp -d "${TMPDIR:-/tmp}/hipcc_agpr.XXXXXX")
trap 'rm -rf "$TMP"' EXIT
B=$TMP/$(basename "${SRC%.hip}")
hipcc "$@" --cuda-device-only -emit-llvm -S "$SRC" -o "$B.ll"
python3 "$(dirname "$0")/agpr_attr.py" "$B.ll" "$B.agpr.ll" "$SPEC"
"$LLVM/clang" -x ir "$B.agpr.ll" -target amdgcn-amd-amdhsa -mcpu=gfx950 -O3 -fPIC -mllvm -amdgpu-mfma-vgpr-form -c -o "$B.dev.o"
"$LLVM/lld" -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$B.hsaco" "$B.dev.o"
"$LLVM/clang-offload-bundler" -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
    -input=/dev/null -input="$B.hsaco" -output="$B.hipfb"
hipcc "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$B.hipfb" -c "$SRC" -o "$OUT"
if [ -n "${RVSR_KEEP_ISA:-}" ]; then "$LLVM/clang" -x ir "$B.agpr.ll" -target amdgcn-amd-amdhsa -mcpu=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -S -o "$RVSR_KEEP_ISA"; fi
