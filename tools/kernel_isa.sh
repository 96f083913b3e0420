#!/bin/bash
# usage: tools/kernel_isa.sh <csrc name> <kernel name substring> [extra hipcc flags] -> /tmp/isa_<name>.s (disassembly of that kernel) + instruction histogram
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; K=$2; shift; shift
FL=""
[ "$N" = attention ] && FL="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $FL "$@" --cuda-device-only -c $R/mdgat_matcher_amd/csrc/$N.hip -o /tmp/ki_$N.co 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/ki_$N.co --output=/tmp/ki_$N.elf
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn /tmp/ki_$N.elf | awk -v k="$K" '/^[0-9a-f]+ <.*>:/{p=index($0,k)>0} p' > /tmp/isa_$N.s
wc -l /tmp/isa_$N.s
grep -oE "^\s+[a-z_0-9]+" /tmp/isa_$N.s | sed 's/^\s*//' | sed -E 's/_(e32|e64|dpp|sdwa)$//' | sort | uniq -c | sort -rn | head -${TOPN:-25}
