#!/bin/bash
# usage: tools/ab_build.sh <csrc name> <variant name> [-D flags...] -> ab/lib_<variant>.so = the current objects with <name>.hip rebuilt with the flags
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/mdgat_matcher_amd/csrc
N=$1; V=$2; shift; shift
FL=""; [ "$N" = attention ] && FL="-fno-slp-vectorize"; [ "$N" = f64 ] && FL="-mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans"
mkdir -p $R/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -fno-gpu-rdc $FL "$@" -c $C/$N.hip -o $C/build/ab_$V.o 2>&1 | grep -E "error" || true
OBJS=$(ls $C/build/*.o | grep -v "/ab_" | grep -v "/$N.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/lib_$V.so $C/build/ab_$V.o $OBJS
ls -la $R/ab/lib_$V.so
