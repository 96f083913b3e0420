#!/bin/bash
# usage (GPU box): tools/ab_bench.sh [bench args] -> per-kernel-class avg ms of the default library and of every ab/lib_*.so
# (kernel experiments: variants are built by hand into ab/, which is git-ignored)
for lib in default ab/lib_*.so; do
  [ "$lib" = default ] && unset MDGAT_HIP_LIB || export MDGAT_HIP_LIB=$PWD/$lib
  for rep in 1 2; do
    python bench.py --no-cpu-baseline --steps 30 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', round(d['value']), ' '.join(f\"{k['kernel']}={k['avg_ms']*1000:.1f}\" for k in d['kernels']))"
  done
done
