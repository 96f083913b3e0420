#!/bin/bash
# usage (GPU box): tools/bench_kernels.sh [reps] [bench args] -> pairs/s and the per-kernel-class average us of the default library
R=${1:-2}; shift
for rep in $(seq $R); do
  python bench.py --no-cpu-baseline --steps 30 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), ' '.join('%s=%.1f' % (k['kernel'], k['avg_ms']*1000) for k in d['kernels']))"
done
