#!/bin/bash
# usage (GPU box): tools/cs_knock.sh "base no_MFMA ..." -> per-launch averages of the layer classes under MDGAT_LAYER_CS=1, single lane
for v in $1; do
  if [ $v = base ]; then L=""; else L=$PWD/ab/lib_cs_$v.so; fi
  MDGAT_HIP_LIB=$L MDGAT_LAYER_CS=${CS:-1} MDGAT_FORWARD_LANES=1 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --windows 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k={x['kernel']:x['avg_ms'] for x in d.get('kernels',[])}
print('$v ms_per_step %.4f layer %.4f first %.4f last %.4f' % (d['ms_per_step'], k.get('layer',0), k.get('layer_first',0), k.get('layer_last',0)))
"
done
