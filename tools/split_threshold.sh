#!/bin/bash
# usage (GPU box): tools/split_threshold.sh "1 2 4 8 12 16 24 32" -> ms per step of bench.py --config 1 --batch B with the layer tail on
# csrc/layer.hip (MDGAT_LAYER_SPLIT_TILES=0) and on csrc/layer_split.hip (=100000), plus the two kernels' own launch averages
for B in $1; do
  for T in 0 100000; do
    MDGAT_LAYER_SPLIT_TILES=$T python bench.py --config ${CONFIG:-1} --batch $B --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k={x['kernel']:x['avg_ms'] for x in d.get('kernels',[])}
print('B=$B split_tiles=$T ms_per_step %.4f layer %.4f first %.4f last %.4f' % (d['ms_per_step'], k.get('layer',0), k.get('layer_first',0), k.get('layer_last',0)))
"
  done
done
