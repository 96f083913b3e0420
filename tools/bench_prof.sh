#!/bin/bash
# usage (on the GPU box): tools/bench_prof.sh TAG  -> bench line + rocprofv3 kernel stats under gpurun_out/prof_TAG
TAG=${1:-x}
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pairs/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3))"
R=$PWD
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown > /dev/null 2>&1
cd $R; python tools/rocpd_summary.py gpurun_out/prof_$TAG/${TAG}_results.db 2>&1 | head -13 | cut -c1-60,110-200
