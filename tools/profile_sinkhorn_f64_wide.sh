#!/bin/bash
# usage (GPU box): tools/profile_sinkhorn_f64_wide.sh TAG -> gpurun_out/TAG_sinkhorn_f64_wide_{stats,pmc_FETCH_SIZE,pmc_WRITE_SIZE}.txt:
# rocprofv3 kernel stats and HBM traffic passes of the streaming fp64 Sinkhorn (8 pairs and one pair of 2048 keypoints, 200 iterations).
TAG=${1:-rX}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cat > /tmp/sk64_wide_run.py <<PY
import sys, time, torch
sys.path.insert(0, '$R')
from mdgat_matcher_amd import ops
B = int(sys.argv[1])
s = torch.randn(B, 2048, 2048, dtype=torch.float64, device='cuda') * 3
for _ in range(2): ops.sinkhorn_f64_extract(s, 1.0, 200)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): ops.sinkhorn_f64_extract(s, 1.0, 200)
torch.cuda.synchronize(); print('%d pairs of 2048 keypoints, 200 iterations: %.1f us per call' % (B, (time.perf_counter() - t0) / 3 * 1e6))
PY
cd /tmp && export TMPDIR=/tmp
: > $O/${TAG}_sinkhorn_f64_wide_stats.txt
for B in 8 1; do
  rm -rf /tmp/pw_$B
  timeout 280 rocprofv3 --kernel-trace --stats -d /tmp/pw_$B -o s -- python /tmp/sk64_wide_run.py $B > /tmp/pw_log_$B 2>&1 < /dev/null
  ( echo "# rocprofv3 --kernel-trace --stats -- ops.sinkhorn_f64_extract(randn($B, 2048, 2048) * 3, 1.0, 200) x 5"; grep "per call" /tmp/pw_log_$B
    [ -f /tmp/pw_$B/s_results.db ] && (cd $R; timeout 60 python tools/rocpd_summary.py /tmp/pw_$B/s_results.db < /dev/null | head -12) ) >> $O/${TAG}_sinkhorn_f64_wide_stats.txt 2>&1
done
for n in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pw_$n
  timeout 280 rocprofv3 --kernel-trace --pmc $n -d /tmp/pw_$n -o p -- python /tmp/sk64_wide_run.py 8 > /dev/null 2>&1 < /dev/null
  ( echo "# rocprofv3 --kernel-trace --pmc $n -- 8 pairs of 2048 keypoints, 200 iterations; per-launch averages"
    [ -f /tmp/pw_$n/p_results.db ] && (cd $R; timeout 60 python tools/pmc_summary.py /tmp/pw_$n/p_results.db < /dev/null | head -12) ) > $O/${TAG}_sinkhorn_f64_wide_pmc_$n.txt 2>&1
done
ls -la $O/${TAG}_sinkhorn_f64_wide_*
