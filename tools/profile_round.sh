#!/bin/bash
# usage (GPU box): [CONFIG=1] [BENCH_ARGS='--attention-dtype f16'] tools/profile_round.sh TAG -> gpurun_out/TAG_*: bench line,
# rocprofv3 kernel stats of the same command, and the PMC passes (FETCH_SIZE / WRITE_SIZE / SQ issue counters; each its own
# --pmc run, kernel-trace only).  Copy what is to be judged into profiles/ and fold the traffic into
# profiles/pmc_traffic.json with tools/pmc_to_json.py.
TAG=${1:-rX}
CONFIG=${CONFIG:-1}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
python bench.py --config $CONFIG --steps 20 --warmup 5 $BENCH_ARGS 2>/dev/null | tail -1 > $O/${TAG}_bench_line.json
CMD="python $R/bench.py --config $CONFIG --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown --no-latency --no-exact-mode --no-parity $BENCH_ARGS"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pr_stats && rocprofv3 --kernel-trace --stats -d /tmp/pr_stats -o s -- $CMD > /dev/null 2>&1
( echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config $CONFIG --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown --no-latency --no-exact-mode --no-parity $BENCH_ARGS"; cd $R; python tools/rocpd_summary.py /tmp/pr_stats/s_results.db ) > $O/${TAG}_bench_kernel_stats.txt 2>&1
# the same command with every kernel of the batch on one stream (no two launches overlap: what bench.py's `roofline` block measures)
rm -rf /tmp/pr_stats1 && MDGAT_FORWARD_LANES=1 rocprofv3 --kernel-trace --stats -d /tmp/pr_stats1 -o s -- $CMD > /dev/null 2>&1
( echo "# MDGAT_FORWARD_LANES=1 rocprofv3 --kernel-trace --stats -- python bench.py --config $CONFIG --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown --no-latency --no-exact-mode --no-parity $BENCH_ARGS"; cd $R; python tools/rocpd_summary.py /tmp/pr_stats1/s_results.db ) > $O/${TAG}_single_lane_bench_kernel_stats.txt 2>&1
( echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python bench.py --config $CONFIG --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown --no-exact-mode --no-parity $BENCH_ARGS; KB per launch, averages" ) > $O/${TAG}_pmc_fetch_write_kb.txt
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pr_$n && rocprofv3 --kernel-trace --pmc $pass -d /tmp/pr_$n -o p -- $CMD > /dev/null 2>&1
  ( cd $R; python tools/pmc_summary.py /tmp/pr_$n/p_results.db ) > $O/${TAG}_pmc_$n.txt 2>&1
done
cat $O/${TAG}_pmc_FETCH_SIZE.txt $O/${TAG}_pmc_WRITE_SIZE.txt >> $O/${TAG}_pmc_fetch_write_kb.txt
# the same two traffic passes with every launch on one stream (what bench.py's `roofline` block - measured under
# mdgat_set_lanes(1) - cites: profiles/pmc_traffic.json key "<config>:single_lane")
( echo "# MDGAT_FORWARD_LANES=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python bench.py --config $CONFIG --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown --no-exact-mode --no-parity $BENCH_ARGS; KB per launch, averages" ) > $O/${TAG}_single_lane_pmc_fetch_write_kb.txt
for n in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pr1_$n && MDGAT_FORWARD_LANES=1 rocprofv3 --kernel-trace --pmc $n -d /tmp/pr1_$n -o p -- $CMD > /dev/null 2>&1
  ( cd $R; python tools/pmc_summary.py /tmp/pr1_$n/p_results.db ) > $O/${TAG}_single_lane_pmc_$n.txt 2>&1
done
cat $O/${TAG}_single_lane_pmc_FETCH_SIZE.txt $O/${TAG}_single_lane_pmc_WRITE_SIZE.txt >> $O/${TAG}_single_lane_pmc_fetch_write_kb.txt
( echo "# rocprofv3 --kernel-trace --pmc SQ_* (one pass) -- the same command; per-launch averages"; cat $O/${TAG}_pmc_SQ_WAVE_CYCLES.txt ) > $O/${TAG}_pmc_sq_counters.txt
ls -la $O/${TAG}_*
