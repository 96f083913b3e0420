#!/bin/bash
# usage (GPU box): tools/profile_f64.sh TAG [BATCH] -> gpurun_out/TAG_f64_*: the bench line of the reference-exact mode as the timed
# step (python bench.py --arithmetic fp64), rocprofv3 kernel stats of the same command and the SQ / traffic PMC passes.
TAG=${1:-rX}
BATCH=${2:-32}
CONFIG=${CONFIG:-1}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
python bench.py --config $CONFIG --arithmetic fp64 --batch $BATCH --steps 5 --warmup 2 --no-cpu-baseline --no-dict-api --no-latency 2>/dev/null | tail -1 > $O/${TAG}_f64_bench_line.json
CMD="python $R/bench.py --config $CONFIG --arithmetic fp64 --batch $BATCH --steps 3 --warmup 1 --windows 1 --no-cpu-baseline --no-dict-api --no-latency --no-exact-mode --no-parity"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_stats && rocprofv3 --kernel-trace --stats -d /tmp/pf_stats -o s -- $CMD > /dev/null 2>&1
( echo "# rocprofv3 --kernel-trace --stats -- ${CMD#python $R/}"; cd $R; python tools/rocpd_summary.py /tmp/pf_stats/s_results.db ) > $O/${TAG}_f64_kernel_stats.txt 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA"; do
  n=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pf_$n && rocprofv3 --kernel-trace --pmc $pass -d /tmp/pf_$n -o p -- $CMD > /dev/null 2>&1
  ( echo "# rocprofv3 --kernel-trace --pmc $pass -- ${CMD#python $R/}; per-launch averages"; cd $R; python tools/pmc_summary.py /tmp/pf_$n/p_results.db ) > $O/${TAG}_f64_pmc_$n.txt 2>&1
done
ls -la $O/${TAG}_f64_*
