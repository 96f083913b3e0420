#!/bin/bash
# usage (GPU box): tools/pmc_sq.sh [pattern]  -> SQ issue/wait counters per kernel (separate --pmc pass, kernel-trace only)
PAT=${1:-.}
R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_sq && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d /tmp/pmc_sq -o p -- python $R/tools/layer_time.py > /dev/null 2>&1
cd $R; python tools/pmc_summary.py /tmp/pmc_sq/p_results.db 2>&1 | grep -E "$PAT"
