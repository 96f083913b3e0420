#!/bin/bash
# usage (GPU box): tools/kprof.sh LIBTAG [pattern]  -> avg kernel times with tools/_lib_LIBTAG.so (or default lib)
TAG=$1; PAT=${2:-.}
R=$PWD
if [ "$TAG" != "default" ]; then export MDGAT_HIP_LIB=$R/tools/_lib_$TAG.so; fi
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kp_$TAG && rocprofv3 --kernel-trace --stats -d /tmp/kp_$TAG -o k -- python $R/tools/${KPROF_SCRIPT:-layer_time.py} > /dev/null 2>&1
cd $R; echo "== $TAG"; python tools/rocpd_summary.py /tmp/kp_$TAG/k_results.db 2>&1 | grep -E "$PAT" | cut -c1-50,110-170
