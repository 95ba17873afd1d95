#!/bin/bash
# per-kernel time of the matcher at C=100000, Q=1024:  bash profiles/trace_match.sh
set -e
REPO=$(pwd); export TMPDIR=/tmp; OUT=$REPO/gpurun_out/trace_match; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $REPO/profiles/tune_match.py ${1:-100000} ${2:-1024} > $OUT/log.txt 2>&1
grep -E "ts_|fillBuffer" $OUT/t_kernel_stats.csv | cut -c1-60,100-400
