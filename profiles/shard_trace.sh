#!/bin/bash
# kernel timeline of the sharded pipeline on a 1/N shard:  bash profiles/shard_trace.sh [N] [Q]
set -e
REPO=$(pwd); export TMPDIR=/tmp; OUT=$REPO/gpurun_out/shard_trace; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $REPO/profiles/shard_trace.py ${1:-8} ${2:-4096} > $OUT/log.txt 2>&1
python3 $REPO/profiles/shard_timeline.py $OUT/t_kernel_trace.csv 200 > $OUT/timeline_${1:-8}_${2:-4096}.txt
