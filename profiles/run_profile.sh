#!/bin/bash
# Profile recipe for the scene + match kernels (run on the GPU box from the repo root):
#   bash profiles/run_profile.sh <tag>
# Writes rocprofv3 CSVs under gpurun_out/prof_<tag>/ ; copy the *_kernel_stats.csv and the
# PMC summaries you want judged into profiles/.
set -e
TAG=${1:-r1}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
# 1) kernel trace + stats of the bench command without its CPU-baseline and driver (e2e) legs: the e2e leg launches the
#    same scene kernels on 256-frame micro-batches, which would be averaged into the headline kernel's row
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- \
    python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu --no-e2e > $OUT/bench_under_trace.json 2> $OUT/trace.err
# 2) PMC passes, one counter group per run (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- \
    python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --match-steps 3 > $OUT/bench_under_fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- \
    python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --match-steps 3 > $OUT/bench_under_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -50
