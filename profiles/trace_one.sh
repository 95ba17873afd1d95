#!/bin/bash
# Kernel trace only (no counters) of one matcher workload: bash profiles/trace_one.sh <workload> [reps]
# Prints the per-kernel average durations.  Output under gpurun_out/trace_<workload>/.
W=${1:-shard8}; REPS=${2:-12}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/trace_$W
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- \
    python3 $REPO/profiles/match_workloads.py $W $REPS > $OUT/run.json 2> $OUT/run.err
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "")[:34]
    print(f'{n:36s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:9.1f} min_us={float(r["MinNs"])/1e3:9.1f}')
PY
