#!/bin/bash
# kernel trace of profiles/scale_probe.py (1 M rows): which build kernels run and how long
REPO=$(pwd); export TMPDIR=/tmp; OUT=$REPO/gpurun_out/trace_scale; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $REPO/profiles/scale_probe.py > $OUT/run.json 2> $OUT/run.err
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "")[:34]
    if n.startswith(("ix_", "void ix_")):
        print(f'{n:36s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:9.1f} min_us={float(r["MinNs"])/1e3:9.1f}')
PY
