#!/bin/bash
export TVZ_ALLOW_DIAGNOSTIC=1   # variants/libtvz_*.so are diagnostic builds (tvz_version() < 0): only these scripts may load them
# HBM traffic of one matcher workload, two counter passes (FETCH_SIZE, WRITE_SIZE), optionally with a
# variant library:  [TVZ_LIB=variants/libtvz_x.so] bash profiles/pmc_traffic.sh <workload> <name>
W=${1:-index}; NAME=${2:-product}
REPO=$(pwd); export TMPDIR=/tmp; OUT=$REPO/gpurun_out/pmc_traffic_$NAME; mkdir -p $OUT; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- \
      python3 $REPO/profiles/match_workloads.py $W 6 > $OUT/$c.json 2> $OUT/$c.err
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{sys.argv[1]}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:40]][c].append(float(r["Counter_Value"]))
for k, d in agg.items():
    f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])) * 1024 * 2 / 1e6
    w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"])) * 1024 / 1e6
    print(f"{k:42s} read_MB={f:9.1f} write_MB={w:9.1f} total_MB={f + w:9.1f} launches={len(d['FETCH_SIZE'])}")
PY
