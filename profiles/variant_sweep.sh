#!/bin/bash
export TVZ_ALLOW_DIAGNOSTIC=1   # variants/libtvz_*.so are diagnostic builds (tvz_version() < 0): only these scripts may load them
# Experiment harness: time one matcher workload with each prebuilt library variant under variants/
# (built with TVZ_CXXFLAGS=-D...), plus L2 hit/miss and fabric-read counters of the sweep kernel.
#   bash profiles/variant_sweep.sh <workload> <variant>...
W=$1; shift
REPO=$(pwd)
export TMPDIR=/tmp
for v in "$@"; do
  export TVZ_LIB=$REPO/variants/libtvz_$v.so     # selected by tvidz_amd/_lib.py; the product library is never touched
  OUT=$REPO/gpurun_out/variant_$v
  mkdir -p $OUT
  echo "== variant $v"
  python3 profiles/match_workloads.py $W 2>/dev/null | cut -c1-140
  (cd /tmp && rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -o p -- \
      python3 $REPO/profiles/match_workloads.py $W 4 > $OUT/tcc.json 2> $OUT/tcc.err)
  (cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- \
      python3 $REPO/profiles/match_workloads.py $W 4 > $OUT/fetch.json 2> $OUT/fetch.err)
  python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:30]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
    if not n.startswith(("ts_", "void ts_")): continue
    print("  ", n, {k: round(sum(v) / len(v)) for k, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
done
