#!/bin/bash
# Instruction-fetch counters of the lookup kernels on rank 0's 1/8 shard (run on the GPU box from the repo root):
#   [TVZ_LIB=variants/libtvz_x.so] TVZ_SHAPE=0x400|0x800 bash profiles/pmc_icache.sh <name>
export TVZ_ALLOW_DIAGNOSTIC=1
NAME=${1:-product}
REPO=$(pwd); export TMPDIR=/tmp; OUT=$REPO/gpurun_out/pmc_icache_$NAME; mkdir -p $OUT; cd /tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- \
      python3 $REPO/profiles/match_workloads.py shard8 8 > $OUT/p$i.json 2> $OUT/p$i.err
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{sys.argv[1]}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:28]
        if "match" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    der = {}
    if m.get("SQC_ICACHE_REQ"):
        der["icache_miss_rate"] = round((m.get("SQC_ICACHE_MISSES", 0)) / m["SQC_ICACHE_REQ"], 3)
        der["icache_miss_dup_rate"] = round((m.get("SQC_ICACHE_MISSES_DUPLICATE", 0)) / m["SQC_ICACHE_REQ"], 3)
    if m.get("SQ_WAVE_CYCLES"):
        der["wait_inst_share"] = round(m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], 3)
        der["wait_any_share"] = round(m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"], 3)
    print(json.dumps({"kernel": k, "name": sys.argv[1].split("_")[-1], **{c: round(v) for c, v in m.items()}, **der}))
PY
