#!/bin/bash
# SQ counter passes over the corpus matcher (run on the GPU box): bash profiles/pmc_match.sh <tag>
set -e
TAG=${1:-m1}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_match_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA \
  --output-format csv -d $OUT/a -o a -- python3 $REPO/profiles/tune_match.py 100000 1024 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES \
  --output-format csv -d $OUT/b -o b -- python3 $REPO/profiles/tune_match.py 100000 1024 > $OUT/b.log 2>&1
python3 - <<PY
import csv, collections
for sub in ("a","b"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open("$OUT/%s/%s_counter_collection.csv"%(sub,sub))):
        if "ts_match_tile" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()):
        print(k, len(v), sum(v)/len(v))
PY
