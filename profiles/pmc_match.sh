#!/bin/bash
# Kernel trace + counter passes over the corpus matcher (run on the GPU box from the repo root):
#   bash profiles/pmc_match.sh <tag> [workloads...]
# One rocprofv3 run per (workload, counter group); the program comes directly after `--`
# (no env/bash hop), counters never share a run with other trace domains than the kernel trace.
# Output: gpurun_out/pmc_match_<tag>/<workload>/<pass>/...; summarise with profiles/summarize_match.py
TAG=${1:-r2}; shift
WL=${@:-join q1_100k q1_5k tile topk}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_match_$TAG
mkdir -p $OUT
cd /tmp
for w in $WL; do
  mkdir -p $OUT/$w
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w/trace -o t -- \
      python3 $REPO/profiles/match_workloads.py $w 12 > $OUT/$w/trace.json 2> $OUT/$w/trace.err
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
             "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
             "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$w/p$i -o p -- \
        python3 $REPO/profiles/match_workloads.py $w 6 > $OUT/$w/p$i.json 2> $OUT/$w/p$i.err
    echo "$w pass $i done"
  done
done
find $OUT -name "*_counter_collection.csv" | wc -l
