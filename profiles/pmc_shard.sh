#!/bin/bash
# Kernel trace + counter passes of the lookup that keeps the top-k on rank 0's 1/8 shard, once per shape
# (run on the GPU box from the repo root):  bash profiles/pmc_shard.sh <tag> [shapes: wave block auto]
# Output gpurun_out/pmc_match_<tag>/shard8_<shape>/...; summarise with profiles/summarize_match.py <tag>
TAG=${1:-r5}; shift
SH=${@:-wave block}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_match_$TAG
mkdir -p $OUT
cd /tmp
for s in $SH; do
  case $s in wave) export TVZ_SHAPE=0x400;; block) export TVZ_SHAPE=0x800;; *) export TVZ_SHAPE=0;; esac
  w=shard8_$s
  mkdir -p $OUT/$w
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w/trace -o t -- \
      python3 $REPO/profiles/match_workloads.py shard8 24 > $OUT/$w/trace.json 2> $OUT/$w/trace.err
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
             "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
             "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$w/p$i -o p -- \
        python3 $REPO/profiles/match_workloads.py shard8 8 > $OUT/$w/p$i.json 2> $OUT/$w/p$i.err
    echo "$w pass $i done"
  done
done
