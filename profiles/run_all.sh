#!/bin/bash
# Everything measured for a round, in one gpurun call (run on the GPU box from the repo root):
#   bash profiles/run_all.sh <tag>
# Raw output lands in gpurun_out/ (scratch); `python profiles/collect.py <tag>` (run afterwards, in the
# build container) copies the summaries that are cited into profiles/.
TAG=${1:-r2}
REPO=$(pwd)
OUT=$REPO/gpurun_out/final_$TAG
mkdir -p $OUT
echo "== bench (un-profiled)";            python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo rc=$?
echo "== scene kernel trace + FETCH/WRITE"; bash profiles/run_profile.sh $TAG > $OUT/run_profile.log 2>&1; echo rc=$?
echo "== matcher trace + counters";       bash profiles/pmc_match.sh $TAG index index1 join q1_100k q1_5k tile shard8 > $OUT/pmc_match.log 2>&1; echo rc=$?
echo "== predicted scaling";              python profiles/predict_scaling.py 4096 2>/dev/null | tail -1 > $OUT/predicted_scaling.json; python profiles/predict_scaling.py 1024 2>/dev/null | tail -1 >> $OUT/predicted_scaling.json
echo "== find_duplicates latency (C ABI, no Python)"
bash profiles/fdl.sh > $OUT/find_dup_latency.txt
echo "== driver, N concurrent uploads (1080p Y4M in RAM -> verdicts)"
for a in "1 1024 1" "16 256 16" "32 256 16" "64 256 16" "16 512 16" "64 512 16"; do
  python profiles/e2e_service.py $a 256 64 2>/dev/null | tail -1
done > $OUT/e2e_service.txt
echo "== tile vs join vs q1 grid";        python profiles/ab_match_join.py 2>/dev/null > $OUT/match_ab.txt
ls -la $OUT
