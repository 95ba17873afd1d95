#!/bin/bash
export TVZ_ALLOW_DIAGNOSTIC=1   # variants/libtvz_*.so are diagnostic builds (tvz_version() < 0): only these scripts may load them
# Everything measured for a round, in one gpurun call (run on the GPU box from the repo root):
#   bash profiles/variant_build.sh stamp -DTVZ_IX_STAMP      (build container, beforehand)
#   bash profiles/run_all.sh <tag>
# Raw output lands in gpurun_out/ (scratch); `python profiles/collect.py <tag>` (run afterwards, in the
# build container) copies the summaries that are cited into profiles/.  Every step writes its own file
# and a progress line, so a long run never looks hung.
TAG=${1:-r5}
PART=${2:-all}        # a = bench + traces + counters, b = everything else (a gpurun call is capped at 20 minutes)
REPO=$(pwd)
OUT=$REPO/gpurun_out/final_$TAG
mkdir -p $OUT
if [ "$PART" != "b" ]; then
echo "== bench (un-profiled)";            python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo rc=$?
echo "== scene kernel trace + FETCH/WRITE"; bash profiles/run_profile.sh $TAG > $OUT/run_profile.log 2>&1; echo rc=$?
echo "== matcher trace + counters"
for w in topk shard8 index index1 join q1_100k q1_5k tile; do bash profiles/pmc_match.sh $TAG $w >> $OUT/pmc_match.log 2>&1; echo "  $w done"; done
echo "== the 1/8 shard's lookup, both shapes (one wave per query / one block per query)"
bash profiles/pmc_shard.sh $TAG wave block >> $OUT/pmc_match.log 2>&1; echo "  shapes done"
echo "== index rebuild, 100k rows (kernel trace)"; bash profiles/trace_one.sh rebuild 10 > $OUT/rebuild_trace.txt 2>&1
fi
if [ "$PART" != "a" ]; then
echo "== predicted scaling (100k x 4096 / 1024, 800k x 4096)"
python profiles/predict_scaling.py 4096 2>/dev/null | tail -1 > $OUT/predicted_scaling.json; python profiles/predict_scaling.py 1024 2>/dev/null | tail -1 >> $OUT/predicted_scaling.json
python profiles/predict_scaling.py 4096 800000 2>/dev/null | tail -1 >> $OUT/predicted_scaling.json
echo "== the 1/8 shard: one batch alone per shape, and as a stream of batches"
python profiles/ab_wave.py 40 2 2>/dev/null | tail -1 > $OUT/ab_wave.txt; python profiles/ab_wave.py 40 5 2>/dev/null | tail -1 >> $OUT/ab_wave.txt
python profiles/shard_pipe.py 8 200 2>/dev/null | tail -1 >> $OUT/ab_wave.txt
echo "== fused lookup + top-k vs match -> top-k"; python profiles/topk_probe.py 4096 100000 8 2>/dev/null | tail -1 > $OUT/topk_probe.txt
echo "== find_duplicates latency (C ABI, no Python)"
bash profiles/fdl.sh > $OUT/find_dup_latency.txt 2>&1
echo "== lookups while the index is rebuilt (C, no Python)"
gcc -O1 -pthread -Iinclude tests/rebuild_latency.c -o /tmp/rl -Ltvidz_amd -ltvz -lm -Wl,-rpath,$REPO/tvidz_amd -Wl,-rpath,/opt/rocm/lib \
  && TVZ_DEBUG=1 /tmp/rl 2>&1 | grep -v "^slow" > $OUT/rebuild_latency.txt
echo "== index lookup phase stamps (diagnostic build)"
if [ -f variants/libtvz_stamp.so ]; then
  for w in index shard8; do for m in "" topk; do TVZ_LIB=$REPO/variants/libtvz_stamp.so python3 profiles/ix_stamps.py $w $m 2>/dev/null | tail -1; done; done > $OUT/ix_stamps.txt
  TVZ_LIB=$REPO/variants/libtvz_stamp.so python3 profiles/wq_stamps.py 2 2>/dev/null | tail -1 >> $OUT/ix_stamps.txt
fi
echo "== driver, N concurrent uploads (Y4M in RAM -> verdicts): 1080p, 4K, 8 shards"
for a in "1 1024 1 256 64 0" "16 256 16 256 64 0" "64 256 16 256 64 0" "16 512 16 256 64 0" "64 512 16 256 64 0" \
         "64 64 16 64 128 0 2160 3840" "64 128 16 64 128 0 2160 3840" "64 256 16 256 64 0 1080 1920 8"; do
  python profiles/e2e_service.py $a 2>/dev/null | tail -1
done > $OUT/e2e_service.txt
echo "== index at 1 M rows";              python profiles/scale_probe.py 2>/dev/null | tail -1 > $OUT/scale_probe.txt
echo "== differential soak, 120 s";       python profiles/fuzz_parity.py 120 4242 > $OUT/fuzz_parity.txt 2>&1; tail -1 $OUT/fuzz_parity.txt
echo "== tile vs join vs q1 grid";        python profiles/ab_match_join.py 2>/dev/null > $OUT/match_ab.txt
fi
ls -la $OUT
