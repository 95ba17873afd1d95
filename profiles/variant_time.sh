#!/bin/bash
# Time one matcher workload with each prebuilt library variant under variants/ (experiments only).
#   bash profiles/variant_time.sh <workload> <variant>...
W=$1; shift
cp tvidz_amd/libtvz.so /tmp/libtvz_keep.so
for v in "$@"; do
  cp variants/libtvz_$v.so tvidz_amd/libtvz.so
  echo "== variant $v"; python3 profiles/match_workloads.py $W 2>/dev/null | cut -c1-110
done
cp /tmp/libtvz_keep.so tvidz_amd/libtvz.so
