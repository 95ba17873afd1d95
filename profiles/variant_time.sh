#!/bin/bash
export TVZ_ALLOW_DIAGNOSTIC=1   # variants/libtvz_*.so are diagnostic builds (tvz_version() < 0): only these scripts may load them
# Time matcher workloads with prebuilt library variants under variants/ (experiments only).  The
# variant is selected with TVZ_LIB (tvidz_amd/_lib.py): the product library is never touched.
#   bash profiles/variant_time.sh "<workload> ..." <variant>...      (variant "product" = tvidz_amd/libtvz.so)
WL=$1; shift
for v in "$@"; do
  for w in $WL; do
    if [ "$v" = product ]; then L=""; else L=$(pwd)/variants/libtvz_$v.so; fi
    echo -n "$v $w: "; TVZ_LIB=$L python3 profiles/match_workloads.py $w 14 2>/dev/null | tail -1 | cut -c1-120
  done
done
