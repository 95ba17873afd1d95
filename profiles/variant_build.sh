#!/bin/bash
# Build -D variants of libtvz.so into variants/ (git-ignored; they travel to the GPU box with gpurun):
#   bash profiles/variant_build.sh name1 "-DX=1 -DY=2" name2 "-DZ=3" ...
# Every variant is marked -DTVZ_DIAGNOSTIC=1: its tvz_version() is negative and tvidz_amd/_lib.py loads it only
# with TVZ_ALLOW_DIAGNOSTIC=1 (which the profile scripts that take TVZ_LIB export themselves).
export TVZ_DIAGNOSTIC=1
mkdir -p variants
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -Wno-unused-function \
    -Iinclude -Itvidz_amd/csrc -DTVZ_DIAGNOSTIC=1 $f -o variants/libtvz_$n.so tvidz_amd/csrc/tvz_api.hip tvidz_amd/csrc/tvz_scene.hip \
    tvidz_amd/csrc/tvz_match.hip tvidz_amd/csrc/tvz_comm.hip -ldl &
done
wait
ls -la variants/
