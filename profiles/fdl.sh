#!/bin/bash
# build + run the C latency probe (run on the GPU box from the repo root)
REPO=$(pwd)
gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude profiles/find_dup_latency.c -o /tmp/fdl -Ltvidz_amd -ltvz \
    -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$REPO/tvidz_amd -Wl,-rpath,/opt/rocm/lib && { /tmp/fdl 5000; /tmp/fdl 100000 1000; }
