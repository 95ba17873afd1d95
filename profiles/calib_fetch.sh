#!/bin/bash
# FETCH_SIZE per access for streaming vs random reads (run on the GPU box from the repo root); prints a table.
REPO=$(pwd); export TMPDIR=/tmp; OUT=$REPO/gpurun_out/calib_fetch; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 profiles/calib_fetch.hip -o /tmp/calib_fetch || exit 1
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o c -- /tmp/calib_fetch > $OUT/run.json 2> $OUT/f.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum --output-format csv -d $OUT/r -o c -- /tmp/calib_fetch > /dev/null 2> $OUT/r.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/s -o c -- /tmp/calib_fetch > /dev/null 2> $OUT/s.err
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
meta = json.loads(open(out + "/run.json").read().strip().splitlines()[-1])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in acc.items():
    fs = sum(d.get("FETCH_SIZE", [0])) / max(len(d.get("FETCH_SIZE", [1])), 1) * 1024.0     # KB -> bytes
    rq = sum(d.get("TCC_EA0_RDREQ_sum", [0])) / max(len(d.get("TCC_EA0_RDREQ_sum", [1])), 1)
    by_size = sum(w * sum(d.get(f"TCC_EA0_RDREQ_{w}B_sum", [0])) / max(len(d.get(f"TCC_EA0_RDREQ_{w}B_sum", [1])), 1) for w in (32, 64, 128))
    if "stream16" in k:
        res["stream16"] = {"bytes_read": meta["stream16_bytes"], "FETCH_SIZE_bytes": fs, "FETCH/bytes": fs / meta["stream16_bytes"], "RDREQ": rq, "bytes_by_request_size/bytes": by_size / meta["stream16_bytes"]}
    elif "rand16" in k:
        res["rand16"] = {"reads": meta["rand16_reads"], "FETCH_SIZE_bytes": fs, "FETCH_bytes_per_read": fs / meta["rand16_reads"], "RDREQ_per_read": rq / meta["rand16_reads"], "bytes_by_request_size_per_read": by_size / meta["rand16_reads"]}
    elif "rand2x12" in k:
        res["rand2x12"] = {"reads": meta["rand2x12_reads"], "FETCH_SIZE_bytes": fs, "FETCH_bytes_per_read": fs / meta["rand2x12_reads"], "RDREQ_per_read": rq / meta["rand2x12_reads"], "bytes_by_request_size_per_read": by_size / meta["rand2x12_reads"]}
print(json.dumps(res))
PY
