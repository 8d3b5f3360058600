#!/bin/bash
# HBM traffic of the conv kernels from the rocprofv3 PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2), no tracing domains combined with --pmc.  Run on the GPU box from the repo root:
#   bash scripts/pmc_traffic.sh [workload]        -> gpurun_out/traffic_<workload>.json (+ raw CSVs under gpurun_out/pmc_*)
W=${1:-c2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o t -- python $R/bench.py --workload $W --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-graph > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python - <<PY
import csv, collections, glob, json
out = collections.defaultdict(lambda: dict(fetch_kb=0.0, write_kb=0.0, n_f=0, n_w=0))
for c, key, cnt in (("FETCH_SIZE", "fetch_kb", "n_f"), ("WRITE_SIZE", "write_kb", "n_w")):
    f = glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % c, recursive=True)
    if not f:
        continue
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if not (k.startswith("void k_") or k.startswith("k_")):
            continue
        k = k.replace("void ", "").split("(")[0]
        out[k][key] += float(r["Counter_Value"]); out[k][cnt] += 1
res = {}
for k, v in out.items():
    if not v["n_f"] or not v["n_w"]:
        continue
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts a wide coalesced stream at 1/2 (MI355X_MICROARCH.md, HBM)
    fetch = 2.0 * v["fetch_kb"] / v["n_f"] * 1024.0
    write = v["write_kb"] / v["n_w"] * 1024.0
    res[k] = dict(bytes_per_launch=int(fetch + write), fetch_bytes_x2=int(fetch), write_bytes=int(write), launches_sampled=v["n_f"])
json.dump(res, open("gpurun_out/traffic_$W.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(res.items()):
    print("%-40s %12d B/launch (fetch x2 %d, write %d)" % (k, v["bytes_per_launch"], v["fetch_bytes_x2"], v["write_bytes"]))
PY
