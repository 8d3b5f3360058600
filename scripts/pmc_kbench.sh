#!/bin/bash
# PMC counters of a micro-benchmark command, one counter group per pass (no tracing domains with --pmc).  On the GPU box, from the repo root:
#   bash scripts/pmc_kbench.sh <tag> "<counter group 1>" "<counter group 2>" ... -- <command ...>
# -> gpurun_out/pmck_<tag>.txt : average counter value per launch, per kernel whose name starts with k_
TAG=$1; shift
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "${GROUPS_[@]}"; do
  rm -rf $R/gpurun_out/pmck_${TAG}_$i
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmck_${TAG}_$i -o t -- "$@" > $R/gpurun_out/pmck_${TAG}_$i.log 2>&1
  i=$((i+1))
done
cd $R
python - <<PY > gpurun_out/pmck_$TAG.txt
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/pmck_${TAG}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not k.startswith("k_"):
            continue
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        s, n = acc[k][c]
        print("    %-28s %16.1f  (avg of %d launches)" % (c, s / n, n))
PY
cat gpurun_out/pmck_$TAG.txt
