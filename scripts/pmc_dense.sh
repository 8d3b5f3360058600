#!/bin/bash
# PMC passes over scripts/kbench_dense.py (one variant, child mode): per-kernel counter averages -> gpurun_out/pmc_dense.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD
mkdir -p gpurun_out
: > gpurun_out/pmc_dense.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  d=/tmp/pmcd_$RANDOM
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --output-format csv -d $d -o p -- python $R/scripts/kbench_dense.py --child pmc > /dev/null 2>&1)
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" >> gpurun_out/pmc_dense.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
    if "k_qd_" not in k: continue
    g = row.get("Grid_Size", "")
    acc[(k, g)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for (k, g), cs in sorted(acc.items()):
    print(k, "grid", g, " ".join("%s=%.3g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
PY
done
cat gpurun_out/pmc_dense.txt
