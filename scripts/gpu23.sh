export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc23a -o t -- python $R/scripts/kbench.py --layers L4 --algos 3 --scheme sign8 --iters 3 > $R/gpurun_out/pmc23a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc23b -o t -- python $R/scripts/kbench.py --layers L4 --algos 3 --scheme sign8 --iters 3 > $R/gpurun_out/pmc23b.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("pmc23a", "pmc23b"):
    f = glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_kk"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(d, k, {c: "%.3g" % (sum(v) / len(v)) for c, v in cs.items()})
PY
