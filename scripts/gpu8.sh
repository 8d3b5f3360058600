mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof8 -o kb -- python $R/scripts/kbench.py --algos 3 --layers L4,L7 --iters 5 > $R/gpurun_out/kb8.log 2>&1
f=$(find $R/gpurun_out/prof8 -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-160
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/prof8pmc -o kb -- python $R/scripts/kbench.py --algos 3 --layers L4,L7 --iters 2 > $R/gpurun_out/kb8pmc.log 2>&1
f=$(find $R/gpurun_out/prof8pmc -name "*counter_collection.csv" | head -1); echo $f; head -3 "$f" | cut -c1-300
python - <<PY
import csv,collections
f="$f"
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'k_kk' in k or 'k_pw' in k:
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
