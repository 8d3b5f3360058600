#!/usr/bin/env python
"""Per-step kernel breakdown from a rocprofv3 --kernel-trace CSV of bench.py: aggregates one steady-state step
(between the last two nll_loss_forward launches).  usage: step_breakdown.py trace.csv [topN]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "nll_loss_forward" in n]
a, b = idx[-3], idx[-2]
seg = rows[a:b]
wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
agg = collections.OrderedDict()
for r in seg:
    n = r["Kernel_Name"][:100]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    e = agg.setdefault(n, [0, 0.0]); e[0] += 1; e[1] += d
busy = sum(v[1] for v in agg.values())
print("step wall %.1f us, kernels %d, busy %.1f us" % (wall, len(seg), busy))
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%8.1f us %4d  %s" % (v[1], v[0], n))
