#!/usr/bin/env python
"""rocprofv3 --kernel-trace --stats CSV  ->  markdown summary for profiles/.
usage: prof_summary.py <kernel_stats.csv> <out.md> <title> [note ...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = open(sys.argv[2], "w")
out.write("# %s\n\n" % sys.argv[3])
for n in sys.argv[4:]:
    out.write(n + "\n")
out.write("\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
for r in rows[:45]:
    out.write("| `%s` | %s | %.3f | %.1f | %s |\n" % (r["Name"][:90], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
