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
# per-step launch counts: the loss kernel runs once per training step (graph replays included)
steps = next((int(r["Calls"]) for r in rows if r["Name"].startswith("k_ce_fwd") or "nll_loss_forward" in r["Name"]), 0)          # (k_ce_fwd: the in-tree loss of round 6)
if steps:
    lib = sum(int(r["Calls"]) for r in rows if r["Name"].replace("void ", "").startswith(("k_", "mn_")))
    tot = sum(int(r["Calls"]) for r in rows)
    ns = sum(int(r["TotalDurationNs"]) for r in rows)
    ns_other = sum(int(r["TotalDurationNs"]) for r in rows if not r["Name"].replace("void ", "").startswith(("k_", "mn_")))
    out.write("\n%d traced steps: %.1f kernel launches per step, of which %.1f are not this library's (ATen fills / copies / "
              "counters, what is left of the stock operators: %.2f %% of the kernel time); %.3f ms of kernel time per step.\n" % (steps, tot / steps, (tot - lib) / steps, 100.0 * ns_other / ns, ns / steps / 1e6))
