"""print the per-kernel table of a bench.py --detail file:  python scripts/show_kernels.py gpurun_out/quick_c2.json [top]"""
import json
import sys
d = json.load(open(sys.argv[1]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for w, s in d["sections"].items():
    print(w, s["value"], s["ms_per_step"])
    ks = sorted(s.get("kernels", {}).items(), key=lambda kv: -kv[1]["ms_per_step"])
    for k, v in ks[:top]:
        print("  %-48s %6.3f ms  x%-4.1f %7.1f us  %7.1f GB/s" % (k[:48], v["ms_per_step"], v["launches_per_step"], v["avg_us"], v.get("GBps") or 0))
