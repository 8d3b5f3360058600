"""Fold gpurun_out/parity_r06/*.json (one file per config, written by tests/test_gpu_parity_*.py on the GPU box) into profiles/parity_r06.json."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles", "parity_r06.json")
data = json.load(open(out)) if os.path.exists(out) else {}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "parity_r06", "*.json"))):
    data.update(json.load(open(f)))
json.dump(data, open(out, "w"), indent=1, sort_keys=True)
print(len(data), "configs:", ", ".join(sorted(data)))
