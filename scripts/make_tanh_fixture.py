#!/usr/bin/env python
"""Harvest inputs where the device tanhf (ocml, what the DoReFa weight-quantizer kernels call) and torch-CPU tanh (Sleef, what the reference runs) differ,
for tests/golden/tanh_device_vs_cpu.json (run on a GPU box: `gpurun -- python scripts/make_tanh_fixture.py`; output in gpurun_out/)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(20260922)
x = np.concatenate([rng.standard_normal(1 << 21).astype(np.float32) * s for s in (0.05, 0.3, 1.0, 2.5)])
xt = torch.from_numpy(x)
cpu = torch.tanh(xt).numpy().view(np.int32)
dev = torch.tanh(xt.cuda()).cpu().numpy().view(np.int32)
d = dev.astype(np.int64) - cpu.astype(np.int64)
mis = np.nonzero(d)[0]
print("elements", x.size, "mismatches", mis.size, "max |ulp|", int(np.abs(d).max()) if mis.size else 0)
pick = mis[:: max(1, mis.size // 96)][:96]
out = dict(note="inputs (float32 bit patterns) where device tanhf and torch-CPU tanh differ; ROCm 7.2.0 / torch 2.10.0+rocm7.0, gfx950",
           elements=int(x.size), mismatches=int(mis.size), max_ulp=int(np.abs(d).max()) if mis.size else 0,
           x_bits=[int(v) for v in x.view(np.int32)[pick]], cpu_bits=[int(v) for v in cpu[pick]], dev_bits=[int(v) for v in dev[pick]])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "tanh_device_vs_cpu.json"), "w"), indent=0)
