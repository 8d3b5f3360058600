#!/usr/bin/env python
"""Harvest inputs where the kernels' tanh (mn_tanh_f32: the correctly rounded fp32 value, evaluated in fp64) and torch-CPU tanh (what the reference runs: MKL VML
vsTanh, HA mode, in this torch build) differ, for tests/golden/tanh_device_vs_cpu.json (run on a GPU box: `gpurun -- python scripts/make_tanh_fixture.py`; output
in gpurun_out/)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rng = np.random.default_rng(20260922)
x = np.concatenate([rng.standard_normal(1 << 21).astype(np.float32) * s for s in (0.05, 0.3, 1.0, 2.5)])
xt = torch.from_numpy(x)
cpu = torch.tanh(xt).numpy().view(np.int32)
import ctypes as C
from micronet_amd import _lib
yd = torch.empty(x.size, device="cuda")
lib = _lib.get_lib()
assert lib.mn_tanh_f32(C.c_void_p(xt.cuda().data_ptr()), C.c_void_p(yd.data_ptr()), x.size, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
dev = yd.cpu().numpy().view(np.int32)
cr = np.tanh(x.astype(np.float64)).astype(np.float32).view(np.int32)
print("kernel tanh vs correctly rounded (numpy fp64):", int((dev != cr).sum()), " torch-CPU vs correctly rounded:", int((cpu != cr).sum()))
d = dev.astype(np.int64) - cpu.astype(np.int64)
mis = np.nonzero(d)[0]
print("elements", x.size, "mismatches", mis.size, "max |ulp|", int(np.abs(d).max()) if mis.size else 0)
pick = mis[:: max(1, mis.size // 96)][:96]
out = dict(note="inputs (float32 bit patterns) where the kernels' correctly rounded tanh and torch-CPU tanh (MKL VML vsTanh HA) differ; ROCm 7.2.0 / torch 2.10.0+rocm7.0, gfx950",
           elements=int(x.size), mismatches=int(mis.size), max_ulp=int(np.abs(d).max()) if mis.size else 0,
           x_bits=[int(v) for v in x.view(np.int32)[pick]], cpu_bits=[int(v) for v in cpu[pick]], dev_bits=[int(v) for v in dev[pick]])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "tanh_device_vs_cpu.json"), "w"), indent=0)
