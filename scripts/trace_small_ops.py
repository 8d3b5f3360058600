"""Where do the small ATen launches of a workload's step come from?  One eager step under torch.profiler with Python stacks; prints every aten::copy_ / fill_ / add / zero_ /
clone / mul / div call site (count per step).  usage: python scripts/trace_small_ops.py c5"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from micronet_amd.train import build_model, make_optimizer, synth_batch, prefetch_weight_path, bump_bn_counters  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "c5"
arch, scheme, kw, wd = bench.WORKLOADS[key]
Q = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
model = Q.prepare(build_model(arch), inplace=True, **kw).cuda().train()
opt = make_optimizer(model, 0.01, wd)
opt.capturable = True          # the graphed step's optimizer path
x, y = synth_batch(256, device="cuda")


def step():
    prefetch_weight_path(model)
    bump_bn_counters(model)
    out = model(x)
    loss = torch.nn.functional.cross_entropy(out, y)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU], with_stack=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step()
torch.cuda.synchronize()
want = ("aten::copy_", "aten::fill_", "aten::add", "aten::add_", "aten::zero_", "aten::clone", "aten::mul", "aten::div", "aten::div_", "aten::cat", "aten::sub", "aten::to",
        "aten::_to_copy", "aten::zeros", "aten::ones", "aten::full", "aten::contiguous")
sites = {}
for ev in prof.events():
    if ev.name in want:
        st = [s for s in (ev.stack or []) if "site-packages/torch/nn/modules/module.py" not in s and "torch/autograd/function.py" not in s][:4]
        k = (ev.name, tuple(st))
        sites[k] = sites.get(k, 0) + 1
for (name, st), n in sorted(sites.items(), key=lambda kv: -kv[1])[:40]:
    print(n, name, " <- ".join(s.split("/")[-1][:90] for s in st))
