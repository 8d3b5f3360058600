import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import importlib
from micronet_amd.train import GraphedTrainStep, build_model, make_optimizer, synth_batch
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group(os.environ.get("MN_DIST_BACKEND", "gloo"), rank=rank, world_size=world)
q = importlib.import_module("micronet.compression.quantization.wbwtab.quantize")
model = q.prepare(build_model("nin_gc"), inplace=True, A=2, W=3).cuda().train()
opt = make_optimizer(model, 0.01, 0.0)
x, y = synth_batch(64, seed=1 + rank, device="cuda")
g = GraphedTrainStep(model, opt, x, y)
def t(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(rank, "graph_a %.2f ms" % t(g.graph_a.replay), "allreduce %.2f ms" % t(lambda: dist.all_reduce(g.flat)), "graph_b %.2f ms" % t(g.graph_b.replay), "step %.2f ms" % t(g.step), flush=True)
def step_sync():
    g.graph_a.replay(); torch.cuda.current_stream().synchronize(); dist.all_reduce(g.flat); g.graph_b.replay()
print(rank, "step with a stream sync before the all-reduce %.2f ms" % t(step_sync), flush=True)
def step_sync2():
    g.graph_a.replay(); torch.cuda.synchronize(); dist.all_reduce(g.flat); torch.cuda.synchronize(); g.graph_b.replay()
print(rank, "step with device syncs both sides %.2f ms" % t(step_sync2), flush=True)
tmp = torch.zeros_like(g.flat)
print(rank, "allreduce on a plain tensor %.2f ms" % t(lambda: dist.all_reduce(tmp)), flush=True)
