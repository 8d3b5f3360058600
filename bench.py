#!/usr/bin/env python
"""QAT training-step throughput of the fake-quantized conv hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = forward + CrossEntropy + zero_grad + backward + Adam.step on one synthetic CIFAR-10-shaped batch
(mirror of the reference's wqaq/dorefa/main.py:77-82), data already resident in HBM.  Workload at every N:
BASELINE.json configs[1] -- nin_gc, wbwtab W-ternary / A-binary, batch 256 PER GPU (weak scaling), data-parallel with a
RCCL all-reduce of the gradients.  Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_16x16x4_f32 rate
NIN_GC_GFLOP_PER_IMG = 0.9068   # fwd+bwd, all nine convs
NIN_GC_MB_PER_IMG = 22.71       # fused-ideal fp32 activation traffic fwd+bwd

WORKLOADS = {
    # name: (arch, scheme module, prepare kwargs, weight decay)   (BASELINE.json configs)
    "c2": ("nin_gc", "wbwtab", dict(A=2, W=3), 0.0),
    "c1": ("nin_gc", "wqaq.dorefa", dict(a_bits=8, w_bits=8), 1e-5),
    "c1_w2a2": ("nin_gc", "wqaq.dorefa", dict(a_bits=2, w_bits=2), 1e-5),
    "c3": ("nin_gc", "wqaq.iao", dict(a_bits=8, w_bits=8, q_type=0, q_level=0, weight_observer=0, bn_fuse=True), 1e-5),
    "c4": ("resnet18", "wqaq.dorefa", dict(a_bits=2, w_bits=2), 1e-5),
    "c5": ("resnet18", "wqaq.iao", dict(a_bits=4, w_bits=4, q_type=0, q_level=0), 1e-5),
}
WORKLOAD_DESC = {
    "c2": "nin_gc CIFAR-10 wbwtab W-ternary/A-binary QAT, batch=256 per GPU (BASELINE configs[1])",
    "c1": "nin_gc CIFAR-10 DoReFa W8A8 QAT (BASELINE configs[0] scheme)",
    "c1_w2a2": "nin_gc CIFAR-10 DoReFa W2A2 QAT",
    "c3": "nin_gc CIFAR-10 IAO W8A8 per-channel + BN-fuse QAT (BASELINE configs[2])",
    "c4": "resnet18 CIFAR-10 DoReFa W2A2 QAT (BASELINE configs[3])",
    "c5": "resnet18 CIFAR-10 IAO W4A4 per-channel + quant_add QAT (BASELINE configs[4])",
}


class KernelProfiler:
    """Per-kernel timing with HIP events recorded by the library itself around every main kernel it launches (conv, BN+sign,
    pool), on the launch stream: mn_profile_enable / mn_profile_collect (include/micronet_hip.h)."""

    def __init__(self):
        from micronet_amd import _lib
        self._lib = _lib
        self.lib = _lib.get_lib()

    def start(self):
        self.lib.mn_profile_enable(1)

    def stop(self):
        """Call after torch.cuda.synchronize(): {kernel: dict(ms, bytes, launches)}."""
        buf = (self._lib.ProfEntry * 128)()
        n = self.lib.mn_profile_collect(buf, 128)
        self.lib.mn_profile_enable(0)
        return {buf[i].name.decode(): dict(ms=buf[i].total_ms, bytes=buf[i].bytes, launches=buf[i].launches) for i in range(n)}


def build(workload, device):
    import importlib
    from micronet_amd.train import build_model, make_optimizer
    arch, scheme, kw, wd = WORKLOADS[workload]
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    if scheme == "wbwtab" and os.environ.get("MN_BENCH_WBWTAB_KW"):      # A/B switches, e.g. "fuse_conv_bn=0,packed_activations=1"
        kw = dict(kw, **{k: bool(int(v)) for k, v in (it.split("=") for it in os.environ["MN_BENCH_WBWTAB_KW"].split(","))})
    model = quantize.prepare(build_model(arch), inplace=True, **kw).to(device)
    model.train()
    return model, make_optimizer(model, 0.01, wd)


def cpu_baseline(workload, batch, steps, threads=0):
    """The reference's algorithm on the host cores: the torch-CPU oracle ("port", bit-identical to the reference on CPU)."""
    from oracle import torch_oracle as TO
    from micronet_amd.train import build_model, synth_batch
    arch, scheme, kw, wd = WORKLOADS[workload]
    scheme_name = scheme.split(".")[-1]
    torch.set_num_threads(threads if threads > 0 else os.cpu_count())
    model = TO.prepare(build_model(arch), scheme_name, inplace=True, **kw).train()
    opt = TO.make_optimizer(model, 0.01, wd)
    x, y = synth_batch(batch)
    TO.train_step(model, opt, x, y)                     # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        TO.train_step(model, opt, x, y)
    dt = time.perf_counter() - t0
    return dict(value=round(batch * steps / dt, 2), unit="images/s", cores=torch.get_num_threads(), kind="port",
                sample="%d timed steps (after 1 warm-up) of the same train step at batch %d, torch-CPU restatement of the "
                       "reference modules (oracle/torch_oracle.py), %.1f s of CPU work" % (steps, batch, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--workload", default="c2", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=64)
    ap.add_argument("--cpu-steps", type=int, default=16, help="timed CPU steps at --cpu-batch (about 10-15 s of CPU work)")
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of replaying captured HIP graphs")
    ap.add_argument("--kernel-steps", type=int, default=5, help="eager steps of the per-kernel HIP-event timing pass")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="threads for the CPU baseline; 0 = os.cpu_count(). 16 is the fastest setting measured on the MI355X host "
                         "(2x EPYC 9575F: 97 img/s at 16 threads, 72 at 32, 41 at 64, 24 at 128, 1.1 at 256)")
    ap.add_argument("--cpu-only", action="store_true", help="only time the CPU baseline (no GPU work)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    if args.cpu_only:
        print(json.dumps(cpu_baseline(args.workload, args.cpu_batch, args.cpu_steps, args.cpu_threads)), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MN_DIST_BACKEND", "nccl")      # "gloo": functional check of the multi-rank path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from micronet_amd import dp
    from micronet_amd.train import GraphedTrainStep, synth_batch
    model, opt = build(args.workload, device)
    dp.broadcast_parameters(model)
    x, y = synth_batch(args.batch, seed=1234 + rank, device=device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The step is captured once in HIP graphs and replayed (micronet_amd.train.GraphedTrainStep): the eager step issues ~150
    # launches through Python and is host-bound.  --no-graph (or a failed capture) runs it eagerly with the bucketed,
    # backward-overlapped gradient all-reduce of micronet_amd.dp.GradSync.
    graphed, graph_err, sync = None, None, None
    if not args.no_graph:
        try:
            graphed = GraphedTrainStep(model, opt, x, y)
        except Exception as e:                      # noqa: BLE001 -- report and measure the eager step instead
            graph_err = "%s: %s" % (type(e).__name__, str(e)[:200])
            graphed = None
            model, opt = build(args.workload, device)
            dp.broadcast_parameters(model)
    if graphed is None:
        sync = dp.GradSync(model)

    def one_step():
        if graphed is not None:
            return graphed.step()[0]
        return dp.train_step_dp(model, opt, sync, x, y)[0]

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss)

    # per-kernel HIP-event timing: the same step, same process, run eagerly right after the timed region (events cannot be
    # read back from inside a replayed graph); single-GPU runs only
    agg = {}
    if world == 1 and not args.no_kernel_timing:
        if sync is None:
            sync = dp.GradSync(model)
        prof = KernelProfiler()
        dp.train_step_dp(model, opt, sync, x, y)
        torch.cuda.synchronize()
        prof.start()
        for _ in range(args.kernel_steps):
            dp.train_step_dp(model, opt, sync, x, y)
        torch.cuda.synchronize()
        agg = prof.stop()
    if graphed is not None:
        graphed.finish()

    if rank == 0:
        imgs = args.batch * world * args.steps
        value = imgs / dt
        out = {
            "metric": "QAT images/sec (nin_gc CIFAR-10, W-ternary/A-binary)" if args.workload == "c2" else "QAT images/sec (%s)" % args.workload,
            "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD_DESC[args.workload], "global_batch": args.batch * world,
                       "per_gpu_batch": args.batch, "parallelism": "dp%d" % world, "optimizer": "Adam lr=0.01",
                       "hip_graph": graphed is not None, "final_loss": round(final_loss, 4)},
        }
        if graph_err:
            out["config"]["hip_graph_error"] = graph_err
        if agg:
            ks = args.kernel_steps
            dom = max(agg, key=lambda k: agg[k]["ms"])
            d = agg[dom]
            achieved = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            # HBM bytes per launch of that kernel from the rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950 note in
            # MI355X_MICROARCH.md, + WRITE_SIZE), collected separately (scripts/pmc_traffic.sh) and committed under profiles/
            traffic = None
            tj = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
            if os.path.exists(tj):
                traffic = json.load(open(tj)).get(dom, {}).get("bytes_per_launch")
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                               "bytes_per_launch": int(d["bytes"] / d["launches"]),
                               "avg_launch_us": round(1000.0 * d["ms"] / d["launches"], 2), "launches": d["launches"],
                               "timing": "HIP events on the launch stream around every launch, %d eager steps after the timed region" % ks}
            out["kernels"] = {k: {"ms_per_step": round(v["ms"] / ks, 4), "launches_per_step": v["launches"] / ks,
                                  "avg_us": round(1000.0 * v["ms"] / v["launches"], 1),
                                  "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
        if args.workload in ("c1", "c2", "c3", "c1_w2a2"):
            per_gpu = value / world
            out["step_level"] = {"algorithmic_GBps": round(per_gpu * NIN_GC_MB_PER_IMG / 1e3, 1),
                                 "hbm_frac": round(per_gpu * NIN_GC_MB_PER_IMG / 1e3 / HBM_PEAK_GBS, 4),
                                 "algorithmic_TFLOPs": round(per_gpu * NIN_GC_GFLOP_PER_IMG / 1e3, 2),
                                 "fp32_mfma_frac": round(per_gpu * NIN_GC_GFLOP_PER_IMG / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_batch, args.cpu_steps, args.cpu_threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
