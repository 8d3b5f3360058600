#!/usr/bin/env python
"""QAT training-step throughput of the fake-quantized conv hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = forward + CrossEntropy + zero_grad + backward + Adam.step on one synthetic CIFAR-10-shaped batch
(mirror of the reference's wqaq/dorefa/main.py:77-82), data already resident in HBM.  BASELINE.json's metric names nin_gc
under BOTH low-bit schemes, so one run measures both, batch 256 PER GPU (weak scaling), data-parallel with a RCCL all-reduce
of the gradients:
  * primary (`value`, `roofline`, `cpu_baseline`): configs[1] -- wbwtab W-ternary / A-binary;
  * `also.{c1_w2a2, c1, c3, c4, c5}`: the same measurement (value, ms_per_step, roofline of its own dominant kernel) for DoReFa W2A2 on nin_gc (the
    second half of the headline metric) and for every other BASELINE config at 256 images per GPU; `values` repeats all images/s figures at top level.
`--only W` measures a single workload (profiling runs).  Prints ONE JSON line on rank 0.
`--gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks itself (torch.distributed.run, 127.0.0.1) and relays rank 0's line.
The timed region is repeated `--repeats` times (each EXACTLY `--steps` steps between barriers); `value` / `ms_per_step` are the MEDIAN window,
`ms_per_step_min` the best one.

roofline.achieved = the dominant kernel's designed HBM bytes / its HIP-event duration (events recorded by the library on the
launch stream around every launch).  roofline.traffic / roofline.mfma_busy come from rocprofv3 PMC passes that THIS run
spawns on itself at N = 1 (separate passes: FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE; no tracing
domain combined with --pmc; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md) -- `--no-pmc` skips them (traffic null).
"""
import argparse
import re
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_16x16x4_f32 rate
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense
INT8_MFMA_PEAK_TOPS = 5000.0    # dense v_mfma_i32_*_i8 (the k_qd_fwd8 code-domain forward)
N_SIMD = 1024                   # 256 CUs x 4
N_XCD = 8
GFLOP_PER_IMG = {"nin_gc": 0.9068, "resnet18": 3.3290}   # fwd+bwd, all convs
MB_PER_IMG = {"nin_gc": 22.71, "resnet18": 12.91}        # fused-ideal fp32 activation traffic fwd+bwd

WORKLOADS = {
    # name: (arch, scheme module, prepare kwargs, weight decay)   (BASELINE.json configs)
    "c2": ("nin_gc", "wbwtab", dict(A=2, W=3), 0.0),
    "c2b": ("nin_gc", "wbwtab", dict(A=2, W=2), 0.0),             # the literal `wbwtab/main.py --W 2 --A 2` ("W2A2" of BASELINE.json's metric string)
    "c1": ("nin_gc", "wqaq.dorefa", dict(a_bits=8, w_bits=8), 1e-5),
    "c1_w2a2": ("nin_gc", "wqaq.dorefa", dict(a_bits=2, w_bits=2), 1e-5),
    "c3": ("nin_gc", "wqaq.iao", dict(a_bits=8, w_bits=8, q_type=0, q_level=0, weight_observer=0, bn_fuse=True), 1e-5),
    "c4": ("resnet18", "wqaq.dorefa", dict(a_bits=2, w_bits=2), 1e-5),
    "c5": ("resnet18", "wqaq.iao", dict(a_bits=4, w_bits=4, q_type=0, q_level=0), 1e-5),
}
WORKLOAD_DESC = {
    "c2": "nin_gc CIFAR-10 wbwtab W-ternary/A-binary QAT, batch=256 per GPU (BASELINE configs[1])",
    "c2b": "nin_gc CIFAR-10 wbwtab W-binary/A-binary QAT (wbwtab/main.py --W 2 --A 2), batch=256 per GPU",
    "c1": "nin_gc CIFAR-10 DoReFa W8A8 QAT (BASELINE configs[0] scheme)",
    "c1_w2a2": "nin_gc CIFAR-10 DoReFa W2A2 QAT, batch=256 per GPU",
    "c3": "nin_gc CIFAR-10 IAO W8A8 per-channel + BN-fuse QAT (BASELINE configs[2])",
    "c4": "resnet18 CIFAR-10 DoReFa W2A2 QAT (BASELINE configs[3])",
    "c5": "resnet18 CIFAR-10 IAO W4A4 per-channel + quant_add QAT (BASELINE configs[4])",
}
METRIC = {"c2": "QAT images/sec (nin_gc CIFAR-10, W-ternary/A-binary)", "c1_w2a2": "QAT images/sec (nin_gc CIFAR-10, DoReFa W2A2)"}


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
        buf = (self._lib.ProfEntry * 192)()
        n = self.lib.mn_profile_collect(buf, 192)
        self.lib.mn_profile_enable(0)
        return {buf[i].name.decode(): dict(ms=buf[i].total_ms, bytes=buf[i].bytes, flops=buf[i].flops, launches=buf[i].launches) for i in range(n)}


def build(workload, device):
    import importlib
    from micronet_amd.train import build_model, make_optimizer
    arch, scheme, kw, wd = WORKLOADS[workload]
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    envkw = os.environ.get("MN_BENCH_PREPARE_KW") or (os.environ.get("MN_BENCH_WBWTAB_KW") if scheme == "wbwtab" else None)
    if envkw:      # A/B switches, e.g. "fuse_conv_bn=0,packed_activations=1"
        kw = dict(kw, **{k: bool(int(v)) for k, v in (it.split("=") for it in envkw.split(","))})
    model = quantize.prepare(build_model(arch), inplace=True, **kw).to(device)
    model.train()
    return model, make_optimizer(model, 0.01, wd)


REFERENCE_DIR = os.environ.get("MICRONET_REFERENCE", "/root/reference")


def cpu_baseline(workload, batch, steps, threads=0, kind="auto"):
    """The reference's algorithm on the host cores.  kind "reference": the reference's OWN modules (imported from MICRONET_REFERENCE, default /root/reference, in a
    child process -- its package is also called `micronet`; present in the build container only, never on the GPU box); "port": the torch-CPU oracle
    (oracle/torch_oracle.py, bit-identical to the reference on CPU: tests/test_oracle_golden.py); "auto": the reference where it is importable, else the port."""
    if kind in ("auto", "reference") and os.path.isdir(os.path.join(REFERENCE_DIR, "micronet")):
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-ref-child", "--only", workload, "--cpu-batch", str(batch), "--cpu-steps", str(steps),
               "--cpu-threads", str(threads)]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1800)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and line:
            return json.loads(line[-1])
        if kind == "reference":
            raise RuntimeError("reference CPU leg failed: %s" % r.stderr[-400:])
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
    return dict(value=round(batch * steps / dt, 2), unit="images/s", cores=torch.get_num_threads(), host_cores=os.cpu_count(), kind="port", workload=workload, batch=batch,
                sample="%d timed steps (+1 warm-up) of the same train step at batch %d, oracle/torch_oracle.py (torch-CPU restatement of the reference modules), "
                       "%.1f s of CPU work" % (steps, batch, dt))


def cpu_reference_child(workload, batch, steps, threads):
    """Child process of cpu_baseline(kind="reference"): the reference's own `micronet` package first on sys.path (our shim package of the same name must not win),
    its own model files, its own `prepare`, the training step of its main.py (wqaq/dorefa/main.py:77-82) with Adam (main.py:142-146), on the synthetic batch."""
    import importlib
    sys.path[:] = [REFERENCE_DIR] + [q for q in sys.path if os.path.abspath(q or ".") != ROOT] + [ROOT]
    assert not any(m == "micronet" or m.startswith("micronet.") for m in sys.modules), "the shim package was imported before the reference"
    arch, scheme, kw, wd = WORKLOADS[workload]
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    assert os.path.abspath(quantize.__file__).startswith(os.path.abspath(REFERENCE_DIR)), quantize.__file__
    from micronet_amd.train import init_like_main, synth_batch       # (our package: the reference has no synthetic-data helper; init = its main.py:289-297)
    torch.set_num_threads(threads if threads > 0 else os.cpu_count())
    torch.manual_seed(1)
    if arch == "nin_gc":
        model = importlib.import_module("micronet.models.nin_gc").Net()
    else:
        model = importlib.import_module("micronet.models.resnet").resnet18()
    init_like_main(model)
    model = quantize.prepare(model, inplace=True, **kw).train()
    opt = torch.optim.Adam([{"params": [q_], "lr": 0.01, "weight_decay": wd} for q_ in model.parameters()], lr=0.01, weight_decay=wd)
    x, y = synth_batch(batch)

    def step():
        out = model(x)
        loss = torch.nn.functional.cross_entropy(out, y)
        opt.zero_grad()
        loss.backward()
        opt.step()
    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(value=round(batch * steps / dt, 2), unit="images/s", cores=torch.get_num_threads(), host_cores=os.cpu_count(), kind="reference", workload=workload,
                          batch=batch, sample="%d timed steps (+1 warm-up) at batch %d of the reference's own modules (%s), %.1f s of CPU work"
                                                % (steps, batch, os.path.relpath(quantize.__file__, REFERENCE_DIR), dt))), flush=True)


def cpu_more_legs(args, primary):
    """The CPU legs besides the primary one: {name: record}.  c1_b128 = BASELINE.json configs[0] (the reference's CPU configuration: DoReFa W8A8, batch 128)."""
    out = {}
    for name in [w for w in args.cpu_more.split(",") if w]:
        w, b, st = ("c1", 128, 4) if name == "c1_b128" else (name, args.cpu_batch, 2)
        if w == primary and b == args.cpu_batch or w not in WORKLOADS:
            continue
        try:
            out[name] = cpu_baseline(w, b, st, args.cpu_threads, args.cpu_kind)
        except Exception as e:          # noqa: BLE001 -- a failing extra leg must not cost the line
            out[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    return out


def measure(workload, args, world, rank, device):
    """Warm up, time exactly args.steps steps between barriers, then (N = 1) the per-kernel HIP-event pass."""
    from micronet_amd import dp
    from micronet_amd.train import GraphedTrainStep, synth_batch
    model, opt = build(workload, device)
    dp.broadcast_parameters(model)

    def dp_buffers(mdl):
        # IAO activation / QuantAdd ranges: reduced over the global batch before use (default, SURVEY 8e ii: one blocking collective per quantizer; no-op for DoReFa /
        # wbwtab) or -- MN_DP_OBSERVERS=replica -- the reference's nn.DataParallel semantics: per-rank ranges, rank 0's buffers broadcast once per step (dp.ReplicaBuffers)
        if dp.active():
            if os.environ.get("MN_DP_OBSERVERS", "global") == "replica":
                dp.replica_buffers(mdl)
            else:
                dp.sync_observers(mdl)
    dp_buffers(model)
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
            model, opt = build(workload, device)
            dp.broadcast_parameters(model)
            dp_buffers(model)
    if graphed is None:
        sync = dp.GradSync(model)

    def one_step():
        if graphed is not None:
            return graphed.step()[0]
        return dp.train_step_dp(model, opt, sync, x, y)[0]

    for _ in range(args.warmup):
        one_step()
    windows, rank_dts = [], []
    for _ in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = one_step()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            every = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            rank_dts.append([float(e.item()) for e in every])
            dt = max(rank_dts[-1])
        windows.append(dt)
    from micronet_amd import ops as _ops
    fallbacks = _ops.fallback_counts(reset=True)           # stock-operator fall-throughs seen while this workload ran (warm-up, capture, eager steps): expected {}
    dt = sorted(windows)[len(windows) // 2]                 # the median window (an odd count by default)
    final_loss = float(loss.detach())
    # data-parallel runs: what the step's collectives cost by themselves -- the gradient bucket(s) all-reduced 10 times back to back, HIP events on the stream
    dp_info = None
    if dp.active() and graphed is not None and getattr(graphed, "flat", None) is not None:
        bufs = [b for b in (graphed.flat, getattr(graphed, "flat2", None)) if b is not None]
        scratch = [torch.zeros_like(b) for b in bufs]
        us = []
        for b in scratch:
            for _ in range(3):
                dist.all_reduce(b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dist.all_reduce(b)
            e1.record()
            torch.cuda.synchronize()
            us.append(round(100.0 * e0.elapsed_time(e1), 1))
        med = rank_dts[len(rank_dts) // 2] if rank_dts else None
        dp_info = {"grad_buckets_bytes": [int(b.numel() * b.element_size()) for b in bufs], "allreduce_us": us,
                   "two_buckets": getattr(graphed, "graph_a2", None) is not None, "one_bucket_reason": getattr(graphed, "one_bucket_reason", None),
                   "rank_ms_per_step": ([round(1000.0 * d / args.steps, 3) for d in (min(med), max(med))] if med else None)}
        del scratch
    # IAO models: the EAGER data-parallel step (dp.GradSync; at world > 1 one 2-float range collective per activation quantizer, ~360 launches issued from
    # Python) next to the graph-replayed one (at world > 1 captured in segments cut at those collectives: micronet_amd/train.py), at every N, so that the cost of
    # either path is a number (`eager_dp_value`, `graph_segments`)
    eager_dp = None
    is_iao = WORKLOADS[workload][1].endswith("iao")
    segments = len(getattr(graphed, "segments", ())) if graphed is not None else 0
    if graphed is not None and is_iao and not args.no_kernel_timing:
        sync2 = dp.GradSync(model)
        for _ in range(2):
            dp.train_step_dp(model, opt, sync2, x, y)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            dp.train_step_dp(model, opt, sync2, x, y)
        barrier()
        eager_dp = time.perf_counter() - t0
        sync2.remove()

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
    del graphed, model, opt, sync
    torch.cuda.empty_cache()
    return dict(dt=dt, dt_min=min(windows), windows=windows, final_loss=final_loss, hip_graph=graph_err is None and not args.no_graph, graph_err=graph_err, agg=agg,
                eager_dp=eager_dp, fallbacks=fallbacks, segments=segments, dp_info=dp_info)


def dp_single_rank(workloads, args, device):
    """The DATA-PARALLEL step on one rank (MN_DP_SINGLE=1, an RCCL process group of size 1): what one GPU of an N-GPU job executes -- gradients packed into the
    flat bucket, its all-reduce issued between graph A and graph B, for the IAO models every observer range collective issued between two graph segments -- with
    nothing on the links.  The difference to the single-GPU step is the data-parallel step's own cost; link time comes on top of it at N > 1."""
    import copy
    import datetime
    out = {}
    os.environ["MN_DP_SINGLE"] = "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(args.master_port + 7))
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device, timeout=datetime.timedelta(seconds=120))
        a = copy.copy(args)
        a.no_kernel_timing, a.repeats = True, 1
        for w in workloads:
            try:
                m = measure(w, a, 1, 0, device)
                out[w] = {"value": round(args.batch * args.steps / m["dt"], 1), "ms_per_step": round(1000.0 * m["dt"] / args.steps, 3), "hip_graph": m["hip_graph"],
                          **({"graph_segments": m["segments"]} if m.get("segments") else {}), **({"allreduce_us": m["dp_info"]["allreduce_us"], "two_buckets": m["dp_info"]["two_buckets"]} if m.get("dp_info") else {})}
            except Exception as e:          # noqa: BLE001
                out[w] = {"error": "%s: %s" % (type(e).__name__, str(e)[:160])}
        # the IAO workloads once more with the reference's DataParallel buffer semantics (MN_DP_OBSERVERS=replica): no collective inside forward, one flat broadcast per step
        os.environ["MN_DP_OBSERVERS"] = "replica"
        for w in [w_ for w_ in workloads if WORKLOADS[w_][1].endswith("iao")]:
            try:
                m = measure(w, a, 1, 0, device)
                out[w + "_replica"] = {"value": round(args.batch * args.steps / m["dt"], 1), "ms_per_step": round(1000.0 * m["dt"] / args.steps, 3), "graph_segments": m.get("segments", 0)}
            except Exception as e:          # noqa: BLE001
                out[w + "_replica"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:160])}
    except Exception as e:          # noqa: BLE001 -- a reported extra: the line must survive it
        out["error"] = "%s: %s" % (type(e).__name__, str(e)[:160])
    finally:
        os.environ.pop("MN_DP_SINGLE", None)
        os.environ.pop("MN_DP_OBSERVERS", None)
        if dist.is_initialized():
            dist.destroy_process_group()
    return out


def _grad_terms():
    from micronet_amd import _lib
    return int(_lib.get_lib().mn_dense_grad_terms())


def dp_single_child(workloads, args):
    """dp_single_rank in a process of its own: it brings up a process group (an RCCL communicator, its watchdog thread) next to HIP-graph captures -- whatever goes wrong
    there must not cost the line of the run that asked for it."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--dp-single-child", ",".join(workloads), "--batch", str(args.batch), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--master-port", str(args.master_port), "--no-pmc", "--no-cpu-baseline"]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300)
        return json.loads(r.stdout.decode().strip().splitlines()[-1])
    except Exception as e:          # noqa: BLE001 -- a reported extra
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:160])}


def exact_terms_leg(workloads, args):
    """The ResNet workloads again with the EXACT three-term bf16 split of the fp32 gradient in the dense backward kernels (MN_GRAD_TERMS=3; the library reads the knob
    once per process: a child per workload): {workload: {"value", "ms_per_step", "grad_terms": 3}} -- reported beside the default two-term figures."""
    out = {}
    for w in workloads:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--only", w, "--batch", str(args.batch), "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--repeats", str(max(1, min(3, args.repeats))), "--no-pmc", "--no-cpu-baseline", "--no-kernel-timing", "--no-dp-single", "--detail", "/tmp/mn_terms3_%s.json" % w]
        try:
            r = subprocess.run(cmd, env=dict(os.environ, MN_GRAD_TERMS="3"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=240)
            line = json.loads(r.stdout.decode().strip().splitlines()[-1])
            assert line["config"]["grad_terms"]["dense_kernels"] == 3
            out[w] = {"value": line["value"], "ms_per_step": line["ms_per_step"], "grad_terms": 3}
        except Exception as e:          # noqa: BLE001 -- a reported extra
            out[w] = {"error": "%s: %s" % (type(e).__name__, str(e)[:120])}
    return out


def section(workload, m, args, world, pmc):
    """value / ms_per_step / roofline / kernels / step_level of one measured workload."""
    arch = WORKLOADS[workload][0]
    value = args.batch * world * args.steps / m["dt"]
    out = {"workload": WORKLOAD_DESC[workload], "value": round(value, 1), "unit": "images/s", "ms_per_step": round(1000.0 * m["dt"] / args.steps, 3),
           "ms_per_step_min": round(1000.0 * m["dt_min"] / args.steps, 3), "value_best_window": round(args.batch * world * args.steps / m["dt_min"], 1),
           "repeats": len(m["windows"]), "window_ms": [round(1000.0 * w, 2) for w in m["windows"]],
           "hip_graph": m["hip_graph"], "final_loss": round(m["final_loss"], 4)}
    if m["graph_err"]:
        out["hip_graph_error"] = m["graph_err"]
    out["stock_fallbacks"] = m.get("fallbacks") or {}
    out["grad_terms"] = _grad_terms() if arch.startswith("resnet") else 3
    if m.get("eager_dp"):
        out["eager_dp_value"] = round(args.batch * world * args.steps / m["eager_dp"], 1)
    if m.get("segments"):
        out["graph_segments"] = m["segments"]
    if m.get("dp_info"):
        out["dp"] = m["dp_info"]
    agg = m["agg"]
    if agg:
        ks = args.kernel_steps
        dom = max(agg, key=lambda k: agg[k]["ms"])
        d = agg[dom]
        achieved = d["bytes"] / (d["ms"] * 1e-3) / 1e9
        p = (pmc or {}).get(workload, {}).get(dom, {})
        # a dense-conv kernel (qgemm_dense.hip) whose matrix-core time bound exceeds its HBM time bound is priced against the bf16 MFMA peak:
        # achieved = ALGORITHMIC flops (2 x MACs; the backward kernels issue 3 bf16 term passes per algorithmic flop) / HIP-event duration
        # bf16 term passes the matrix cores execute per algorithmic flop: the backward kernels split the fp32 gradient -- the dense (ResNet) kernels into
        # mn_dense_grad_terms() terms (2 by default, 3 under MN_GRAD_TERMS=3), every other kernel into the exact three
        terms = (float(_grad_terms()) if dom.startswith("k_qd_") else 3.0) if ("dgrad" in dom or "wgrad" in dom or dom.startswith("k_pwb")) else 1.0
        peak_tf = INT8_MFMA_PEAK_TOPS if "fwd8" in dom else BF16_MFMA_PEAK_TFLOPS     # the int8 code-domain forward is priced against the int8 peak
        mfma_bound = d.get("flops", 0.0) > 0 and terms * d["flops"] / (peak_tf * 1e12) > d["bytes"] / (HBM_PEAK_GBS * 1e9)
        if mfma_bound:
            ach_tf = d["flops"] / (d["ms"] * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma" if mfma_bound else "hbm", "kernel": dom, "achieved": round(ach_tf, 1) if mfma_bound else round(achieved, 1),
                           "peak": peak_tf if mfma_bound else HBM_PEAK_GBS,
                           "unit": "TFLOP/s" if mfma_bound else "GB/s",
                           "frac": round(ach_tf / peak_tf, 4) if mfma_bound else round(achieved / HBM_PEAK_GBS, 4),
                           **({"hbm_GBps": round(achieved, 1), "flops_per_launch": int(d["flops"] / d["launches"]), "bf16_term_passes": terms,
                               "mfma_issued_frac": round(terms * ach_tf / peak_tf, 4)} if mfma_bound else {}),
                           "traffic": p.get("bytes_per_launch"),
                           "traffic_source": ("rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes spawned by this run" if p.get("bytes_per_launch")
                                              else None),
                           "mfma_busy": p.get("mfma_busy"),
                           "bytes_per_launch": int(d["bytes"] / d["launches"]),
                           "avg_launch_us": round(1000.0 * d["ms"] / d["launches"], 2), "launches": d["launches"],
                           "timing": "HIP events on the launch stream around every launch, %d eager steps after the timed region" % ks}
        # every instantiation of the dominant kernel's template together (k_pwb<BNH, XENC, WIDE, UP> is one source kernel: the pooled / un-pooled / upstream-sums
        # variants of a net's pointwise layers appear as separate rows): designed bytes of all their launches over the sum of their HIP-event durations
        stem = dom.split("<")[0]
        fam = {k: v for k, v in agg.items() if k.split("<")[0] == stem}
        if len(fam) > 1 and not mfma_bound:
            fb, fm, fl = sum(v["bytes"] for v in fam.values()), sum(v["ms"] for v in fam.values()), sum(v["launches"] for v in fam.values())
            out["roofline"]["template_family"] = {"kernel": stem + "<*>", "instantiations": sorted(fam), "launches": fl, "avg_launch_us": round(1000.0 * fm / fl, 2),
                                                  "achieved": round(fb / (fm * 1e-3) / 1e9, 1), "frac": round(fb / (fm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        out["kernels"] = {k: {"ms_per_step": round(v["ms"] / ks, 4), "launches_per_step": v["launches"] / ks,
                              "avg_us": round(1000.0 * v["ms"] / v["launches"], 1),
                              "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                              **({"TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)} if v.get("flops", 0.0) > 0 else {}),
                              **({"pmc_bytes_per_launch": (pmc or {}).get(workload, {}).get(k, {}).get("bytes_per_launch"),
                                  "mfma_busy": (pmc or {}).get(workload, {}).get(k, {}).get("mfma_busy")} if (pmc or {}).get(workload, {}).get(k) else {})}
                          for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
    per_gpu = value / world
    out["step_level"] = {"algorithmic_GBps": round(per_gpu * MB_PER_IMG[arch] / 1e3, 1),
                         "hbm_frac": round(per_gpu * MB_PER_IMG[arch] / 1e3 / HBM_PEAK_GBS, 4),
                         "algorithmic_TFLOPs": round(per_gpu * GFLOP_PER_IMG[arch] / 1e3, 2),
                         "bf16_mfma_frac": round(per_gpu * GFLOP_PER_IMG[arch] / 1e3 / BF16_MFMA_PEAK_TFLOPS, 5),
                         "fp32_mfma_frac": round(per_gpu * GFLOP_PER_IMG[arch] / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4)}
    if (pmc or {}).get(workload, {}).get("_step"):
        out["step_level"].update(pmc[workload]["_step"])
    return out


# ------------------------------------------------------------------------------------------------ PMC passes (rocprofv3, spawned on this script)
PMC_PASSES = [("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")]


def _kname(full):
    k = full.replace("void ", "").split("(")[0]
    # k_pwb's fourth template argument (UP: upstream sums ride along) defaults to 0 and is left out of the name the library reports (mn_last_kernel) when it is 0
    k = re.sub(r"^(k_pwb<\d+, \d+, \d+), 0>$", r"\1>", k)
    return k


def pmc_collect(workloads, batch, timeout_s=240):
    """rocprofv3 --pmc passes of `bench.py --pmc-child` (2 eager steps per workload): per workload and kernel the HBM bytes per
    launch (FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes) and the MFMA-busy fraction (sum SQ_VALU_MFMA_BUSY_CYCLES / (max GRBM_GUI_ACTIVE x 1024
    SIMDs), the MfmaUtil definition of rocprofiler-sdk's counter_defs.yaml).  Returns ({workload: {kernel: {...}, "_step": {...}}}, error)."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="mn_pmc_", dir="/tmp")
    acc = {w: collections.defaultdict(lambda: collections.defaultdict(float)) for w in workloads}
    cnt = {w: collections.defaultdict(lambda: collections.defaultdict(int)) for w in workloads}
    err = None
    env = dict(os.environ, TMPDIR="/tmp", PYTHONDONTWRITEBYTECODE="1")
    t_start = time.perf_counter()
    for counters in PMC_PASSES:
        for w in workloads:
            if time.perf_counter() - t_start > PMC_BUDGET_S:
                err = "PMC time budget (%d s) exhausted: remaining passes skipped" % PMC_BUDGET_S
                continue
            d = os.path.join(tmp, counters[0] + "_" + w)
            cmd = [rocprof, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--pmc-child", "--only", w, "--batch", str(batch)]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
                if r.returncode != 0:
                    err = "rocprofv3 %s rc=%d: %s" % (counters[0], r.returncode, r.stdout.decode(errors="replace")[-300:])
                    continue
            except subprocess.TimeoutExpired:
                err = "rocprofv3 %s timed out" % counters[0]
                continue
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                err = "no counter_collection.csv for %s" % counters[0]
                continue
            for row in csv.DictReader(open(files[0])):
                k = _kname(row["Kernel_Name"])
                c = row["Counter_Name"]
                acc[w][k][c] += float(row["Counter_Value"])
                cnt[w][k][c] += 1
    shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for w in workloads:
        res[w] = {}
        tot_bytes = tot_mfma = tot_gui = 0.0
        stock_launches = 0
        for k, v in acc[w].items():
            e = {}
            n_f, n_w = cnt[w][k].get("FETCH_SIZE", 0), cnt[w][k].get("WRITE_SIZE", 0)
            if n_f and n_w:
                fetch = 2.0 * v["FETCH_SIZE"] / n_f * 1024.0
                write = v["WRITE_SIZE"] / n_w * 1024.0
                e.update(bytes_per_launch=int(fetch + write), fetch_bytes_x2=int(fetch), write_bytes=int(write))
                tot_bytes += 2.0 * v["FETCH_SIZE"] * 1024.0 + v["WRITE_SIZE"] * 1024.0
            n_m = cnt[w][k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
            if n_m and v.get("GRBM_GUI_ACTIVE", 0) > 0:
                e["mfma_busy"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / N_XCD * N_SIMD), 4)      # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs
                tot_mfma += v["SQ_VALU_MFMA_BUSY_CYCLES"]
                tot_gui += v["GRBM_GUI_ACTIVE"]
            if e and (k.startswith("k_") or "k_" in k[:8]):
                res[w][k] = e
            elif not (k.startswith("k_") or "k_" in k[:8]):
                c0 = max(cnt[w][k].values()) if cnt[w][k] else 0          # dispatches of a kernel that is not this library's (MIOpen / ATen): the stock operators left in the step
                stock_launches += c0
        step = {}
        if tot_bytes:
            step["pmc_hbm_bytes_per_step_all_kernels"] = int(tot_bytes / PMC_CHILD_STEPS)
        if tot_gui:
            step["pmc_mfma_busy_all_kernels"] = round(tot_mfma / (tot_gui / N_XCD * N_SIMD), 4)
        if acc[w]:
            step["stock_kernel_launches_per_step"] = round(stock_launches / PMC_CHILD_STEPS, 1)
        res[w]["_step"] = step
    return res, err


PMC_CHILD_STEPS = 2
PMC_BUDGET_S = 240


def pmc_child(args, device):
    """What the PMC passes profile: PMC_CHILD_STEPS eager steps of one workload (every dispatch is serialised by the counter collection)."""
    from micronet_amd import dp
    from micronet_amd.train import synth_batch
    model, opt = build(args.only, device)
    x, y = synth_batch(args.batch, device=device)
    sync = dp.GradSync(model)
    lib_steps = PMC_CHILD_STEPS
    for _ in range(lib_steps):
        dp.train_step_dp(model, opt, sync, x, y)
    torch.cuda.synchronize()

MAX_LINE_BYTES = 4000


def compact_roofline(r):
    """The contract's roofline object for the final line: bound / achieved / peak / unit / frac / traffic plus the few fields that say how it was measured."""
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "mfma_busy", "bytes_per_launch", "flops_per_launch", "avg_launch_us", "launches",
            "mfma_issued_frac", "hbm_GBps")
    out = {k: r[k] for k in keep if k in r}
    out["timing"] = "HIP events on the launch stream"
    if "template_family" in r:          # all instantiations of the dominant kernel's template together (short form)
        out["template_family"] = {k: r["template_family"][k] for k in ("kernel", "launches", "avg_launch_us", "achieved", "frac")}
    if r.get("traffic") is not None:
        out["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE(x2 gfx950)+WRITE_SIZE"
    return out

def compose_line(primary, sec, also_secs, also_err, pmc_err, cpu, args, world, dist_info=None, cpu_more=None, dp1=None):
    """(final-line dict, detail dict).  The final stdout line is what the driver parses: the contract's keys, the HEADLINE workload's roofline, cpu_baseline, one
    short record per secondary workload, every images/s figure under `values` -- bounded by MAX_LINE_BYTES.  Everything else (per-kernel tables, step_level,
    timed windows of every workload) is `detail`, written to a side file."""
    out = {
        "metric": METRIC.get(primary, "QAT images/sec (%s)" % primary),
        "value": sec["value"], "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": sec["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": WORKLOAD_DESC[primary], "global_batch": args.batch * world,
                   "per_gpu_batch": args.batch, "parallelism": "dp%d" % world, "optimizer": "Adam lr=0.01",
                   "hip_graph": sec["hip_graph"], "final_loss": sec["final_loss"],
                   # quantized layers that fell through to a stock torch operator while any workload of this run executed (expected 0)
                   "quant_layer_fallbacks": sum(sec.get("stock_fallbacks", {}).values()) + sum(sum(s_.get("stock_fallbacks", {}).values()) for s_ in also_secs.values()),
                   "stock_fallbacks": sum(sec.get("stock_fallbacks", {}).values()) + sum(sum(s_.get("stock_fallbacks", {}).values()) for s_ in also_secs.values()),          # (the key's name until round 5: kept for readers of older lines)
                   # launches per EAGER step of kernels that are NOT this library's (MIOpen BatchNorm of the un-quantised tail, loss, ATen fills / copies / counters),
                   # counted in the PMC passes (null without them); the graph-replayed step's own count is in profiles/r05_<w>_summary.md
                   "stock_kernels_per_eager_step": sec.get("step_level", {}).get("stock_kernel_launches_per_step"),
                   # bf16 terms carrying the fp32 gradient operand through the matrix cores: the nin_gc kernels use the exact three; the dense (ResNet: c4, c5) backward
                   # kernels mn_dense_grad_terms() -- 2 by default (|error| <= 2^-18 |g| per element), 3 under MN_GRAD_TERMS=3; `values_exact_terms` has the 3-term figures
                   "grad_terms": {"nin_gc_kernels": 3, "dense_kernels": _grad_terms()}},
    }
    out["ms_per_step_min"], out["value_best_window"], out["repeats"], out["window_ms"] = sec["ms_per_step_min"], sec["value_best_window"], sec["repeats"], sec["window_ms"]
    if dist_info:
        out["config"]["dist"] = dist_info
    if "hip_graph_error" in sec:
        out["config"]["hip_graph_error"] = sec["hip_graph_error"][:200]
    for k in ("graph_segments", "eager_dp_value", "dp"):    # IAO data parallel: the segmented replay and the eager step it replaces; dp: bucket bytes, the all-reduce's own us, per-rank ms
        if k in sec:
            out["config"][k] = sec[k]
    detail = {"headline": dict(out), "sections": {primary: sec}}
    if "roofline" in sec:
        out["roofline"] = compact_roofline(sec["roofline"])
    if pmc_err:
        out.setdefault("roofline", {})["pmc_error"] = pmc_err[:160]
        detail["pmc_error"] = pmc_err
    if "step_level" in sec:
        out["step_level"] = {k: sec["step_level"][k] for k in ("algorithmic_GBps", "hbm_frac", "pmc_hbm_bytes_per_step_all_kernels") if k in sec["step_level"]}
    also_out = {}
    for w, s in also_secs.items():
        s["metric"] = METRIC.get(w, "QAT images/sec (%s)" % w)
        s["steps"], s["warmup"], s["n_gpus"] = args.steps, args.warmup, world
        detail["sections"][w] = s
        also_out[w] = {"value": s["value"], "ms_per_step": s["ms_per_step"], "hip_graph": s["hip_graph"], **({k: s[k] for k in ("eager_dp_value", "graph_segments", "dp") if k in s}),
                       **({"roofline": {k: s["roofline"][k] for k in ("bound", "kernel", "frac", "avg_launch_us", "traffic") if k in s["roofline"]}} if "roofline" in s else {})}
    for w, e in also_err.items():
        also_out[w] = {"error": e[:200]}
        detail["sections"][w] = {"error": e}
    if also_out:
        out["also"] = also_out
    # every images/s figure of the line at top level: the metric string names nin_gc under BOTH low-bit schemes (c2 = `value`, c1_w2a2)
    out["values"] = {primary: sec["value"], **{w: v["value"] for w, v in also_out.items() if "value" in v}}
    if cpu is not None:
        out["cpu_baseline"] = cpu
        detail["cpu_baseline"] = cpu
    if cpu_more:
        detail["cpu_baselines"] = cpu_more
        if "c1_b128" in cpu_more:                            # BASELINE.json configs[0]: nin_gc DoReFa W8A8, batch 128, CPU reference path (wqaq/dorefa/main.py:135,171,189-190)
            out["cpu_baseline_c1_b128"] = {k: cpu_more["c1_b128"][k] for k in ("value", "unit", "cores", "kind", "batch", "sample") if k in cpu_more["c1_b128"]}
            if "sample" in out["cpu_baseline_c1_b128"]:          # (the long form stays in the detail file: the line is bounded)
                out["cpu_baseline_c1_b128"]["sample"] = out["cpu_baseline_c1_b128"]["sample"].split(",")[0].replace(" of the same train step", "") + ", as cpu_baseline"
        out["cpu_values"] = {k: v["value"] for k, v in cpu_more.items() if "value" in v}
    if dp1:
        detail["dp_single_rank"] = dp1
        # the line carries the short form (images/s, ms per step, graph segments); bucket bytes / all-reduce times per workload are in the detail file
        out["dp_single_rank"] = {w: ({k: v[k] for k in ("value", "ms_per_step", "graph_segments", "error") if k in v} if isinstance(v, dict) else v) for w, v in dp1.items()}
    for drop in ("window_ms", "step_level", "cpu_values", "dp_single_rank", "also"):      # the driver parses the LAST stdout line: never let it outgrow its reader again (round 3: 34 KB -> parsed null)
        if len(json.dumps(out)) + 64 <= MAX_LINE_BYTES:
            break
        out.pop(drop, None)
    return out, detail


def drain_c_stdio():
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:          # noqa: BLE001
        pass


def write_detail(detail, args):
    """Per-kernel tables, step_level and the timed windows of every measured workload: a side file, NOT the final stdout line."""
    path = args.detail or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(detail, f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError as e:
        return "unwritten: %s" % e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--workload", default="c2", choices=list(WORKLOADS), help="primary workload (value / roofline / cpu_baseline)")
    ap.add_argument("--also", default="c1_w2a2,c2b,c1,c3,c4,c5", help="comma-separated secondary workloads reported under `also` ('' = none)")
    ap.add_argument("--repeats", type=int, default=5, help="timed windows of --steps steps each; value = the median window")
    ap.add_argument("--master-port", type=int, default=29531, help="rendezvous port when --gpus N > 1 starts its own ranks")
    ap.add_argument("--only", default=None, choices=list(WORKLOADS), help="measure this single workload (no `also`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=256, help="batch of the CPU baseline: the benched per-GPU batch")
    ap.add_argument("--cpu-steps", type=int, default=6, help="timed CPU steps at --cpu-batch (about 20 s of CPU work at batch 256)")
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of replaying captured HIP graphs")
    ap.add_argument("--kernel-steps", type=int, default=5, help="eager steps of the per-kernel HIP-event timing pass")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="threads for the CPU baseline; 0 = os.cpu_count(). 16 is the fastest setting measured on the MI355X host "
                         "(2x EPYC 9575F: 97 img/s at 16 threads, 72 at 32, 41 at 64, 24 at 128, 1.1 at 256)")
    ap.add_argument("--cpu-only", action="store_true", help="only time the CPU baseline (no GPU work)")
    ap.add_argument("--cpu-kind", default="auto", choices=["auto", "reference", "port"], help="CPU leg: the reference's own modules (when MICRONET_REFERENCE is importable) or the port")
    ap.add_argument("--cpu-more", default="c1_b128,c1_w2a2,c2b,c3,c4,c5",
                    help="further CPU legs (detail file + cpu_values): c1_b128 = configs[0] (DoReFa W8A8 at batch 128, 4 steps); the others 2 steps at --cpu-batch; '' = none")
    ap.add_argument("--cpu-ref-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 PMC passes (roofline.traffic / mfma_busy stay null)")
    ap.add_argument("--detail", default=None, help="where the per-kernel tables / step_level / windows go (default gpurun_out/bench_detail.json)")
    ap.add_argument("--no-dp-single", action="store_true", help="skip the single-rank data-parallel leg (dp_single_rank: the N-GPU step's own cost measured on one rank)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dp-single-child", default="", help=argparse.SUPPRESS)
    args = ap.parse_args()

    primary = args.only or args.workload
    also = [] if args.only else [w for w in args.also.split(",") if w and w != primary]
    if args.gpus > 1 and args.also == ap.get_default("also"):
        # multi-GPU runs measure the scaling of the headline step: by default the two nin_gc schemes the metric names (same graphed data-parallel step), plus the
        # BASELINE config that is DEFINED at this GPU count (configs[3]: resnet DoReFa W2A2 on 4 GPUs; configs[4]: resnet IAO W4A4 + quant_add on 8); the other
        # configs are single-GPU lines (`also` of the --gpus 1 run) unless --also asks for them explicitly
        keep = {"c1_w2a2"} | ({"c4"} if args.gpus == 4 else set()) | ({"c5"} if args.gpus == 8 else set())
        also = [w for w in also if w in keep]
    for w in also:
        if w not in WORKLOADS:
            raise SystemExit("unknown workload in --also: %s" % w)
    if args.cpu_ref_child:
        cpu_reference_child(primary, args.cpu_batch, args.cpu_steps, args.cpu_threads)
        return
    if args.cpu_only:
        out = {primary: cpu_baseline(primary, args.cpu_batch, args.cpu_steps, args.cpu_threads, args.cpu_kind)}
        out.update(cpu_more_legs(args, primary))
        print(json.dumps(out), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.pmc_child:
        # no launcher: start the N ranks ourselves (one process per GPU, RCCL over xGMI) and relay rank 0's JSON line
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: refusing to report a line whose n_gpus differs from --gpus" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # gradients first accumulated on the warm-up / capture side stream and later on the default stream: the hand-over is ordered by wait_stream
    # (micronet_amd/train.py), the per-parameter warning only clutters the driver's tail
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
    if args.pmc_child:
        pmc_child(args, device)
        return
    if args.dp_single_child:          # (a child of the default run: the single-rank data-parallel leg in a process of its own)
        res = dp_single_rank([w for w in args.dp_single_child.split(",") if w in WORKLOADS], args, device)
        drain_c_stdio()          # (RCCL's version banner sits in C stdio: out before the line the parent parses)
        print(json.dumps(res), flush=True)
        return
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MN_DIST_BACKEND", "nccl")      # "gloo": functional check of the multi-rank path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    m_primary = measure(primary, args, world, rank, device)
    m_also, also_err = {}, {}
    for w in also:
        try:
            m_also[w] = measure(w, args, world, rank, device)
        except Exception as e:          # noqa: BLE001 -- the primary line must survive a failing secondary workload
            if world > 1:
                raise                   # ranks must stay in lock-step
            also_err[w] = "%s: %s" % (type(e).__name__, str(e)[:300])

    dp1 = None
    if world == 1 and not args.only and not args.no_kernel_timing and not args.no_dp_single:
        dp1 = dp_single_child([w for w in (primary, "c3", "c4", "c5") if w == primary or w in m_also], args)

    terms3 = None
    if world == 1 and not args.only and not args.no_kernel_timing and _grad_terms() != 3:
        torch.cuda.synchronize()
        terms3 = exact_terms_leg([w for w in ("c4", "c5") if w in m_also], args)

    pmc, pmc_err = None, None
    if world == 1 and rank == 0 and not args.no_pmc and not args.no_kernel_timing:
        torch.cuda.synchronize()
        pmc, pmc_err = pmc_collect([primary] + list(m_also), args.batch)

    if rank == 0:
        sec = section(primary, m_primary, args, world, pmc)
        also_secs = {w: section(w, m, args, world, pmc) for w, m in m_also.items()}
        dist_info = None
        if world > 1:
            dist_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                         "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None}
        cpu = cpu_baseline(primary, args.cpu_batch, args.cpu_steps, args.cpu_threads, args.cpu_kind) if (world == 1 and not args.no_cpu_baseline) else None
        cpu_more = cpu_more_legs(args, primary) if (world == 1 and not args.no_cpu_baseline and not args.only) else None
        out, detail = compose_line(primary, sec, also_secs, also_err, pmc_err, cpu, args, world, dist_info, cpu_more, dp1)
        if terms3:
            detail["values_exact_terms"] = terms3
            if len(json.dumps(out)) + len(json.dumps(terms3)) + 32 <= MAX_LINE_BYTES:
                out["values_exact_terms"] = {w: v.get("value", v.get("error")) for w, v in terms3.items()}
        out["detail_file"] = write_detail(detail, args)
    if world > 1:
        if rank != 0:
            drain_c_stdio()          # (whatever a library buffered on the other ranks goes out BEFORE rank 0's line, not at their exit)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # The driver parses the LAST stdout line.  RCCL prints a version banner through C stdio on communicator creation (fully buffered on a pipe, i.e. flushed at
        # process exit -- BEHIND a line printed from Python): drain every C stream first, after the process group is gone, then print the line.
        drain_c_stdio()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
