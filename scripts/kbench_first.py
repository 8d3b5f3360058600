"""Kernel-level timing of the first-layer convolution forward (conv_first.hip) on nin_gc's shape at batch 256 (3 -> 256 channels, 5x5, 32x32).

    python scripts/kbench_first.py [LABEL:ENV=V,ENV=V ...]        (e.g. base: ko_store:MN_LIB_PATH=micronet_amd/lib/libmicronet_hip_kostore.so)

Every variant runs in a child process (the library is loaded once per process); torch events around 50 launches.  GPU only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    from micronet_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((256, 3, 32, 32), device="cuda", generator=gen)
    w = torch.randn((256, 3, 5, 5), device="cuda", generator=gen) * 0.1
    b = torch.randn(256, device="cuda", generator=gen) * 0.1
    with torch.no_grad():
        for _ in range(5):
            y = ops.qconv2d(x, w, b, 1, 2, 1, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            y = ops.qconv2d(x, w, b, 1, 2, 1, 1)
        e1.record()
        torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=2)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    print(json.dumps({"us": round(e0.elapsed_time(e1) * 1000 / 50, 1), "kernel": ops.last_kernel() if hasattr(ops, "last_kernel") else None, "rel_err_vs_fp64": err}))


if __name__ == "__main__":
    if os.environ.get("_KB_CHILD"):
        run()
        sys.exit(0)
    variants = sys.argv[1:] or ["base:"]
    for v in variants:
        label, _, envs = v.partition(":")
        env = dict(os.environ, _KB_CHILD="1")
        for kv in filter(None, envs.split(",")):
            k, _, val = kv.partition("=")
            env[k] = val
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(label, line[-1] if line else ("FAILED " + r.stderr[-300:]), flush=True)
