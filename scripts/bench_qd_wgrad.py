"""Kernel-level timing of the dense backward-weight on the resnet18 layer shapes (batch 256): `python scripts/bench_qd_wgrad.py` prints one line per layer shape with the
average launch time of mn_conv2d_bwd_weight (kernel + reduction, HIP events on the launch stream) for the default kernel and, in a child process (the library reads its
knobs once), for MN_QD_WGRAD32=1; the maximum |dw| difference between the two is printed as a sanity figure (both are float accumulations of exact products: ~1e-6
relative).  GPU only; a few seconds per shape."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SHAPES = [((256, 64, 32, 32), 64), ((256, 128, 16, 16), 128), ((256, 256, 8, 8), 256), ((256, 512, 4, 4), 512)]


def run(tag):
    import numpy as np
    import torch
    import abi_driver
    be = abi_driver.Backend("gpu")
    out = {}
    for xs, Oc in SHAPES:
        N, Cin, H, W = xs
        g = be.geom(xs, (Oc, Cin, 3, 3), stride=1, padding=1)
        gen = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randint(0, 16, xs, dtype=torch.uint8, device="cuda", generator=gen)
        gy = torch.randn((N, Oc, H, W), dtype=torch.float32, device="cuda", generator=gen)
        aq = be.actq(4, 4)
        dw = torch.empty((Oc, Cin, 3, 3), dtype=torch.float32, device="cuda")
        nb = int(be.lib.mn_conv2d_ws_bytes(C.byref(g), 2, 0))
        ws = torch.empty(nb // 4 + 8, dtype=torch.float32, device="cuda")
        call = lambda: be.call("mn_conv2d_bwd_weight", C.byref(g), C.byref(aq), be.ptr(gy), be.ptr(x), be.ptr(dw), None, be.ptr(ws), nb, 0, be.stream)
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        out["%dx%d@%d" % (Cin, Oc, H)] = {"us": e0.elapsed_time(e1) * 1e3 / reps, "kernel": be.lib.mn_last_kernel().decode(),
                                         "dw": dw.double().abs().sum().item(), "dw_path": "/tmp/qdw_%s_%d.pt" % (tag, Cin)}
        torch.save(dw.cpu(), out["%dx%d@%d" % (Cin, Oc, H)]["dw_path"])
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
        sys.exit(0)
    res = {}
    for tag, env in (("base", {}), ("wgrad32", {"MN_QD_WGRAD32": "1"})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), tag], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(tag, "failed", r.stderr[-2000:])
            continue
        res[tag] = json.loads(line[-1][7:])
    import torch
    for k in res.get("base", {}):
        b, v = res["base"][k], res.get("wgrad32", {}).get(k)
        if v is None:
            print(k, b)
            continue
        d0, d1 = torch.load(b["dw_path"]), torch.load(v["dw_path"])
        rel = ((d0.double() - d1.double()).abs().max() / d0.double().abs().max()).item()
        print("%-14s %-22s %7.1f us   %-22s %7.1f us   x%.2f   max rel diff %.1e" % (k, b["kernel"], b["us"], v["kernel"], v["us"], b["us"] / v["us"], rel))
