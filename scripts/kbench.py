#!/usr/bin/env python
"""Per-kernel micro-benchmark through the C ABI: every nin_gc conv layer at batch 256 (BASELINE configs[1] shapes),
fwd / bwd-data / bwd-weight, per algorithm, timed with HIP events on the launch stream.  Prints algorithmic GB/s
(4 B x (tensors read + written)) and the fraction of the 8 TB/s HBM peak.

    python scripts/kbench.py [--algos 2,3] [--batch 256] [--iters 20] [--scheme wbwtab|dorefa|iao] [--layers L2,L4]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import abi_driver  # noqa: E402

LAYERS = {   # name: (Cin, Cout, k, pad, groups, HW side)
    "L1": (3, 256, 5, 2, 1, 32), "L2": (256, 256, 1, 0, 2, 32), "L3": (256, 256, 1, 0, 2, 32),
    "L4": (256, 512, 3, 1, 16, 16), "L5": (512, 512, 1, 0, 4, 16), "L6": (512, 512, 1, 0, 4, 16),
    "L7": (512, 1024, 3, 1, 32, 8), "L8": (1024, 1024, 1, 0, 8, 8), "L9": (1024, 10, 1, 0, 1, 8),
    # L2 / L5 with image sides that are not powers of two (probe for power-of-two row-stride effects in HBM channel mapping)
    "X2": (256, 256, 1, 0, 2, 36), "X2b": (256, 256, 1, 0, 2, 28), "X5": (512, 512, 1, 0, 4, 20),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algos", default="2,3")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--scheme", default="wbwtab")
    ap.add_argument("--layers", default="L2,L4,L5,L7,L8,L9")
    ap.add_argument("--json", default="")
    ap.add_argument("--which", default="fwd,dgrad,wgrad")
    args = ap.parse_args()
    be = abi_driver.Backend("gpu")
    algos = [int(a) for a in args.algos.split(",")]
    gen = torch.Generator(device="cuda").manual_seed(3)
    rows = []
    for name in args.layers.split(","):
        Cin, Cout, k, pad, G, S = LAYERS[name]
        N = args.batch
        g = be.geom((N, Cin, S, S), (Cout, Cin // G, k, k), padding=pad, groups=G)
        if args.scheme == "wbwtab":
            x = (torch.rand((N, Cin, S, S), device="cuda", generator=gen) > 0.5).float() * 2 - 1
            t = torch.randint(-1, 2, (Cout, Cin // G, k, k), device="cuda", generator=gen).float()
            t[:, 0] = 1
            w = t * (torch.rand((Cout, 1, 1, 1), device="cuda", generator=gen) * 0.2 + 0.05)
            aq, wq = be.actq(0), be.wq(mode=1)
        elif args.scheme == "dorefa":
            x = torch.randn((N, Cin, S, S), device="cuda", generator=gen) * 4
            n = 255.0
            kk = torch.randint(0, 256, (Cout, Cin // G, k, k), device="cuda", generator=gen).float()
            w = 2 * (kk * (1.0 / n)) - 1
            aq, wq = be.actq(1, 8), be.wq(mode=2, bits=8)
        elif args.scheme == "sign8":     # packed +-1 activations (int8 codes), ternary weights
            x = (torch.randint(0, 2, (N, Cin, S, S), device="cuda", generator=gen, dtype=torch.int8) * 2 - 1)
            t = torch.randint(-1, 2, (Cout, Cin // G, k, k), device="cuda", generator=gen).float()
            t[:, 0] = 1
            w = t * (torch.rand((Cout, 1, 1, 1), device="cuda", generator=gen) * 0.2 + 0.05)
            aq, wq = be.actq(3), be.wq(mode=1)
        elif args.scheme == "real":      # un-quantised layer (the first conv of wbwtab / dorefa nets): fp32 x, fp32 w
            x = torch.randn((N, Cin, S, S), device="cuda", generator=gen)
            w = torch.randn((Cout, Cin // G, k, k), device="cuda", generator=gen) * 0.1
            aq, wq = be.actq(0), None
        else:
            x = torch.randn((N, Cin, S, S), device="cuda", generator=gen) * 4
            code = torch.randint(-127, 128, (Cout, Cin // G, k, k), device="cuda", generator=gen).float()
            scale = torch.rand(Cout, device="cuda", generator=gen) * 0.01 + 0.002
            w = code * scale.view(-1, 1, 1, 1)
            qp = torch.tensor([4 * 4.0 / 127.5, 0.0, -127.5, 127.5], device="cuda")
            aq, wq = be.actq(2, 8, 0, qp), be.wq(mode=3, bits=8, per_channel=1, scale=scale)
        y = be.conv_fwd(g, aq, x, w, None, 0, wq=wq)
        gy = torch.randn(y.shape, device="cuda", generator=gen)
        nx, ny, nw = x.numel() * x.element_size(), y.numel() * 4, w.numel() * 4
        ste = args.scheme != "wbwtab"
        nbytes = {"fwd": nx + ny + nw, "dgrad": ny + nx + nw + (nx if ste else 0), "wgrad": ny + nx + nw}
        dx = torch.empty(x.shape, device="cuda")
        dw = torch.empty_like(w)
        db = torch.empty(Cout, device="cuda")
        P = be.ptr
        for algo in algos:
            wsb = [int(be.lib.mn_conv2d_ws_bytes(C.byref(g), k_, algo)) for k_ in range(3)]
            ws = torch.empty(max(wsb) // 4 + 64, device="cuda")

            def f_fwd():
                be.call("mn_conv2d_fwd", C.byref(g), C.byref(aq), C.byref(wq) if wq is not None else None, P(x), P(w), None, P(y), P(ws), wsb[0], algo, be.stream)

            def f_dgrad():
                be.call("mn_conv2d_bwd_data", C.byref(g), C.byref(aq), C.byref(wq) if wq is not None else None, P(gy), P(w), P(x), P(dx), P(ws), wsb[1], algo, be.stream)

            def f_wgrad():
                be.call("mn_conv2d_bwd_weight", C.byref(g), C.byref(aq), P(gy), P(x), P(dw), P(db), P(ws), wsb[2], algo, be.stream)

            for which, fn in (("fwd", f_fwd), ("dgrad", f_dgrad), ("wgrad", f_wgrad)):
                if which not in args.which.split(","):
                    continue
                try:
                    fn()
                except RuntimeError as e:
                    rows.append(dict(layer=name, which=which, algo=algo, error=str(e)[:60]))
                    continue
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(args.iters):
                    fn()
                b.record()
                torch.cuda.synchronize()
                us = a.elapsed_time(b) * 1e3 / args.iters
                gbs = nbytes[which] / us / 1e3
                rows.append(dict(layer=name, which=which, algo=algo, us=round(us, 1), GBps=round(gbs, 1), frac=round(gbs / 8000, 4)))
                print("%-3s %-6s algo %d  %9.1f us  %8.1f GB/s  %5.1f%% of 8 TB/s" % (name, which, algo, us, gbs, gbs / 80), flush=True)
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
