#!/usr/bin/env python
"""Micro-benchmark of the pointwise binary block's backward through the C ABI at nin_gc's layer shapes (batch 256): the one-kernel form (mn_conv2d_bwd_bnh, k_pwb)
against the two-kernel form (mn_conv2d_bwd_data_bnh[_pool] + mn_conv2d_bwd_weight_bnh[_pool]), HIP events on the launch stream.

    python scripts/kbench_pwb.py [--batch 256] [--iters 20] [--layers L2,L3,L5,L6,L8] [--two]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import abi_driver  # noqa: E402

LAYERS = {"L2": (256, 2, 32, 0, False), "L3": (256, 2, 32, 2, True), "L5": (512, 4, 16, 16, False), "L6": (512, 4, 16, 4, True), "L8": (1024, 8, 8, 32, False)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--layers", default="L2,L3,L5,L6,L8")
    ap.add_argument("--two", action="store_true", help="also time the two-kernel form")
    args = ap.parse_args()
    be = abi_driver.Backend("gpu")
    gen = torch.Generator(device="cuda").manual_seed(5)
    for name in args.layers.split(","):
        Cc, G, S, shuf, pooled = LAYERS[name]
        N = args.batch
        g = be.geom((N, Cc, S, S), (Cc, Cc // G, 1, 1), groups=G)
        g.in_shuffle = shuf
        wq = be.wq(mode=1)
        x = ((torch.rand((N, Cc, S, S), device="cuda", generator=gen) > 0.5).to(torch.int8) * 2 - 1)
        own = ((torch.rand((N, Cc, S, S), device="cuda", generator=gen) > 0.6).to(torch.int8) * 2 - 1)
        h = torch.randint(30, 100, (N, Cc, S, S), device="cuda", generator=gen, dtype=torch.int32).to(torch.uint8)
        da = torch.randn((N, Cc, S // 2, S // 2) if pooled else (N, Cc, S, S), device="cuda", generator=gen)
        t = torch.randint(-1, 2, (Cc, Cc // G, 1, 1), device="cuda", generator=gen).float()
        t[:, 0] = 1
        w = t * (torch.rand((Cc, 1, 1, 1), device="cuda", generator=gen) * 0.2 + 0.05)
        chan = torch.zeros((8, Cc), device="cuda")
        chan[1] = 1.0; chan[2] = -25.0; chan[3] = 25.0; chan[4] = 0.1; chan[5] = 0.01; chan[6] = 1.3; chan[7] = 128.0
        sums = torch.randn((2, Cc), device="cuda", generator=gen)
        dx, dw, db = torch.empty((N, Cc, S, S), device="cuda"), torch.empty_like(w), torch.empty(Cc, device="cuda")
        nb = int(be.lib.mn_conv2d_bwd_bnh_ws_bytes(C.byref(g)))
        ws = torch.empty(nb // 4 + 4, device="cuda")
        nel = N * Cc * S * S

        def fused():
            be.call("mn_conv2d_bwd_bnh", C.byref(g), C.byref(wq), be.ptr(da), be.ptr(h), be.ptr(own) if pooled else None, be.ptr(chan), be.ptr(sums), 1, be.ptr(w),
                    be.ptr(x), be.ptr(dx), be.ptr(dw), be.ptr(db), be.ptr(ws), nb, be.stream)

        nb1, nb2 = int(be.lib.mn_conv2d_ws_bytes(C.byref(g), 1, 0)), int(be.lib.mn_conv2d_ws_bytes(C.byref(g), 2, 0))
        ws1, ws2 = torch.empty(nb1 // 4 + 4, device="cuda"), torch.empty(nb2 // 4 + 4, device="cuda")

        def two():
            if pooled:
                be.call("mn_conv2d_bwd_data_bnh_pool", C.byref(g), C.byref(wq), be.ptr(da), be.ptr(h), be.ptr(own), be.ptr(chan), be.ptr(sums), 1, be.ptr(w), be.ptr(dx),
                        be.ptr(ws1), nb1, be.stream)
                be.call("mn_conv2d_bwd_weight_bnh_pool", C.byref(g), be.ptr(da), be.ptr(h), be.ptr(own), be.ptr(chan), be.ptr(sums), 1, be.ptr(x), be.ptr(dw), be.ptr(db),
                        be.ptr(ws2), nb2, be.stream)
            else:
                be.call("mn_conv2d_bwd_data_bnh", C.byref(g), C.byref(wq), be.ptr(da), be.ptr(h), be.ptr(chan), be.ptr(sums), 1, be.ptr(w), be.ptr(dx), be.ptr(ws1), nb1,
                        be.stream)
                be.call("mn_conv2d_bwd_weight_bnh", C.byref(g), be.ptr(da), be.ptr(h), be.ptr(chan), be.ptr(sums), 1, be.ptr(x), be.ptr(dw), be.ptr(db), be.ptr(ws2), nb2,
                        be.stream)

        def timeit(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / args.iters

        byts = nel * ((1 + 1 + 1 + 1 + 4) if pooled else (4 + 1 + 1 + 4))
        us = timeit(fused)
        line = "%s%s  one-kernel %7.1f us  %6.2f TB/s (%d B/elt incl. the pack + reduce launches)" % (name, " pooled" if pooled else "", us, byts / us * 1e-6, byts // nel)
        if args.two:
            us2 = timeit(two)
            line += "   two-kernel %7.1f us" % us2
        print(line, flush=True)


if __name__ == "__main__":
    main()
