#!/usr/bin/env python
"""Micro-benchmark of the fused conv + BatchNorm + sign kernels (qgemm_sign.hip) on the nin_gc pointwise layers at batch 256:
forward (statistics + sign), backward of the BN+sign (partials + apply) and the plain forward on sign codes."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import abi_driver  # noqa: E402
from micronet_amd import _lib  # noqa: E402

LAYERS = {"L2": (256, 256, 2, 32, 0), "L3": (256, 256, 2, 32, 2), "L5": (512, 512, 4, 16, 16), "L6": (512, 512, 4, 16, 4), "L8": (1024, 1024, 8, 8, 32)}


def main():
    be = abi_driver.Backend("gpu")
    N = 256
    P = be.ptr
    for name, (Cin, Cout, G, S, sh) in LAYERS.items():
        g = be.geom((N, Cin, S, S), (Cout, Cin // G, 1, 1), groups=G)
        g.in_shuffle = sh
        a = (torch.randint(0, 2, (N, Cin, S, S), device="cuda", dtype=torch.int8) * 2 - 1)
        t = torch.randint(-1, 2, (Cout, Cin // G, 1, 1), device="cuda").float()
        t[:, 0] = 1
        w = t * (torch.rand((Cout, 1, 1, 1), device="cuda") * 0.2 + 0.05)
        bias = torch.randn(Cout, device="cuda") * 0.1
        gamma, beta = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda") * 0.1
        rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
        save = torch.empty(2, Cout, device="cuda")
        out8 = torch.empty((N, Cout, S, S), device="cuda", dtype=torch.int8)
        y = torch.empty((N, Cout, S, S), device="cuda")
        da, dy = torch.randn_like(y), torch.empty_like(y)
        dg, db = torch.empty(Cout, device="cuda"), torch.empty(Cout, device="cuda")
        wq = be.wq(mode=1)
        nb = int(be.lib.mn_qconv_bnsign_ws_bytes(C.byref(g)))
        ws = torch.empty(nb // 4 + 8, device="cuda")
        aq = be.actq(3)
        nbc = int(be.lib.mn_conv2d_ws_bytes(C.byref(g), 0, 3))
        wsc = torch.empty(nbc // 4 + 8, device="cuda")

        def f_fwd():
            be.call("mn_qconv_bnsign_fwd", C.byref(g), C.byref(wq), P(a), P(w), P(bias), P(gamma), P(beta), 1e-5, 0.1, 1, P(rm), P(rv), P(save), P(out8), P(ws), nb, be.stream)

        def f_bwd():
            be.call("mn_qconv_bnsign_bwd", C.byref(g), C.byref(wq), P(a), P(w), P(bias), P(gamma), P(beta), P(save), P(da), 1, P(dy), P(dg), P(db), P(ws), nb, be.stream)

        def f_y():
            be.call("mn_conv2d_fwd", C.byref(g), C.byref(aq), C.byref(wq), P(a), P(w), P(bias), P(y), P(wsc), nbc, 3, be.stream)

        for tag, fn in (("fused fwd", f_fwd), ("fused bwd", f_bwd), ("plain y", f_y)):
            fn(); torch.cuda.synchronize()
            be.lib.mn_profile_enable(1)
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            buf = (_lib.ProfEntry * 64)()
            n = be.lib.mn_profile_collect(buf, 64)
            be.lib.mn_profile_enable(0)
            for i in range(n):
                e = buf[i]
                us = 1e3 * e.total_ms / e.launches
                print("%-3s %-10s %-34s %8.1f us  %7.1f GB/s" % (name, tag, e.name.decode(), us, e.bytes / e.launches / us / 1e3), flush=True)


if __name__ == "__main__":
    main()
