"""Kernel-level timing of the dense (ResNet) backward kernels on the resnet18 layer shapes at batch 256: mn_conv2d_bwd_data / mn_conv2d_bwd_weight on DoReFa codes.

    python scripts/kbench_dense.py [LABEL:ENV=V,ENV=V ...]        (default: one run with the default knobs)

Every variant runs in a child process (the library reads its MN_* knobs once); per kernel the library's own HIP-event timing (mn_profile_*: events on the launch
stream around every launch) and the maximum difference to the first variant's result, relative to max |result|.  GPU only."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SHAPES = [((256, 64, 32, 32), 64, 3, 1), ((256, 128, 16, 16), 128, 3, 1), ((256, 256, 8, 8), 256, 3, 1), ((256, 512, 4, 4), 512, 3, 1),
          ((256, 64, 32, 32), 128, 3, 2), ((256, 64, 32, 32), 128, 1, 2)]


def run(tag):
    import torch
    import abi_driver
    from micronet_amd import _lib
    be = abi_driver.Backend("gpu")
    out = {}
    for xs, Oc, k, st in SHAPES:
        N, Cin, H, W = xs
        pad = 1 if k == 3 else 0
        g = be.geom(xs, (Oc, Cin, k, k), stride=st, padding=pad)
        gen = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randint(0, 4, xs, dtype=torch.uint8, device="cuda", generator=gen)
        gy = torch.randn((N, Oc, H // st, W // st), dtype=torch.float32, device="cuda", generator=gen) * 1e-3
        kw = torch.randint(0, 4, (Oc, Cin, k, k), device="cuda", generator=gen)
        w = (2.0 * (kw.float() * (1.0 / 3.0)) - 1.0).contiguous()
        aq, wq = be.actq(4, 2), be.wq(mode=2, bits=2)
        dw = torch.empty((Oc, Cin, k, k), dtype=torch.float32, device="cuda")
        dx = torch.empty(xs, dtype=torch.float32, device="cuda")
        nb2 = int(be.lib.mn_conv2d_ws_bytes(C.byref(g), 2, 0))
        ws2 = torch.empty(nb2 // 4 + 8, dtype=torch.float32, device="cuda")
        nb1 = int(be.lib.mn_conv2d_ws_bytes(C.byref(g), 1, 0))
        ws1 = torch.empty(nb1 // 4 + 8, dtype=torch.float32, device="cuda")
        wg = lambda: be.call("mn_conv2d_bwd_weight", C.byref(g), C.byref(aq), be.ptr(gy), be.ptr(x), be.ptr(dw), None, be.ptr(ws2), nb2, 0, be.stream)
        dg = lambda: be.call("mn_conv2d_bwd_data", C.byref(g), C.byref(aq), C.byref(wq), be.ptr(gy), be.ptr(w), None, be.ptr(dx), be.ptr(ws1), nb1, 0, be.stream)
        for _ in range(3):
            wg(); dg()
        torch.cuda.synchronize()
        be.lib.mn_profile_enable(1)
        reps = 20
        for _ in range(reps):
            wg(); dg()
        torch.cuda.synchronize()
        buf = (_lib.ProfEntry * 64)()
        n = be.lib.mn_profile_collect(buf, 64)
        be.lib.mn_profile_enable(0)
        key = "%dx%d@%d k%d s%d" % (Cin, Oc, H, k, st)
        out[key] = {buf[i].name.decode(): round(1000.0 * buf[i].total_ms / buf[i].launches, 1) for i in range(n)}
        torch.save((dw.cpu(), dx.cpu()), "/tmp/kbd_%s_%s.pt" % (tag, key.replace(" ", "_")))
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        run(sys.argv[2])
        sys.exit(0)
    specs = sys.argv[1:] or ["default:"]
    res = {}
    for spec in specs:
        tag, _, envs = spec.partition(":")
        env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", tag], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(tag, "failed", r.stdout[-500:], r.stderr[-1500:])
            continue
        res[tag] = json.loads(line[-1][7:])
    import torch
    first = next(iter(res), None)
    for tag, r in res.items():
        for key, ks in r.items():
            diff = ""
            if tag != first and first is not None:
                a = torch.load("/tmp/kbd_%s_%s.pt" % (first, key.replace(" ", "_")))
                b = torch.load("/tmp/kbd_%s_%s.pt" % (tag, key.replace(" ", "_")))
                diff = "  rel diff vs %s: dw %.1e dx %.1e" % (first, ((a[0].double() - b[0].double()).abs().max() / a[0].double().abs().max()).item(),
                                                               ((a[1].double() - b[1].double()).abs().max() / a[1].double().abs().max()).item())
            print("%-10s %-18s %s%s" % (tag, key, "  ".join("%s %.1f us" % kv for kv in sorted(ks.items())), diff))
