"""Shared driver for the IAO non-conv surface against tests/golden/iao_ops.npz (generated from the reference by
tests/golden/make_golden.py: gen_iao_ops).  `factory(kind, op, **kw)` builds the module under test (oracle on CPU, product on the GPU);
`dev` moves tensors.  Returns the worst errors; quantised values / buffers must be bit-exact for the oracle, the product is held to the
tolerances its caller states."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    return np.load(os.path.join(GOLDEN, "iao_ops.npz")), json.load(open(os.path.join(GOLDEN, "iao_ops_meta.json")))


def run_op(m, g, op, mode, dev, exact):
    """two training steps + eval of one (op, mode); returns {name: max abs err / max |ref|}"""
    key = f"ops_{op}_{mode}"
    errs = {}
    m.train()
    for s_ in range(2):
        x = torch.from_numpy(g[f"ops_x{s_}"].copy()).to(dev).requires_grad_(True)
        y = m(x)
        y.backward(torch.from_numpy(g[f"ops_{op}_g{s_}"].copy()).to(dev))
        for name, got, ref in (("y", y, g[f"{key}_s{s_}_y"]), ("dx", x.grad, g[f"{key}_s{s_}_dx"])):
            got = got.detach().cpu().numpy()
            if exact:
                assert np.array_equal(got, ref), (key, s_, name, float(np.max(np.abs(got - ref))))
            errs[f"s{s_}_{name}"] = float(np.max(np.abs(got - ref)) / max(np.max(np.abs(ref)), 1e-30))
    m.eval()
    y = m(torch.from_numpy(g["ops_x0"].copy()).to(dev)).detach().cpu().numpy()
    ref = g[f"{key}_eval_y"]
    if exact:
        assert np.array_equal(y, ref), (key, "eval")
    errs["eval_y"] = float(np.max(np.abs(y - ref)) / max(np.max(np.abs(ref)), 1e-30))
    return errs
