"""The reference's OWN training script, executed unchanged against this package (missing 4).

`wqaq/dorefa/main.py` is loaded with runpy from MICRONET_REFERENCE (default /root/reference; the test is skipped where that tree is absent, i.e. on the GPU box) with
  * `quantize`  -> this repository's module (the script does a bare `import quantize`, dorefa/main.py:21),
  * `models`    -> micronet/models of THIS repository (the script appends "../../../.." of its working directory, dorefa/main.py:7,19),
  * `torchvision` -> a 30-line stub (not installed here) whose CIFAR10 yields synthetic 3 x 32 x 32 images.
The build container has no GPU and the product has no CPU path, so the run is expected to go exactly this far: argument parsing, data loaders, `nin_gc.Net()`,
the script's own initialisation loop, `quantize.prepare(model, inplace=True, a_bits=8, w_bits=8)` (dorefa/main.py:299), the per-tensor Adam groups (303-310), and
inside ITS `train()` (70-95) the first `model(data)` -- which must fail LOUDLY in our operator (MicronetHipError: no CPU fallback) instead of silently running a
stock kernel.  What happens after that call on a GPU is the subject of tests/test_gpu_models.py (same step, same order of calls)."""
import os
import runpy
import sys
import traceback
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MICRONET_REFERENCE", "/root/reference")
MAIN = os.path.join(REF, "micronet", "compression", "quantization", "wqaq", "dorefa", "main.py")


def _stub_torchvision():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    ds = types.ModuleType("torchvision.datasets")

    class _T:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    class Compose(_T):
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    for n in ("RandomCrop", "RandomHorizontalFlip", "ToTensor", "Normalize"):
        setattr(tr, n, type(n, (_T,), {}))
    tr.Compose = Compose

    class CIFAR10(torch.utils.data.Dataset):
        def __init__(self, root=None, train=True, download=False, transform=None):
            g = torch.Generator().manual_seed(7 if train else 8)
            self.x = torch.randn(16, 3, 32, 32, generator=g)
            self.y = torch.randint(0, 10, (16,), generator=g)
            self.transform = transform

        def __len__(self):
            return len(self.x)

        def __getitem__(self, i):
            x = self.x[i]
            return (self.transform(x) if self.transform else x), int(self.y[i])

    ds.CIFAR10 = CIFAR10
    tv.transforms, tv.datasets = tr, ds
    return {"torchvision": tv, "torchvision.transforms": tr, "torchvision.datasets": ds}


@pytest.mark.skipif(not os.path.exists(MAIN), reason="the reference tree (MICRONET_REFERENCE) is not present on this machine")
def test_reference_dorefa_main_runs_unchanged_against_this_package(tmp_path, monkeypatch, capsys):
    from micronet_amd._lib import MicronetHipError
    import micronet.compression.quantization.wqaq.dorefa.quantize as ours
    shim_dir = os.path.join(ROOT, "micronet", "compression", "quantization", "wqaq", "dorefa")
    saved = {k: sys.modules.get(k) for k in ("quantize", "models", "models.nin", "models.nin_gc", "models.resnet", "torchvision", "torchvision.transforms", "torchvision.datasets")}
    try:
        sys.modules.update(_stub_torchvision())
        sys.modules["quantize"] = ours                              # pre-seed (the script directory would otherwise win)
        for k in ("models", "models.nin", "models.nin_gc", "models.resnet"):
            sys.modules.pop(k, None)
        monkeypatch.chdir(shim_dir)                                  # "../../../.." of the working directory = this repository's micronet/ (its models/ package)
        monkeypatch.setattr(sys, "argv", ["main.py", "--cpu", "--w_bits", "8", "--a_bits", "8", "--train_batch_size", "8", "--eval_batch_size", "8",
                                          "--num_workers", "0", "--model_type", "1", "--data", str(tmp_path)])
        monkeypatch.setattr(sys, "dont_write_bytecode", True)
        with pytest.raises(MicronetHipError) as ei:
            runpy.run_path(MAIN, run_name="__main__")
        frames = traceback.extract_tb(ei.value.__traceback__)
        files = [f.filename for f in frames]
        names = [f.name for f in frames]
        assert any(os.path.abspath(f) == os.path.abspath(MAIN) and n == "train" for f, n in zip(files, names)), "the failure is not inside the reference's train()"
        assert any(os.path.abspath(f).startswith(os.path.join(ROOT, "micronet_amd")) for f in files), "the failure is not raised by this package's operators"
        out = capsys.readouterr().out
        assert "***quant_model***" in out and "QuantConv2d" in out          # the script printed the model its own prepare() call rewrote with OUR classes
        import models.nin_gc as used_models
        assert os.path.abspath(used_models.__file__).startswith(os.path.join(ROOT, "micronet", "models"))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
