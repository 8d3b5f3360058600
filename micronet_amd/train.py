"""Training-step harness: the caller of the hot path, mirrored from the reference's ``main.py`` scripts.

``build_model`` + ``init_like_main`` reproduce ``wqaq/dorefa/main.py:200,283-297`` (seed 1, xavier conv weights,
zero biases, N(0, 0.01) linear weights); ``make_optimizer`` is ``main.py:308-315`` (Adam, one param group per
tensor); ``train_step`` is the loop body ``main.py:77-82``; ``synth_batch`` is the CIFAR-10-shaped synthetic batch
of SURVEY.md 8(d) (generated on CPU so the CPU oracle and the GPU see identical bits).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from micronet_amd.models import nin, nin_gc, resnet


def init_like_main(model):
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, 0, 0.01)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
    return model


def build_model(arch, seed=1):
    torch.manual_seed(seed)
    ctor = {"nin": nin.Net, "nin_gc": nin_gc.Net, "resnet18": resnet.resnet18, "resnet34": resnet.resnet34,
            "resnet50": resnet.resnet50}[arch]
    return init_like_main(ctor())


def synth_batch(batch, seed=1234, device=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, 3, 32, 32, generator=g)
    y = torch.randint(0, 10, (batch,), generator=g)
    if device is not None:
        x, y = x.to(device), y.to(device)
    return x, y


def make_optimizer(model, lr=0.01, weight_decay=1e-5, fused=None, **kw):
    """Adam with one parameter group per tensor, as main.py:308-315 builds it.  On the GPU the step runs as ONE launch
    over all tensors (micronet_amd.optim.Adam, same update and state layout as torch.optim.Adam); ``fused=False`` keeps
    torch's implementation (used by tests to compare)."""
    groups = [{"params": [p], "lr": lr, "weight_decay": weight_decay} for _, p in model.named_parameters()]
    if fused is None:
        fused = all(p.is_cuda for g in groups for p in g["params"])
    if fused:
        from micronet_amd.optim import Adam
        return Adam(groups, lr=lr, weight_decay=weight_decay, **kw)
    return torch.optim.Adam(groups, lr=lr, weight_decay=weight_decay, **kw)


def train_step(model, optimizer, data, target):
    output = model(data)
    loss = F.cross_entropy(output, target)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss, output
