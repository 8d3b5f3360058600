"""Diagnose the L4 (3x3 g16) d weight mismatch: product vs torch-CPU fp32 oracle vs fp64 truth, teacher forced."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from micronet_amd import ops
torch.manual_seed(0)
N, C, O, G = 8, 256, 512, 16
x = torch.sign(torch.randn(N, C, 16, 16)); x[x == 0] = 1
w = torch.randn(O, C // G, 3, 3) * 0.1
gy = torch.randn(N, O, 16, 16) * torch.rand(1, O, 1, 1) * 1e-3
# fp64 truth of dwq (pure contraction) and dw through the ternary backward
def ternary(w):
    E = w.abs().mean(dim=(1, 2, 3), keepdim=True); thr = 0.7 * E
    t = torch.sign(torch.sign(w + thr) + torch.sign(w - thr))
    mask = (w.abs() > thr).to(w.dtype)
    alpha = (w.abs() * mask).sum(dim=(1, 2, 3), keepdim=True) / mask.sum(dim=(1, 2, 3), keepdim=True)
    return t * alpha
def run(dtype):
    wd = w.to(dtype).clone().requires_grad_(True)
    y = torch.nn.functional.conv2d(x.to(dtype), ternary(wd), None, 1, 1, 1, G)
    y.backward(gy.to(dtype))
    return wd.grad
d64 = run(torch.float64)
d32 = run(torch.float32)
from micronet.compression.quantization.wbwtab import quantize as Q
conv = Q.QuantConv2d(C, O, 3, padding=1, groups=G, bias=False, W=3).cuda()
conv.weight.data.copy_(w)
for algo in (0, 2):
    ops.CONV_ALGO = algo
    conv.weight.grad = None
    y = conv(x.cuda()); y.backward(gy.cuda())
    dg = conv.weight.grad.double().cpu()
    sc = d64.abs().max()
    print("algo", algo, "gpu-vs-f64 %.2e  cpu32-vs-f64 %.2e  gpu-vs-cpu32 %.2e" % ((dg - d64).abs().max() / sc, (d32.double() - d64).abs().max() / sc, (dg - d32.double()).abs().max() / sc))
